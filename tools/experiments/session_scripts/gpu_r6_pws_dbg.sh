cd "${GRAFT_REPO_ROOT}" || exit 1
export TMPDIR=/tmp ONE_RING=4
for d in 0 1 2; do
  echo "### dbg=$d"
  for shape in "8 64 192 192 256 1 0 1" "8 128 96 96 512 1 0 1"; do PM_PWSTREAM_DBG=$d timeout 120 python tools/one_conv32.py $shape 20 2>&1 | tail -1; done
done
