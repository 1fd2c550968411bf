cd "${GRAFT_REPO_ROOT}" || exit 1
export TMPDIR=/tmp ONE_RING=4
for cfg in "2 2" "3 2" "3 3" "2 3" "2 4" "4 2"; do
  set -- $cfg
  echo "### per_cu=$1 nb=$2"
  for shape in "8 64 192 192 256 1 0 1" "8 64 192 192 64 1 0 1" "8 128 96 96 512 1 0 1"; do PM_PWSTREAM_PER_CU=$1 PM_PWSTREAM_NB=$2 timeout 120 python tools/one_conv32.py $shape 20 2>&1 | tail -1; done
  ONE_MODE=dgrad PM_PWSTREAM_PER_CU=$1 PM_PWSTREAM_NB=$2 timeout 120 python tools/one_conv32.py 8 256 192 192 64 1 0 1 20 2>&1 | tail -1
done
