cd "${GRAFT_REPO_ROOT}" || exit 1
export TMPDIR=/tmp
for mk in 129 0; do
 echo "### PM_SPLIT_MIN_K=$mk"
 for shape in "8 64 192 192 256 1 0 1" "8 128 96 96 512 1 0 1" "8 64 192 192 64 1 0 1"; do PM_SPLIT_MIN_K=$mk timeout 120 python tools/one_conv32.py $shape 20 2>&1 | tail -1; done
 PM_SPLIT_MIN_K=$mk bash tools/gpu_r6_traffic1.sh sk$mk "8 64 192 192 256 1 0 1" "8 128 96 96 512 1 0 1"
done
