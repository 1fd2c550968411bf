cd "${GRAFT_REPO_ROOT}" || exit 1
export TMPDIR=/tmp ONE_RING=4
for cfg in "PM_SPLIT_MIN_K=129" "PM_SPLIT_MIN_K=0" "PM_SPLIT_MIN_K=0 PM_FORCE_BM=64 PM_FORCE_BN=64" "PM_SPLIT_MIN_K=0 PM_FORCE_BM=64 PM_FORCE_BN=128" "PM_SPLIT_MIN_K=0 PM_FORCE_BM=128 PM_FORCE_BN=64"; do
 echo "### $cfg"
 for shape in "8 64 192 192 256 1 0 1" "8 128 96 96 512 1 0 1" "8 64 192 192 64 1 0 1"; do env $cfg timeout 120 python tools/one_conv32.py $shape 20 2>&1 | tail -1; done
done
