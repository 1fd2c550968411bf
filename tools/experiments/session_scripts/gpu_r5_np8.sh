cd "$GRAFT_REPO_ROOT" || exit 1
for p in 1 8 1 8; do
echo "== persistent 128 x 256, PM_C16P=$p"; PM_C16P=$p PM_C16W_CFG=1 PROBE_CONV16=3 timeout 300 python tools/conv16_probe.py 2>&1 | grep -v amdgpu | grep "aspp 3x3 d12\|final1\|dsn\|layer4.conv2\|@96\|sum"
done
