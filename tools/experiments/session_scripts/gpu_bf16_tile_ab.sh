cd "$GRAFT_REPO_ROOT"
for E in "-" "PM_NST1_STEPS=0" "PM_FORCE_BM=64" "PM_FORCE_BN=64" "PM_NST1_STEPS=8"; do
  if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
  env $EV timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-profile --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$E', d['ms_per_step'])"
done
