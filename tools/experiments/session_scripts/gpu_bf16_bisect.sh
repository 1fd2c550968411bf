cd "$GRAFT_REPO_ROOT"
for E in "-" "PM_COMMIT_OVERLAP=0" "PM_BN_MASK=0" "-" "PM_BF16_WGRAD_TR=0"; do
  if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
  env $EV timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-profile --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$E', d['ms_per_step'])"
done
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('with profile pass', d['ms_per_step'])"
