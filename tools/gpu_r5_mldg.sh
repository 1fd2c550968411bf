cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_model_parity.py -x -q -m gpu -k "mldg" 2>&1 | tail -4
timeout 300 python tools/host_profile.py mldg bf16 3 2>&1 | grep "enqueue"
timeout 600 python bench.py --workload mldg --dtype bf16 --steps 5 --warmup 2 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('mldg bf16', j['ms_per_step'], j['config']['host_enqueue_ms'])"
timeout 600 python bench.py --workload mldg --steps 5 --warmup 2 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('mldg f32', j['ms_per_step'], j['config']['host_enqueue_ms'])"
