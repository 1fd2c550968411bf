cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "--warmup 3 --steps 10" "--warmup 3 --steps 10" "--warmup 15 --steps 10" "--warmup 3 --steps 20" "--warmup 3 --steps 10" "--warmup 30 --steps 20"; do
  echo -n "$cfg: "; python bench.py $cfg --no-cpu-baseline --no-side --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
