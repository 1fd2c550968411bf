#!/bin/bash
# bf16 step: weight gradients inline / on the side stream x commit forward inline / on its stream, captured and eager forms, one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do
for form in "--graph" ""; do
for w in 0 1; do for c in 0 1; do
  PM_OVERLAP_WGRAD=$w PM_COMMIT_OVERLAP=$c timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --no-profile $form --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('form [$form] wgrad overlap $w commit overlap $c:', j['ms_per_step'], j['config'].get('step_form'))"
done; done; done; done
