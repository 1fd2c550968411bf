cd "${GRAFT_REPO_ROOT}" || exit 1
export TMPDIR=/tmp
for cfg in "PM_X=1" "PM_PWSTREAM=0" "PM_PWSTREAM=0 PM_BATCH_XCD=0 PM_N_GROUP=0" "PM_SPLIT=0"; do env $cfg python tools/experiments/session_scripts/mem_grad_probe.py 2>&1 | grep -v "^Model" | tail -9; done
