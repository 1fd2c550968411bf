"""bench.py --gpus N without a launcher environment starts N rank processes itself.

The reference reads its world from the launcher's environment (/root/reference/train.py:63-76) and its multi-GPU scripts go through
`torch.distributed.launch --nproc_per_node N`; the driver calls `python bench.py --gpus N`, so bench.py does that hop: fresh children,
never an exec of a process that has initialised the GPU. Children are stubbed here, so no GPU is needed.
"""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stub(tmp_path, body):
    p = tmp_path / 'child.py'
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_launch_relays_rank0_line_and_sets_rank_env(tmp_path, capsys):
    out = tmp_path / 'seen'
    out.mkdir()
    child = _stub(tmp_path, '''
        import json, os, sys
        r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
        assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
        open(os.path.join(%r, 'rank%%d' %% r), 'w').write(os.environ['MASTER_PORT'])
        print('not json chatter rank %%d' %% r)
        if r == 0:
            print(json.dumps({'metric': 'm', 'n_gpus': w}))
    ''' % str(out))
    rc = bench.launch_ranks(3, [], child=child)
    assert rc == 0
    assert sorted(os.listdir(out)) == ['rank0', 'rank1', 'rank2']
    assert len({(out / f).read_text() for f in os.listdir(out)}) == 1          # one rendezvous port for all ranks
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0]) == {'metric': 'm', 'n_gpus': 3}      # only rank 0's stdout is relayed


def test_launch_returns_worst_exit_code(tmp_path):
    child = _stub(tmp_path, '''
        import os, sys
        r = int(os.environ['RANK'])
        if r == 0:
            print('{"metric": "m"}')
        sys.exit(7 if r == 1 else 0)
    ''')
    assert bench.launch_ranks(2, [], child=child) == 7


def test_launch_fails_without_a_json_line(tmp_path):
    child = _stub(tmp_path, 'print("no line")\n')
    assert bench.launch_ranks(2, [], child=child) == 1


def test_bench_py_gpus_2_takes_the_hop_on_a_cpu_box():
    """End to end through main(): no RANK / WORLD_SIZE in the environment -> two children of bench.py itself; without a GPU each rank
    stops at the 'needs a GPU' assertion (no CPU fallback) and the parent reports the failure instead of printing n_gpus: 1."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('CPU-box behaviour')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert 'n_gpus' not in p.stdout
    assert p.stderr.count('needs a GPU') >= 2            # both ranks ran


def test_gpus_flag_must_match_launcher_world():
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and 'WORLD_SIZE=1' in p.stderr


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_bench_py_gpus_2_self_launch_on_the_gpu_box(tmp_path, dtype):
    """the driver's call -- `python bench.py --gpus 2`, no outer launcher -- as a FRESH child process on the GPU box: the parent starts two
    ranks of itself (gloo: both share the box's one GPU; RCCL refuses two ranks on a device), rank 0's single JSON line comes back with n_gpus 2, two ranks
    seen, one step time per rank, and a finite loss; the whole N > 1 code path (SyncBatchNorm exchanges, bucketed gradient all-reduce, memory-slot sum,
    overlapped commit forward with its deferred sum) ran inside it -- on the fp32 line and on the bf16 tier (bf16 activations under SyncBatchNorm's fp32 statistics
    exchange, the batched refresh of the kept bf16 filters after the all-reduced step). The log is kept under profiles/ by tools/gpu_round4_profiles.sh."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['PM_BENCH_BACKEND'] = 'gloo'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--size', '256', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
                        '--dtype', dtype], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    c = d['config']
    assert d['n_gpus'] == 2 and c['ranks_seen'] == 2 and len(c['ms_per_step_per_rank']) == 2 and c['global_batch'] == 16 and c['parallelism'] == 'dp2'
    assert d['scaling'] == 'weak' and d['value'] > 0 and d['cpu_baseline'] is None and d['dtype'] == dtype
    assert abs(d['value'] - 16 / (d['ms_per_step'] * 1e-3)) < 1e-2 * d['value']                 # whole-job rate: both ranks' images over the slowest rank's time
    assert c['final_loss'] == c['final_loss'] and abs(c['final_loss']) < 1e3                     # finite
    # round 5: the line says how the exchanges travelled and how many there were, and what the host spent enqueueing one step
    assert c['rccl_direct'] is False and 'gloo' in c['rccl_direct_reason']
    # round 6: 138 -> 122: the independent SyncBatchNorm exchanges travel together -- ASPP's five branches (5 -> 1 per direction), bn3 + downsample.1 of
    # each stage's first block (2 -> 1 per direction, four stages); what is left: 49 + 49 SyncBN exchanges, the gradient buckets, the memory-slot sum
    assert 100 <= c['collectives_per_step'] <= 128, c['collectives_per_step']
    assert c['host_enqueue_ms'] > 0 and c['step_form'] == 'eager launches'
