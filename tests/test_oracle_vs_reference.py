"""Pins the oracle against the reference ITSELF (imported from /root/reference; build container only).
Skipped where the reference tree is absent (GPU box)."""
import pytest
import torch

from oracle import import_reference as IR
from oracle.ref_cpu import deeplab, harness
from pinthememory_amd import synth

pytestmark = pytest.mark.skipif(not IR.available(), reason='reference tree only exists in the build container')
CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)


@pytest.fixture(scope='module')
def refmods():
    return IR.load()


def _pair(refmods, which, seed=None):
    rv3, rv2, _ = refmods
    args = synth.model_args()
    fac = {'v3': (rv3.DeepR50V3PlusD, deeplab.DeepR50V3PlusD), 'v2': (rv2.DeepR50V2D, deeplab.DeepR50V2D)}[which]
    if seed is not None:
        torch.manual_seed(seed)
    ref = fac[0](args, 19, CRIT, CRIT)
    if seed is not None:
        torch.manual_seed(seed)
    mine = fac[1](args, 19, CRIT, CRIT)
    return ref, mine


@pytest.mark.parametrize('which', ['v3', 'v2'])
def test_same_seed_same_init_and_keys(refmods, which):
    ref, mine = _pair(refmods, which, seed=5)
    sr, sm = ref.state_dict(), mine.state_dict()
    assert list(sr.keys()) == list(sm.keys())
    assert all(torch.equal(sr[k], sm[k]) for k in sr)
    assert torch.equal(ref.memory.m_items, mine.memory.m_items)


@pytest.mark.parametrize('which', ['v3', 'v2'])
def test_eval_forward_bit_equal(refmods, which):
    ref, mine = _pair(refmods, which)
    synth.load_det_weights(ref), synth.load_det_weights(mine)
    ref.eval(), mine.eval()
    x, _ = synth.make_batch(1, 96)
    with torch.no_grad():
        a, b = ref(x), mine(x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    assert all(torch.equal(u, v) for u, v in zip(a[1], b[1]))


def test_agg_train_step_bit_equal(refmods):
    ref, mine = _pair(refmods, 'v3')
    synth.load_det_weights(ref), synth.load_det_weights(mine)
    ref.dsn[3].p = mine.dsn[3].p = 0.0
    x, y = synth.make_batch(2, 96)
    o1, _ = harness.make_optimizer(ref)
    o2, _ = harness.make_optimizer(mine)
    l1, l2 = harness.agg_train_step(ref, o1, x, y), harness.agg_train_step(mine, o2, x, y)
    assert all(torch.equal(l1[k], l2[k]) for k in l1)
    assert torch.equal(ref.memory.m_items, mine.memory.m_items)
    gr, gm = dict(ref.named_parameters()), dict(mine.named_parameters())
    assert all(torch.equal(gr[k].grad, gm[k].grad) for k in gr)
    sr, sm = ref.state_dict(), mine.state_dict()
    assert all(torch.equal(sr[k], sm[k]) for k in sr)


def test_mldg_train_step_bit_equal(refmods):
    """train_memory_mldg semantics (train.py:493-632) restated in oracle.ref_cpu.harness, driven on the reference's own
    model classes vs the oracle's: identical losses, memory, gradients and post-step parameters."""
    import copy
    ref, mine = _pair(refmods, 'v3')
    synth.load_det_weights(ref), synth.load_det_weights(mine)
    ref.dsn[3].p = mine.dsn[3].p = 0.0
    x, y = synth.make_batch(4, 96)
    res = []
    for net in (ref, mine):
        u1, u2 = copy.deepcopy(net), copy.deepcopy(net)
        opt, _ = harness.make_optimizer(net)
        res.append(harness.mldg_train_step(net, u1, u2, opt, x[:2], y[:2], x[2:], y[2:]))
    assert all(torch.equal(res[0][k], res[1][k]) for k in res[0])
    assert torch.equal(ref.memory.m_items, mine.memory.m_items)
    sr, sm = ref.state_dict(), mine.state_dict()
    assert all(torch.equal(sr[k], sm[k]) for k in sr)
