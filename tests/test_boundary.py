"""CPU: the drop-in boundary (no kernels launched): arch-string resolution exactly as the reference's get_model does it
(network/__init__.py:36-46), same-seed initial weights as the oracle (== the reference, tests/test_oracle_vs_reference.py),
memory attribute handling, and the loud failure without a GPU."""
import argparse
import importlib

import pytest
import torch

from oracle.ref_cpu import deeplab as o_deeplab
from pinthememory_amd import network, synth
from pinthememory_amd.network import deepv2, deepv3plus

CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)


@pytest.mark.parametrize('arch,ofac', [('pinthememory_amd.network.deepv3plus.DeepR50V3PlusD', o_deeplab.DeepR50V3PlusD),
                                      ('pinthememory_amd.network.deepv3plus.DeepR50V3PlusD_OS8', o_deeplab.DeepR50V3PlusD_OS8),
                                      ('pinthememory_amd.network.deepv2.DeepR50V2D', o_deeplab.DeepR50V2D)])
def test_arch_string_resolves_and_same_seed_init_matches_reference(arch, ofac):
    args = synth.model_args()
    args.arch = arch
    torch.manual_seed(11)
    net = network.get_model(args, 19, CRIT, CRIT)             # importlib path of the reference's get_model
    torch.manual_seed(11)
    ref = ofac(synth.model_args(), 19, CRIT, CRIT)
    sn, sr = net.state_dict(), ref.state_dict()
    assert list(sn) == list(sr) and all(torch.equal(sn[k], sr[k]) for k in sr)      # same RNG consumption order as the reference
    assert torch.equal(net.memory.m_items, ref.memory.m_items)
    assert [n for n, _ in net.named_parameters()] == [n for n, _ in ref.named_parameters()]
    assert sum(n.split('.')[0] == 'memory' for n, _ in net.named_parameters()) == 8  # train.py:549-552 freezes by this name component
    assert net.output_stride in (8, 16) and hasattr(net.aspp, 'features' if 'v3plus' in arch else 'conv2d_list')


def test_memory_attribute_follows_module_and_can_be_reassigned():
    net = deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT)
    assert 'm_items' not in net.state_dict() and not any('m_items' in k for k in net.state_dict())   # plain attribute
    net.memory.m_items = torch.ones(19, 256)                                                      # train.py:332 style reset
    assert net.double().memory.m_items.dtype == torch.float64                                     # follows _apply like the buffers do
    assert net.memory.mem_cls.tolist() == list(range(19))


def test_get_net_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    args = synth.model_args()
    args.arch = 'pinthememory_amd.network.deepv3plus.DeepR50V3PlusD'
    with pytest.raises(RuntimeError, match='no CPU path'):
        network.get_net(args, CRIT, CRIT)


def test_unsupported_configurations_are_rejected():
    with pytest.raises(AssertionError):
        deepv3plus.DeepR50V3PlusD(synth.model_args(wt_layer=[0, 0, 1, 0, 0, 0, 0]), 19, CRIT, CRIT)     # whitening: out of scope
    with pytest.raises(ValueError):
        deepv3plus.DeepV3Plus(19, trunk='resnet-101', criterion=CRIT, criterion_aux=CRIT, variant='D16', args=synth.model_args())


def test_synthetic_domain_source_is_deterministic():
    """The loader stand-in of the input edge: [B, D, H, W, 3] uint8 + [B, D, H, W] uint8 train ids, the same bytes for the
    same (seed, batch index) whatever the host ring size; `static` replays the ring."""
    from pinthememory_amd import input_edge
    a = input_edge.SyntheticDomainSource(2, 3, (32, 48), n_buffers=2, seed=1)
    b = input_edge.SyntheticDomainSource(2, 3, (32, 48), n_buffers=1, seed=1)
    for i in range(4):
        ia, la = next(a)
        ib, lb = next(b)
        assert ia.shape == (2, 3, 32, 48, 3) and la.shape == (2, 3, 32, 48) and ia.dtype == la.dtype == torch.uint8
        assert torch.equal(ia, ib) and torch.equal(la, lb)
        assert set(la.unique().tolist()) <= set(range(19)) | {255}
    s = input_edge.SyntheticDomainSource(1, 2, 32, n_buffers=2, seed=2, static=True)
    first = [next(s)[0].clone() for _ in range(4)]
    assert torch.equal(first[0], first[2]) and torch.equal(first[1], first[3]) and not torch.equal(first[0], first[1])


def test_stitching_beyond_64_tiles_takes_the_torch_formulation():
    """pm_sliding_stitch takes at most 64 tiles by value; a 1024 x 2048 image at crop 256 has 72 (eval.py:158-182). harness.sliding_logits then stitches
    with the float64 torch formulation (any device, no cap): same sums in tile order / true count / un-flip as the oracle's sliding_logits."""
    import torch
    from oracle.ref_cpu import harness as o_h
    from pinthememory_amd import harness as h
    tiles = h.sliding_tiles(1024, 2048, 256)
    assert tiles == o_h.sliding_tiles(1024, 2048, 256) and len(tiles) == 72 > h.STITCH_MAX_TILES
    g = torch.Generator().manual_seed(3)
    acc, ref = None, None
    for flip in (False, True):
        lg = torch.randn(len(tiles), 3, 256, 256, generator=g)
        full, cnt = torch.zeros(3, 1024, 2048, dtype=torch.float64), torch.zeros(1, 1024, 2048, dtype=torch.float64)
        for t, (x1, y1, x2, y2) in zip(lg, tiles):
            full[:, y1:y2, x1:x2] += t.double()
            cnt[:, y1:y2, x1:x2] += 1
        full = full / cnt
        ref = (torch.flip(full, dims=[2]) if flip else full) + (0 if ref is None else ref)
        acc = h._stitch_torch(lg, tiles, 1024, 2048, flip, acc)
    assert torch.equal(acc, ref)


def test_m_items_stays_a_plain_get_set_attribute():
    """The reference treats Memory_sup.m_items as a plain attribute: read, re-assigned from outside (train.py:312,332,547,558,580,1040; optimizer.py:65),
    never part of state_dict, and networks are deep-copied with it (train.py:246-277). Here it is a property (it orders a reader behind an overlapped
    commit forward on the GPU); on the surface nothing changes."""
    import copy
    import torch
    from pinthememory_amd.network.memory import Memory_sup
    m = Memory_sup(19, 256, 256, 0.8, 1, True)
    assert tuple(m.m_items.shape) == (19, 256) and 'm_items' not in m.state_dict() and not any('m_items' in k for k in m.state_dict())
    t = torch.nn.functional.normalize(torch.rand(19, 256), dim=1)
    m.m_items = t
    assert m.m_items is t and m.pending is None
    m2 = copy.deepcopy(m)
    assert torch.equal(m2.m_items, t) and m2.m_items is not t
    m2.m_items = t * 2
    assert torch.equal(m.m_items, t)
    m.double()                                             # nn.Module._apply follows the attribute (the reference hard-codes .cuda(), memory.py:111,120)
    assert m.m_items.dtype == torch.float64 and m.mem_cls.dtype == torch.int64


def test_m_items_never_hides_a_collective():
    """with more than one rank the commit forward leaves the cross-rank sum of its write to a point every rank passes (the next forward's read,
    harness.finish_commit). An attribute read in between -- the reference saves on rank 0 only, train.py:188-191 -- must fail loudly, not issue an
    all-reduce that the other ranks answer with a different collective. Assigning the attribute (a restore) drops the owed write."""
    import pytest
    import torch
    from pinthememory_amd.network import memory
    m = memory.Memory_sup(19, 256, 256, 0.8, 1, True)
    t = m.m_items
    memory._DEFERRED[m] = (t, torch.zeros(20 * 257))
    with pytest.raises(RuntimeError, match='finish_commit'):
        m.m_items
    assert m in memory._DEFERRED                    # the failed read consumed nothing
    # a restore while the sum is owed fails loudly too (on a subset of ranks it would drop the all-reduce there and strand the others: ADVICE r4)
    from pinthememory_amd import checkpoint

    class Holder(torch.nn.Module):
        def __init__(self, mem):
            super().__init__()
            self.memory = mem
    holder = Holder(m)
    assert m.commit_owed
    with pytest.raises(RuntimeError, match='finish_commit'):
        checkpoint.restore_snapshot(holder, None, None, {'state_dict': {}, 'memory': t * 3})
    assert m in memory._DEFERRED
    m.m_items = t * 1
    assert m not in memory._DEFERRED and not m.commit_owed and torch.equal(m.m_items, t)


def test_set_mode_respects_train_overrides():
    """harness.set_mode writes the flag into each module's __dict__ (0.5 ms per step saved) -- unless some class of the tree overrides train(): then the real
    net.train(mode) runs (frozen-BatchNorm variants and user wrappers rely on it)."""
    import torch
    from pinthememory_amd import harness

    class Frozen(torch.nn.BatchNorm2d):
        def train(self, mode=True):
            return super().train(False)      # stays in eval mode whatever the parent says
    plain = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
    harness.set_mode(plain, False)
    assert not plain.training and not plain[1].training
    harness.set_mode(plain, True)
    assert plain.training and plain[1].training
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), Frozen(4))
    harness.set_mode(net, True)
    assert net.training and net[0].training and not net[1].training


def test_weight_gradient_overlap_default_is_per_tier(monkeypatch):
    """hip/ops.overlap_wgrad (round 5): unset PM_OVERLAP_WGRAD = side stream on the fp32 tier, inline on the bf16 tier (its weight-gradient kernel is persistent: one block
    per CU); the module attribute stays the master switch bench.py and the tests flip; an explicit PM_OVERLAP_WGRAD=1 keeps the side stream on both tiers."""
    from pinthememory_amd.hip import ops, kernels as K
    prec, act = K.CONV_PREC, K.ACT_DTYPE
    try:
        monkeypatch.setattr(ops, '_OVERLAP_ENV', None)
        monkeypatch.setattr(ops, 'OVERLAP_WGRAD', True)
        K.set_conv_precision('f32')
        assert ops.overlap_wgrad()
        K.set_conv_precision('bf16')
        assert not ops.overlap_wgrad()
        monkeypatch.setattr(ops, '_OVERLAP_ENV', '1')
        assert ops.overlap_wgrad()
        monkeypatch.setattr(ops, 'OVERLAP_WGRAD', False)
        assert not ops.overlap_wgrad()
    finally:
        K.CONV_PREC, K.ACT_DTYPE = prec, act
