"""CPU: the reference's checkpoint wire format (utils/misc.py:195-216, optimizer.py:45-89) round-trips through the HIP
model's module tree (no kernels are launched: only parameter storage is exercised)."""
import io

import torch

from oracle.ref_cpu import deeplab as o_deeplab, harness as o_harness
from pinthememory_amd import checkpoint, synth
from pinthememory_amd.network import deepv3plus

CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)


def _reference_style_checkpoint():
    """What evaluate_eval() writes for a DDP-wrapped reference net: 'module.'-prefixed keys, SyncBN counters, 'memory'."""
    ref = synth.load_det_weights(o_deeplab.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT))
    opt, sched = o_harness.make_optimizer(ref)
    sd = {'module.' + k: v.clone() for k, v in ref.state_dict().items()}
    ckpt = {'state_dict': sd, 'optimizer': opt.state_dict(), 'scheduler': sched.state_dict(), 'epoch': 7, 'mean_iu': 0.4321,
            'memory': ref.memory.m_items.clone()}
    buf = io.BytesIO()
    torch.save(ckpt, buf)
    buf.seek(0)
    return ref, torch.load(buf)


def test_reference_checkpoint_loads_into_hip_model():
    ref, ckpt = _reference_style_checkpoint()
    net = deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT)
    net.memory.m_items = torch.zeros(19, 256)
    opt, sched = o_harness.make_optimizer(net)
    net, opt, sched, epoch, miou = checkpoint.restore_snapshot(net, opt, sched, ckpt, restore_optimizer_bool=True)
    assert (epoch, miou) == (7, 0.4321)
    sr, sn = ref.state_dict(), net.state_dict()
    assert list(sr) == list(sn) and all(torch.equal(sr[k], sn[k]) for k in sr)
    assert torch.equal(net.memory.m_items, ref.memory.m_items)
    w = net.final1[0].weight                                          # KRSC layout survives load_state_dict
    assert w.permute(0, 2, 3, 1).is_contiguous()


def test_hip_checkpoint_loads_into_reference_layout_and_back():
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT))
    snap = checkpoint.snapshot_dict(net, epoch=3, mean_iu=0.5)
    assert all(k.startswith('module.') for k in snap['state_dict']) and 'memory' in snap
    ref = o_deeplab.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT)
    ref.load_state_dict({k[len('module.'):]: v for k, v in snap['state_dict'].items()})       # what the reference does under DDP
    assert all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), net.state_dict().values()))
    # forgiving restore: a checkpoint trained with another class count skips the mismatching heads (optimizer.py:73-89)
    other = deepv3plus.DeepR50V3PlusD(synth.model_args(), 11, CRIT, CRIT)
    _, skipped = checkpoint.forgiving_state_restore(other, snap['state_dict'])
    assert skipped == ['dsn.4.bias', 'dsn.4.weight', 'final2.0.bias', 'final2.0.weight']
