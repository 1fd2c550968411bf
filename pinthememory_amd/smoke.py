"""smoke(): one tiny train step of R50-DeepLabV3+ + memory on cuda:0 through the HIP path, checked against the CPU oracle."""
import torch


def run(size=96, batch=2, verbose=True):
    from oracle.ref_cpu import deeplab as o_deeplab, harness as o_harness   # checker only
    from . import harness, synth
    from .network import deepv3plus
    crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
    args = synth.model_args()
    x, y = synth.make_batch(batch, size)
    ref = synth.load_det_weights(o_deeplab.DeepR50V3PlusD(args, 19, crit, crit))
    ref.dsn[3].p = 0.0
    o_opt, _ = o_harness.make_optimizer(ref)
    want = o_harness.agg_train_step(ref, o_opt, x, y)
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(args, 19, crit, crit)).cuda()
    net.dsn[3].p = 0.0
    opt, _ = harness.make_optimizer(net)
    got = harness.agg_train_step(net, opt, x.cuda(), y.cuda())
    torch.cuda.synchronize()
    for k in want:
        a, b = got[k].item(), want[k].item()
        if verbose:
            print('smoke %-8s hip %.6f  oracle %.6f' % (k, a, b))
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (k, a, b)
    # post-step memory: parameters moved by fp32-noisy trunk gradients (train-mode BN at batch 2), so only a loose bound
    # is meaningful here; tests/test_model_parity.py bounds it against an fp64 oracle
    dm = (net.memory.m_items.cpu() - ref.memory.m_items).abs().max().item()
    assert dm < 5e-3, dm
    if verbose:
        print('smoke ok: max |m_items - oracle| = %.2e' % dm)
