"""CPU: the oracle restatement reproduces the golden vectors captured from the imported reference
(oracle/make_golden.py). Exact equality is expected at the capturing thread count; a 2e-5 tolerance
absorbs OpenMP/oneDNN reduction-order differences on other hosts."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.ref_cpu import deeplab, harness
from oracle.ref_cpu.memory import Memory_sup
from pinthememory_amd import synth

CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
TOL = 2e-5
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / (1.0 + np.abs(b))) <= tol if a.size else True


def argmax_ok(lg, g, tol=1e-3):
    """argmax must match wherever the reference's top-2 margin exceeds 2*tol (SURVEY.md 'Hard parts')."""
    am = lg.argmax(1).to(torch.uint8).numpy()
    safe = g['margin'].astype(np.float32) > 2 * tol
    return bool(np.all(am[safe] == g['argmax'][safe])), float(np.mean(am == g['argmax']))


@pytest.mark.parametrize('name,fac', [('v3plus_r50', deeplab.DeepR50V3PlusD), ('v2_r101', deeplab.DeepR101V2D)])
def test_state_dict_layout(name, fac):
    want = json.load(open(os.path.join(GOLDEN, 'state_dict_%s.json' % name)))
    net = fac(synth.model_args(), 19, CRIT, CRIT)
    got = [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in net.state_dict().items()]
    assert got == want
    assert len(want) == {'v3plus_r50': 397, 'v2_r101': 679}[name]


def test_config1_eval_forward(golden):
    g = golden('config1_v3plus_eval256.npz')
    net = synth.load_det_weights(deeplab.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT)).eval()
    x, _ = synth.make_batch(1, 256)
    with torch.no_grad():
        out = net(x)
    assert close(out[0][:, :, ::8, ::8].numpy(), g['sub'])
    ok, frac = argmax_ok(out[0], g)
    assert ok and frac > 0.999
    assert close(out[1][1][:, ::4, ::4].numpy(), g['score_memory_sub'])
    assert close(out[1][0].sum((0, 1, 2)).numpy(), g['score_query_colsum'])
    assert close(out[2][:, ::16, ::4, ::4].numpy(), g['inter_sub'])


def test_memory_kat(golden):
    g = golden('memory_kat.npz')
    M = Memory_sup(19, 256, 256, 0.8, 1, gumbel_read=False)
    M.load_state_dict(synth.det_state_dict(M))
    M.m_items = synth.det_memory()
    M.train()
    q = torch.relu(synth.det_tensor((2, 256, 12, 12), 99))
    _, mask = synth.make_batch(2, 96, seed=11, block=16)
    assert close(M.m_items.numpy(), g['m_before'], 0)
    out, sq, sm, readloss, (div, cls) = M(q, mask, memory_writing=True, writing_detach=True)
    for got, key in ((out, 'out'), (sq, 'score_query'), (sm, 'score_memory'), (readloss, 'readloss'), (div, 'div'),
                     (cls, 'cls'), (M.m_items, 'm_after')):
        assert close(got.detach().numpy(), g[key]), key
    gg = golden('memory_kat_grad.npz')
    M.load_state_dict(synth.det_state_dict(M))
    M.m_items = synth.det_memory()
    qg = q.clone().requires_grad_(True)
    out, sq, sm, readloss, (div, cls) = M(qg, mask, memory_writing=True, writing_detach=False)
    (out.sum() * 1e-3 + readloss + div + cls).backward()
    assert close(qg.grad.numpy(), gg['dq'], 1e-4)
    for k, v in M.named_parameters():
        assert close(v.grad.numpy(), gg['d_' + k], 1e-4), k


def test_trainstep_kat(golden):
    g = golden('trainstep_v3plus_128.npz')
    net = synth.load_det_weights(deeplab.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT))
    net.dsn[3].p = 0.0
    x, y = synth.make_batch(2, 128)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    opt, _ = harness.make_optimizer(net)
    losses = harness.agg_train_step(net, opt, x, y)
    for k in ('loss1', 'loss2', 'readloss', 'div', 'cls', 'total'):
        assert close(losses[k].numpy(), g[k]), k
    assert close(net.memory.m_items.detach().numpy(), g['m_after'])
    params = dict(net.named_parameters())
    names = [str(s) for s in g['probe_names']]
    gn = np.array([params[k].grad.double().norm().item() for k in names])
    dn = np.array([(params[k].detach() - before[k]).double().norm().item() for k in names])
    assert np.all(np.abs(gn - g['grad_norm']) <= 1e-3 * g['grad_norm'] + 1e-9)
    assert np.all(np.abs(dn - g['delta_norm']) <= 1e-3 * g['delta_norm'] + 1e-9)
    assert close(net.state_dict()['layer1.0.bn1.running_var'].numpy(), g['bn_running_var'])


def test_memory_initialize(golden):
    g = golden('memory_init_v3plus_128.npz')
    net = synth.load_det_weights(deeplab.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT))
    batches = [synth.make_batch(2, 128, seed=304 + i) for i in range(2)]
    assert close(harness.memory_initialize(net, batches).numpy(), g['m_items'])


def test_config5_v2_r101(golden):
    g = golden('config5_v2_r101_eval128.npz')
    net = synth.load_det_weights(deeplab.DeepR101V2D(synth.model_args(), 19, CRIT, CRIT)).eval()
    x, _ = synth.make_batch(1, 128)
    with torch.no_grad():
        out = net(x)
    assert close(out[0][:, :, ::4, ::4].numpy(), g['sub'])
    assert argmax_ok(out[0], g)[0]
    gs = golden('config5_v2_r101_sliding.npz')
    img, _ = synth.make_batch(1, (160, 288), seed=77)
    full = harness.sliding_logits(net, img[0], crop=128)
    assert close(full[:, ::8, ::8].numpy(), gs['sub'])
    assert np.mean(full.argmax(0).numpy() == gs['argmax']) > 0.999


def test_sliding_tiles_kat():
    # SURVEY.md Appendix C3 (formulas of /root/reference/eval.py:158-182)
    assert harness.sliding_tiles(1024, 2048, 1024) == [(0, 0, 1024, 1024), (683, 0, 1707, 1024), (1024, 0, 2048, 1024)]
    assert harness.sliding_tiles(1024, 2048, 768) == [
        (0, 0, 768, 768), (0, 256, 768, 1024), (512, 0, 1280, 768), (512, 256, 1280, 1024), (1024, 0, 1792, 768),
        (1024, 256, 1792, 1024), (1280, 0, 2048, 768), (1280, 256, 2048, 1024)]
    t = harness.sliding_tiles(1024, 2048, 640, overlap=0.5)
    assert len(t) == 18 and sorted({a for a, _, _, _ in t}) == [0, 320, 640, 960, 1280, 1408]
    assert sorted({b for _, b, _, _ in t}) == [0, 320, 384]


def test_fast_hist_miou():
    gt = np.array([0, 0, 1, 1, 255, 2])
    pr = np.array([0, 1, 1, 1, 3, 2])
    h = harness.fast_hist(pr, gt)
    assert h.sum() == 5 and h[0, 0] == 1 and h[0, 1] == 1 and h[1, 1] == 2 and h[2, 2] == 1
    m, iu = harness.miou(h)
    assert abs(iu[0] - 0.5) < 1e-12 and abs(iu[1] - 2 / 3) < 1e-12 and abs(iu[2] - 1) < 1e-12


def test_bf16_storage_emulation_of_the_oracle_is_removable_and_rounds():
    """oracle/bf16_emulation.py (the yardstick of tests/test_model_parity.py::test_bf16_tier_assembled_gradients_vs_fp64_oracle): inside the context every stored activation
    is a bf16 value (so the eval logits move by ~2^-9 of their range, not more, not zero); after it the oracle computes exactly what it computed before."""
    import torch
    from oracle import bf16_emulation
    from oracle.ref_cpu import deeplab
    from pinthememory_amd import synth
    crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
    net = synth.load_det_weights(deeplab.DeepR50V3PlusD(synth.model_args(), 19, crit, crit)).eval()
    x, _ = synth.make_batch(1, 64)
    with torch.no_grad():
        want = net(x)[0]
        with bf16_emulation.bf16_tier(net, deeplab):
            got = net(x)[0]
        again = net(x)[0]
    assert torch.equal(want, again)
    scale = (want.max() - want.min()).item()
    err = (got - want).abs().max().item()
    assert 1e-5 * scale < err < 2e-2 * scale, (err, scale)
    assert deeplab.upsample.__name__ == 'upsample'
