#!/usr/bin/env python3
"""bench.py -- train imgs/sec of the reference's aggregation step (train.py:312-335 semantics) on the HIP path.

Workload (BASELINE.json configs[1]): ResNet-50 DeepLabV3+ + memory, bs=8 per GPU, 768x768 synthetic Cityscapes-shaped
batch, fp32, one step = train forward (memory read + non-detached write) -> 5-term loss -> backward -> SGD ->
eval-mode second forward that commits the memory (all of it inside the timed region).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS_F32_MFMA = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_TFLOPS_BF16_MFMA = 2500.0        # same guide: bf16 MFMA dense peak (never the 2:1-sparsity figure)
STEP_GFLOP_PER_IMG = 1311.8            # train fwd 334.16 + bwd 665.55 + eval-mode 2nd fwd 312.07


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU (weak scaling)')
    ap.add_argument('--size', type=int, default=768)
    ap.add_argument('--truncate-second-forward', action='store_true',
                    help='skip the decoder in the memory-commit forward (identical results; NOT the default measurement)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-side', action='store_true', help='skip the side.bf16 sub-measurement of the default (f32, one GPU) run')
    ap.add_argument('--launch-timeout', type=float, default=3600.0, help='--gpus N self-launch: seconds after which every rank is killed and the run fails')
    ap.add_argument('--cpu-batch', type=int, default=2)
    ap.add_argument('--no-profile', action='store_true', help='do not bracket conv launches with HIP events')
    ap.add_argument('--dtype', choices=['f32', 'bf16', 'bf16_operands', 'bf16_staged'], default='f32',
                    help='f32 = BASELINE configs[1] (the metric; default). bf16 = configs[2], the whole tier (round 4): activations and their gradients are stored as '
                         'bf16 between layers, bf16 MFMA convolutions with fp32 accumulation, fp32 statistics / losses / memory / parameters. bf16_operands = round 2-3 '
                         'form (fp32 activations, every convolution converts its operands to bf16 in HBM). bf16_staged = the first form (fp32 tiles in LDS rounded per '
                         'fragment); both kept for A/B')
    ap.add_argument('--graph', action='store_true',
                    help='time harness.GraphedAggStep(pipelined=True): the step replayed as ONE hipGraph launch -- [commit forward of the previous step || training forward] '
                         '+ backward + SGD, the eager step\'s cross-step overlap captured inside the graph; same operations per step, bit-identical results '
                         '(tests/test_model_parity.py::test_graphed_agg_step_is_bit_identical_to_eager). Single GPU only')
    ap.add_argument('--input-edge', action='store_true',
                    help='side measurement (not the metric): every step takes a fresh uint8 [B, D, H, W, 3] batch from pinned host memory, copied and '
                         'converted on a side stream while the previous step computes (pinthememory_amd/input_edge.py)')
    ap.add_argument('--workload', choices=['train', 'config5', 'meminit', 'mldg'], default='train',
                    help="train = the metric (default). config5 = side measurement of BASELINE configs[4]: ResNet-101 DeepLabV2 sliding-window "
                         "evaluation of 1024x2048 images (crop 1024, overlap 1/3 -> 3 tiles x 2 flips, eval.py:148-274), single GPU. meminit = side "
                         "measurement of the caller before training (train.py:1000-1042): class-prototype initialisation of the memory over --steps batches of "
                         "bs=8 768x768, eval forwards + the write kernel's soft-label accumulation. mldg = side measurement of the regime every pinmem script "
                         "runs (train.py:493-632, train_GS_pinmem_DR50V3P.sh:9-10): meta-train 4 + meta-test 4 images at 768x768, three weight sets, retain_graph")
    return ap.parse_args()


CONFIG5_GFLOP_PER_TILE = 2057.0        # R101-DeepLabV2 'D' forward on one 1x3x1024x1024 tile


def config5(a):
    """Side measurement (not the metric): sliding-window logits of one resident 1024x2048 image through DeepR101V2D + argmax,
    `--steps` images after `--warmup`; properties checked on the result: finite logits, every pixel covered, deterministic repeat."""
    import torch
    from pinthememory_amd import harness, synth
    from pinthememory_amd.network import deepv2
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the HIP path has no CPU fallback'
    dev = torch.device('cuda', 0)
    crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
    net = synth.load_det_weights(deepv2.DeepR101V2D(synth.model_args(), 19, crit, crit)).to(dev)
    img = torch.randn(3, 1024, 2048, generator=torch.Generator().manual_seed(304)).to(dev)
    tiles = harness.sliding_tiles(1024, 2048, 1024, 1.0 / 3)

    def one():
        return harness.sliding_logits(net, img, 1024).argmax(0)

    for _ in range(a.warmup):
        first = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pred = one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    logits = harness.sliding_logits(net, img, 1024)
    assert torch.isfinite(logits).all() and tuple(logits.shape) == (19, 1024, 2048)
    assert a.warmup == 0 or torch.equal(first, pred), 'sliding-window evaluation is not run-to-run deterministic'
    tf = CONFIG5_GFLOP_PER_TILE * len(tiles) * 2 / 1e3
    roof = None
    if not a.no_profile:
        # dominant convolution kernel of one image, from two extra untimed images with the in-library HIP events on (one stream: no overlap to serialise)
        from pinthememory_amd.hip import kernels as K
        K.profile_enable(True)
        for _ in range(2):
            one()
        torch.cuda.synchronize()
        K.profile_enable(False)
        if os.environ.get('PM_PROFILE_DUMP'):
            K.profile_dump(os.environ['PM_PROFILE_DUMP'])
        best = dominant_conv_kernel(K, False)
        tot_ms, tot_fl, tot_n = K.profile_read(clear=True)
        if best:
            (ms, fl, n), sym, what = best
            ach = fl / (ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': '%s (%s)' % (sym, what), 'achieved': round(ach, 2), 'peak': PEAK_TFLOPS_F32_MFMA, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_TFLOPS_F32_MFMA, 4), 'traffic': None, 'launches_per_image': n / 2, 'avg_launch_ms': round(ms / n, 5),
                    'gflop_per_launch': round(fl / n / 1e9, 3),
                    'measured': '2 extra images after the timed region, every convolution launch bracketed by HIP events on its stream (pm_profile_*)',
                    'all_conv_kernels': {'achieved': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), 'ms_per_image': round(tot_ms / 2, 3), 'launches_per_image': tot_n / 2,
                                         'executed_tflop_per_image': round(tot_fl / 2 / 1e12, 3)},
                    'image': {'executed_tflop': round(tot_fl / 2 / 1e12, 3), 'mfma_frac_executed': round(tot_fl / 2 / 1e12 / (dt / a.steps) / PEAK_TFLOPS_F32_MFMA, 4)}}
    print(json.dumps({'metric': 'eval imgs/sec 1024x2048 R101-DeepLabV2 sliding window (crop 1024, 2 flips)', 'value': round(a.steps / dt, 3),
                      'unit': 'imgs/sec', 'n_gpus': 1, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3),
                      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                      'config': {'workload': 'configs[4]: ResNet-101 DeepLabV2 (network/deepv2.py) 1024x2048 sliding-window inference, side measurement',
                                 'step_form': 'eager launches', 'host_enqueue_ms': None if a.no_profile else host_enqueue_ms(one),
                                 'tiles': [list(t) for t in tiles], 'flips': 2, 'conv_tflop_per_image': round(tf, 2),
                                 'direct_equivalent_mfma_frac': round(tf * a.steps / dt / PEAK_TFLOPS_F32_MFMA, 4)},
                      'roofline': roof, 'cpu_baseline': None}), flush=True)


def meminit(a):
    """Side measurement (not the metric): harness.memory_initialize (train.py:1000-1042) at the size it runs at -- `--steps` resident batches of bs x 3 x 768 x 768
    per epoch, one epoch timed after `--warmup` untimed batches; properties checked: finite unit-norm prototypes, bit-identical repeat."""
    import torch
    from pinthememory_amd import harness, synth
    from pinthememory_amd.network import deepv3plus
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the HIP path has no CPU fallback'
    crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, crit, crit)).cuda()
    batches = [tuple(t.cuda() for t in synth.make_batch(a.batch, a.size, seed=304 + i)) for i in range(max(a.steps, 1))]
    if a.warmup:
        harness.memory_initialize(net, batches[:max(1, min(a.warmup, len(batches)))], epochs=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m1 = harness.memory_initialize(net, batches, epochs=1).clone()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m2 = harness.memory_initialize(net, batches, epochs=1)
    assert torch.isfinite(m1).all() and (m1.norm(dim=1) - 1).abs().max().item() < 1e-5 and torch.equal(m1, m2)
    print(json.dumps({'metric': 'memory initialisation imgs/sec 768x768 bs=%d R50-DeepLabV3+ (train.py:1000-1042)' % a.batch, 'value': round(a.batch * len(batches) / dt, 3),
                      'unit': 'imgs/sec', 'n_gpus': 1, 'steps': len(batches), 'warmup': a.warmup, 'ms_per_step': round(dt / len(batches) * 1e3, 3),
                      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                      'config': {'workload': 'memory_initalize (train.py:1000-1042): eval-mode forward of each batch + 4-tap soft-label accumulation of the normalised '
                                             'bot_aspp features, one epoch over %d resident batches, side measurement' % len(batches),
                                 'step_form': 'eager launches',
                                 'host_enqueue_ms': None if a.no_profile else round(host_enqueue_ms(lambda: harness.memory_initialize(net, batches, epochs=1)) / len(batches), 3)},
                      'roofline': None, 'cpu_baseline': None}), flush=True)


def mldg(a):
    """Side measurement (not the metric): harness.mldg_train_step -- the meta-learning regime of train_memory_mldg (train.py:493-632) at the scripts' size: the bs=8
    batch split into 4 meta-train + 4 meta-test images at 768 x 768; per step: inner forward + backward (retain_graph), two functional weight sets
    (theta' = theta - lr g), frozen-encoder memory write, meta-test forward + backward through the written memory, SGD, memory-commit forward.
    Reports ms/step, peak memory, per-family kernel time (in-library conv events) and how many filter transforms the functional weights force per step."""
    import copy
    import torch
    from pinthememory_amd import harness, synth
    from pinthememory_amd.hip import kernels as K
    from pinthememory_amd.network import deepv3plus
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the HIP path has no CPU fallback'
    K.set_conv_precision(a.dtype)
    crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
    u1, u2 = copy.deepcopy(net), copy.deepcopy(net)
    opt, sched = harness.make_optimizer(net)
    x, y = synth.make_batch(a.batch, a.size, seed=304)
    x, y = x.cuda(), y.cuda()
    h = a.batch // 2

    inner = [harness.INNER_LR]      # train.py:1208 default 1e-3; every pinmem script anneals it to lr / 4 after each scheduler step (train.py:625-626)

    def step():
        out = harness.mldg_train_step(net, u1, u2, opt, x[:h], y[:h], x[h:], y[h:], inner_lr=inner[0], sched=sched, inner_lr_anneal=True)
        inner[0] = out.pop('next_inner_lr')
        return out
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    peak_gb = torch.cuda.max_memory_allocated() / 1e9
    fam = None
    if not a.no_profile:
        K.profile_enable(True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        K.profile_enable(False)
        if os.environ.get('PM_PROFILE_DUMP'):
            K.profile_dump(os.environ['PM_PROFILE_DUMP'])
        fam = {}
        for name, mode in (('forward-form kernel (fp32: forward + Winograd GEMMs; bf16: forward + stride-1 data gradients, register-staged)', 0), ('data-gradient-form kernel (fp32: direct data gradients; bf16: stride-2 only)', 1), ('weight gradients (bf16 tier: the LDS-DMA persistent ring of wgrad16.hip where the shape allows)', 2), ('LDS-DMA bf16 convolutions (forward + stride-1 data gradients)', 4), ('LDS-DMA bf16 convolutions, one block per CU: persistent producer / consumer ring and the 256 x 256 two-stage form', 5)):
            ms, fl, n = K.profile_read(mode=mode)
            if n:
                fam[name] = {'ms_per_step': round(ms / 2, 3), 'launches_per_step': n / 2, 'achieved_TFLOPs': round(fl / (ms * 1e-3) / 1e12, 1)}
        ms, fl, n = K.profile_read(clear=True)
        fam['all convolution kernels (weight gradients on their side stream: durations inflate)'] = {'ms_per_step': round(ms / 2, 3), 'launches_per_step': n / 2}
    xf0 = K.filter_transform_count(enable=True)
    step()
    torch.cuda.synchronize()
    xforms = K.filter_transform_count(enable=False) - xf0
    assert all(torch.isfinite(v).all() for v in losses.values())
    print(json.dumps({'metric': 'mldg train imgs/sec 768x768 bs=%d+%d R50-DeepLabV3+ +mem (train.py:493-632)' % (h, a.batch - h), 'value': round(a.batch * a.steps / dt, 3),
                      'unit': 'imgs/sec', 'n_gpus': 1, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True,
                      'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
                      'config': {'workload': 'train_memory_mldg (train.py:493-632; every pinmem script: train_GS_pinmem_DR50V3P.sh:9-10): %d meta-train + %d meta-test images %dx%d, '
                                             'inner fwd + bwd (retain_graph), theta\' = theta - inner_lr g for two weight sets (inner_lr 1e-3 at the first iteration, then lr / 4 = 2.5e-3: --inner_lr_anneal), frozen-encoder memory write, meta-test fwd + bwd '
                                             'through the written memory, SGD, eval-mode memory-commit fwd; side measurement' % (h, a.batch - h, a.size, a.size),
                                 'inner_lr_first': harness.INNER_LR, 'inner_lr_last': inner[0], 'peak_memory_GB': round(peak_gb, 2), 'final_losses': {k: round(float(v), 5) for k, v in losses.items()},
                                 'filter_transforms_per_step': xforms, 'step_form': 'eager launches', 'host_enqueue_ms': None if a.no_profile else host_enqueue_ms(step),
                                 'filter_transforms_note': 'Winograd U = G w G^T (fp32 tier) / bf16 filter copies (bf16 tier) computed inside one step; functional weights '
                                                           '(theta\') are not owner-registered parameters, so their transforms are never kept between calls',
                                 'kernel_families': fam},
                      'roofline': None, 'cpu_baseline': None}), flush=True)


def mldg_side(torch, harness, K, net, opt, sched, x, y, tier, steps=5, warm=2):
    """side.mldg / side.mldg_bf16 of the default line: the regime every pinmem script runs (train_GS_pinmem_DR50V3P.sh:9-10,18 -> train.py:493-632),
    4 meta-train + 4 meta-test images of the SAME model and batch, inner_lr 1e-3 then lr / 4 (--inner_lr_anneal), timed as eager launches and as ONE hipGraph launch per
    step (harness.GraphedMldgStep). The reported form follows the side.bf16 rule: the graph form when the eager step's host enqueue exceeds 0.8 x its wall time."""
    import copy
    K.set_conv_precision(tier)
    h = x.shape[0] // 2
    u1, u2 = copy.deepcopy(net), copy.deepcopy(net)
    inner = [harness.INNER_LR]

    def step():
        out = harness.mldg_train_step(net, u1, u2, opt, x[:h], y[:h], x[h:], y[h:], inner_lr=inner[0], sched=sched, inner_lr_anneal=True)
        inner[0] = out.pop('next_inner_lr')
        return out
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        le = step()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / steps * 1e3
    enq = host_enqueue_ms(step)
    what = ('train_memory_mldg (train.py:493-632; train_GS_pinmem_DR50V3P.sh:9-10,18): %d meta-train + %d meta-test images %dx%d of the same model, inner fwd + bwd (retain_graph), '
            "theta' = theta - inner_lr g for two weight sets (inner_lr 1e-3, then lr / 4), frozen-encoder memory write, meta-test fwd + bwd through the written memory, SGD, "
            'eval-mode memory-commit fwd; %s' % (h, x.shape[0] - h, x.shape[2], x.shape[3], 'bf16 tier' if tier == 'bf16' else 'fp32 tier'))
    rec = {'workload': what, 'ms_per_step': round(eager_ms, 3), 'value': round(x.shape[0] / eager_ms * 1e3, 3), 'unit': 'imgs/sec', 'steps': steps, 'warmup': warm, 'dtype': tier,
           'form': 'eager launches', 'host_enqueue_ms': enq, 'final_losses': {k: round(float(v), 5) for k, v in le.items()}, 'measured': 'after the timed fp32 region of this run, same process'}
    try:
        g = harness.GraphedMldgStep(net, u1, u2, opt, x[:h], y[:h], x[h:], y[h:], inner_lr=inner[0], sched=sched, warmup=0, inner_lr_anneal=True)
        for _ in range(2):
            g.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            lg = g.step()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / steps * 1e3
        enq_g = host_enqueue_ms(lambda: g.step())
        lg = {k: round(float(v), 5) for k, v in lg.items() if k != 'next_inner_lr'}
        g.close()
        graphed = {'ms_per_step': round(graph_ms, 3), 'value': round(x.shape[0] / graph_ms * 1e3, 3), 'host_enqueue_ms': enq_g, 'final_losses': lg,
                   'form': 'one hipGraph launch per step (harness.GraphedMldgStep; bit-identical to eager steps: tests/test_model_parity.py::test_graphed_mldg_step_is_bit_identical_to_eager)'}
        if enq > 0.8 * eager_ms:
            rec['eager'] = {k: rec[k] for k in ('ms_per_step', 'value', 'form', 'host_enqueue_ms', 'final_losses')}
            rec.update(graphed)
            rec['form_rule'] = 'host_enqueue_ms %.1f > 0.8 x eager ms_per_step %.1f: the hipGraph form is the one reported' % (enq, eager_ms)
        else:
            rec['graphed'] = graphed
            rec['form_rule'] = 'host_enqueue_ms %.1f <= 0.8 x eager ms_per_step %.1f: eager launches are GPU-bound and stay the reported form' % (enq, eager_ms)
    except Exception as e:      # noqa: BLE001
        rec['graphed'] = {'error': repr(e)}
    return rec


def physical_cores():
    """Physical cores of the host: unique (physical id, core id) pairs of /proc/cpuinfo, capped by the CPUs this process may run on."""
    try:
        pairs, phys = set(), None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                pairs.add((phys, line.split(':')[1].strip()))
        n = len(pairs) or os.cpu_count()
    except OSError:
        n = os.cpu_count()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return max(1, n)


def cpu_baseline(batch, size, steps=2):
    """The oracle (CPU restatement == imported reference, bit-exact) timed on this box's host cores, SURVEY 8(d), on a bounded sample of the
    workload (bs=2 of the bs=8 batch by default; --cpu-batch 8 runs the full batch, ~20 GB RSS). torch's CPU convolutions do not scale to every
    core of a two-socket host at this batch size (2 x EPYC 9575F: 0.49 img/s on 16 threads, 0.45 on 32, 0.28 on 64, 0.15 on all 128 physical cores --
    tools/cpu_threads_probe.py), so the thread count is chosen by measurement: one warm-up + one timed agg step at 16, 32 and all physical cores,
    then `steps` more timed steps at the fastest; `value` is the mean over the timed steps at that count and `cores` the threads it used."""
    import torch
    from oracle.ref_cpu import deeplab, harness
    from pinthememory_amd import synth
    phys = physical_cores()
    prev = torch.get_num_threads()
    cpu_model = ''
    try:
        cpu_model = next(l.split(':')[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name'))
    except (OSError, StopIteration):
        pass
    crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
    net = synth.load_det_weights(deeplab.DeepR50V3PlusD(synth.model_args(), 19, crit, crit))
    opt, _ = harness.make_optimizer(net)
    x, y = synth.make_batch(batch, size)

    def one():
        t0 = time.time()
        harness.agg_train_step(net, opt, x, y)
        return time.time() - t0
    sweep = {}
    for th in sorted({min(16, phys), min(32, phys), phys}):
        torch.set_num_threads(th)
        one()                                            # warm-up at this thread count (allocator, oneDNN primitive caches, thread pool)
        sweep[th] = one()
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = [sweep[best]] + [one() for _ in range(steps)]
    dt = sum(times) / len(times)
    # configs[0] -- the reference's own CPU-runnable case (1 x 3 x 256 x 256 eval forward) -- as the plumbing check, same thread count
    x1, _ = synth.make_batch(1, 256)
    net.eval()
    with torch.no_grad():
        net(x1)
        t0 = time.time()
        for _ in range(3):
            net(x1)
        c1_ms = (time.time() - t0) / 3 * 1e3
    torch.set_num_threads(prev)
    return {'value': batch / dt, 'unit': 'imgs/sec', 'cores': best, 'kind': 'port', 'cpu': cpu_model, 'physical_cores': phys,
            'thread_sweep_s_per_step': {str(k): round(v, 2) for k, v in sweep.items()}, 'config1_eval_forward_1x3x256x256_ms': round(c1_ms, 1),
            'sample': '%d timed agg train steps (fwd+bwd+SGD+memory-commit fwd) of a bs=%d sample of the bs=8 %dx%d fp32 workload, torch CPU oracle on %d threads '
                      '(the fastest of a one-step sweep over %s threads, each after its own warm-up step; %d physical cores), %.1f s per step'
                      % (len(times), batch, size, size, best, '/'.join(str(k) for k in sweep), phys, dt)}


def lib_stamp():
    """Content fingerprint of the library build (pinthememory_amd/build.py); tools/pmc_bench_summary.py writes the same value into its counter summaries."""
    try:
        return open(os.path.join(ROOT, 'pinthememory_amd', 'libpinmem_hip.so.stamp')).read().strip()
    except OSError:
        return None


def counter_file(pattern):
    """Latest committed PMC summary matching `pattern` (one tier, one workload: never another tier's file) -> (json, 'profiles/<name>', stale).
    Counters cannot be read from inside this process, so the committed figure is quoted with its source; `stale` says that the file was produced by
    ANOTHER build of the library than the one measured here (or predates the stamp field): the figure is then context, not a measurement of this code."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)),
                   key=lambda f: [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', os.path.basename(f))])
    if not files:
        return None, None, None
    j = json.load(open(files[-1]))
    stamp = lib_stamp()
    return j, 'profiles/' + os.path.basename(files[-1]), not (stamp and j.get('lib_stamp') == stamp)


PEAK_HBM_GBPS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)


def memory_path_roofline(batch, size):
    """HBM roofline of the memory path (north star: 'achieved HBM GB/s on the memory-read path'): every kernel of Memory_sup.read / write
    and the two fused up-sample + CE losses at the workload's shapes, each launch bracketed by its own pair of HIP events on the stream
    it is launched on (torch's current stream), algorithmic bytes per launch from SURVEY 8(d) / DESIGN section 4."""
    import glob
    import torch
    from pinthememory_amd import synth
    from pinthememory_amd.hip import kernels as K
    B, h, d, m, H = batch, size // 16, 256, 19, size
    N = B * h * h
    g = torch.Generator().manual_seed(304)
    x = torch.relu(torch.randn(B, h, h, d, generator=g)).cuda()
    mem = synth.det_memory().cuda()
    _, lab = synth.make_batch(B, H)
    lab = lab.cuda()
    qr, score, pmem = K.mem_read_fwd(x, mem)
    dqr, dsx = torch.randn(qr.shape, generator=g).cuda(), torch.randn(score.shape, generator=g).cuda()
    lg = score.view(B, h, h, m)
    lo = K.upsample_ce_fwd(lg, lab, 1.0)
    main = K.new((B, H // 4, H // 4, m), x, pitch_pad=True)
    main.copy_(torch.randn(B, H // 4, H // 4, m, generator=g).cuda())
    lo_main = K.upsample_ce_fwd(main, lab, 1.0)
    z = K.mem_write_accum(x, lab, m)
    lo_f, lo_field = K.upsample_ce_fwd_field(lg, lab, 1.0)
    lo_mf, lo_mfield = K.upsample_ce_fwd_field(main, lab, 1.0)
    cases = [     # (name, reference lines, algorithmic bytes per launch, launch)
        ('mem_read_fwd', 'memory.py:317-336', N * d * 4 + m * d * 4 + N * 2 * d * 4 + 2 * N * m * 4, lambda: K.mem_read_fwd(x, mem)),
        ('mem_read_bwd', 'memory.py:317-336 (autograd)', N * 2 * d * 4 + 2 * N * d * 4 + 2 * N * m * 4, lambda: K.mem_read_bwd(x, mem, pmem, dqr, dsx)),
        ('mem_read_fwd_with_p_query', 'memory.py:317-336 + :186 (the step\'s form: the read kernel leaves the column partials of the softmax over all queries, a second '
         'launch normalises)', N * d * 4 + m * d * 4 + N * 2 * d * 4 + 3 * N * m * 4, lambda: K.mem_read_fwd_pq(x, mem)),
        ('mem_colsoftmax', 'memory.py:186 (standalone two-launch form; not on the step since round 4)', 2 * N * m * 4, lambda: K.mem_colsoftmax(score)),
        ('mem_write_accum', 'memory.py:219-231', N * d * 4 + B * 4 * h * h * 8 + (m + 1) * (d + 1) * 4, lambda: K.mem_write_accum(x, lab, m)),
        ('readloss_fwd', 'memory.py:173-176', B * H * H * 8 + N * m * 4, lambda: K.upsample_ce_fwd(lg, lab, 1.0)),
        ('readloss_bwd', 'memory.py:173-176 (autograd)', B * H * H * 8 + 2 * N * m * 4, lambda: K.upsample_ce_bwd(lg, lab, lo, None, 1.0)),
        ('readloss_fwd_with_grad_field', 'memory.py:173-176 (training forward: loss + column-reduced gradient field in one sweep)',
         B * H * H * 8 + N * m * 4 + lo_field.numel() * 4, lambda: K.upsample_ce_fwd_field(lg, lab, 1.0)),
        ('readloss_bwd_from_field', 'memory.py:173-176 (autograd: row pass over the field)', lo_field.numel() * 4 + N * m * 4,
         lambda: K.upsample_ce_bwd_field(lg, (H, H), lo_f, lo_field, None, 1.0)),
        ('main_ce_fwd', 'deepv3plus.py:575-578', B * H * H * 8 + B * (H // 4) ** 2 * m * 4, lambda: K.upsample_ce_fwd(main, lab, 1.0)),
        ('main_ce_fwd_with_grad_field', 'deepv3plus.py:575-578 (training forward: loss + column-reduced gradient field in one sweep)',
         B * H * H * 8 + B * (H // 4) ** 2 * m * 4 + lo_mfield.numel() * 4, lambda: K.upsample_ce_fwd_field(main, lab, 1.0)),      # the field as the library sizes it
        ('main_ce_bwd_from_field', 'deepv3plus.py:575-578 (autograd: row pass over the field)', lo_mfield.numel() * 4 + B * (H // 4) ** 2 * m * 4,
         lambda: K.upsample_ce_bwd_field(main, (H, H), lo_mf, lo_mfield, None, 1.0)),
        ('main_ce_bwd', 'deepv3plus.py:575-578 (autograd)', B * H * H * 8 + 2 * B * (H // 4) ** 2 * m * 4, lambda: K.upsample_ce_bwd(main, lab, lo_main, None, 1.0)),
    ]
    del z
    counters, src, stale = {}, None, None
    try:
        cj, name, stale = counter_file('r*_memory_path_hbm_counters.json')
        counters = cj['kernels']          # {kernel symbol: {FETCH_SIZE_KB_per_launch, WRITE_SIZE_KB_per_launch, hbm_MB_per_launch}}
        src = name + ': rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/mem_probe.py; read = 2 x FETCH_SIZE (gfx950)'
    except (OSError, KeyError, ValueError, IndexError, TypeError):
        src = None
    rows = []
    fill_a = torch.empty(B, H // 4, H // 4, d, device=x.device)      # 302 MB at the flagship size: ~0.12 ms per copy
    fill_b = torch.empty_like(fill_a)
    reps = 20
    for name, ref, nbytes, fn in cases:
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        for _ in range(12):              # ~1.4 ms of queued copies: the host runs ahead, so the timed launches execute back to back
            K.copy(fill_a, fill_b)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        rows.append({'op': name, 'replaces': ref, 'algorithmic_MB': round(nbytes / 1e6, 2), 'launch_us': round(ms * 1e3, 1),
                     'achieved': round(nbytes / ms / 1e6, 1), 'frac': round(nbytes / ms / 1e6 / PEAK_HBM_GBPS, 4)})
    head = dict(rows[0])
    out = {'bound': 'hbm', 'kernel': 'mem_read_fwd (memory.py:317-336: normalise, q.M^T, softmax over slots, P.M, concat) at %d queries x 256 x 19 slots' % N,
           'achieved': head['achieved'], 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': head['frac'], 'launch_us': head['launch_us'],
           'algorithmic_MB': head['algorithmic_MB'],
           'traffic': next((round(v['hbm_MB_per_launch'] * 1e6) for k, v in counters.items() if k.startswith('mem_read_fwd')), None), 'traffic_source': src,
           'traffic_stale': stale,
           'measured': 'after the timed region: %d back-to-back launches between one pair of HIP events on the launch stream, queued behind ~1.4 ms of copies so '
                       'that the host is ahead of the GPU (an event pair per launch adds ~5 us of launch latency to a 20 us kernel); average per launch incl. '
                       'the gaps between kernels; one launch may include the second-stage kernels of the op (fixed-order reductions)' % reps,
           'all_memory_path_ops': rows}
    return out


PREC_NAMES = ('v_mfma_f32_32x32x2_f32', 'v_mfma_f32_32x32x16_bf16 on fp32 tiles rounded per fragment', 'v_mfma_f32_32x32x16_bf16 on bf16 tiles (bf16 operands in HBM)',
              'v_mfma_f32_32x32x16_bf16 on bf16 LDS tiles rounded from gathered fp32 rows, ds_read_b64_tr_b16 fragments',
              'v_mfma_f32_32x32x16_bf16 on bf16 LDS tiles gathered from bf16 activations, ds_read_b64_tr_b16 fragments',
              'v_mfma_f32_32x32x16_bf16 on bf16 LDS tiles filled by global_load_lds (LDS-DMA), source-side XOR swizzle')


DOMINANT = {'split': False, 'bytes': 0.0}      # set by dominant_conv_kernel: the dominant kernel of the fp32 tier is a split-operand instantiation (bf16 matrix pipe)


def dominant_conv_kernel(K, bf16):
    """The convolution-kernel instantiation with the largest total time among the launches recorded since the last clear (in-library HIP events, pm_profile_*):
    -> ((ms, flops, launches), symbol, description) or None. conv_igemm_kernel<mode, bm, bn, wm, wn, km, prec, nst> (csrc/conv_igemm.hip) and, on the bf16 tier,
    conv16_kernel<bm, bn, wm, wn, nst> (csrc/conv16.hip, recorded as mode 4 / prec 5)."""
    best = None
    for mode in ((0, 1, 2) if not bf16 else (0, 1, 2, 4, 5)):
        for bm in (256, 128, 64):
            for bn in (256, 128, 64, 32):
                for km in (0, 1, 2, 3):      # 3 = K_PW: the split path's pointwise K-state
                    for nst in (3, 2, 1):
                        for prec in ((0, 5) if not bf16 else ((5,) if mode >= 4 else (2, 4, 3, 1))):      # fp32 tier: 0 = fp32 MFMA, 5 = split operands on the bf16 pipe
                            r = K.profile_read(mode=mode, bm=bm, bn=bn, km=km, nst=nst, prec=prec)
                            if r[2] and (best is None or r[0] > best[0][0]):
                                best = (r, (mode, bm, bn, km, nst), prec)
                                DOMINANT['bytes'] = K.profile_read_bytes(mode=mode, bm=bm, bn=bn, km=km, nst=nst, prec=prec)
    if best is None:
        return None
    r, (mode, bm, bn, km, nst), kprec = best
    DOMINANT['split'] = (not bf16) and kprec == 5
    if DOMINANT['split']:
        sym = 'conv_igemm_kernel<%d, %d, %d, 2, 2, %d, 5, %d>' % (mode, bm, bn, km, nst)
        what = ('%s, %dx%dx32 tile, %s K-state, %s LDS; fp32 operands, 3-way exact bf16 split (hi + mid + lo == x) as the gathered rows are stored to LDS, 6 cross products on '
                'v_mfma_f32_32x32x16_bf16, fp32 accumulate (csrc/conv_split.hip); direct convolutions and the batched Winograd F(4x4,3x3) / F(2x2,3x3) GEMMs; '
                'FLOPs = 2*M*N*K of the fp32 problem' % (('forward', 'data gradient', 'weight gradient')[mode], bm, bn, ('wave-uniform', 'per-lane', 'per-lane looping', 'wave-uniform pointwise (row validity folded into the base offset)')[km],
                                                        'double-buffered' if nst == 2 else 'single-stage'))
        return r, sym, what
    if mode == 5 and km == 1:
        sym = 'conv16p_kernel<%d, %d, %s, 4>' % (bm, bn, '4, 2' if bm == 256 else '2, 4')
        what = ('forward / stride-1 data gradient on bf16 activations, persistent form: one block per CU walks %dx%dx64 tiles, four producer waves keep a three-stage LDS '
                'ring filled across tile boundaries (counted waits), eight waves multiply, %s; FLOPs = 2*M*N*K executed' % (bm, bn, PREC_NAMES[5]))
    elif mode == 5:
        sym = 'conv16w_kernel<%d, %d, %s>' % (bm, bn, '4, 2, 3' if (bm, bn) == (256, 128) else ('2, 4, 3' if (bm, bn) == (128, 256) else '2, 4, 2'))
        what = ('forward / stride-1 data gradient on bf16 activations, wide form: %dx%dx64 tile, eight waves, one block per CU, %s, %s; '
                'FLOPs = 2*M*N*K executed' % (bm, bn, 'two LDS stages' if (bm, bn) == (256, 256) else 'three-stage LDS ring with counted waits', PREC_NAMES[5]))
    elif mode == 4:
        sym = 'conv16_kernel<%d, %d, 2, 2, %d>' % (bm, bn, nst)
        what = 'forward / stride-1 data gradient on bf16 activations, %dx%dx64 tile, %s, %s; FLOPs = 2*M*N*K executed' % (
            bm, bn, 'two LDS stages' if nst == 2 else 'one LDS stage (single-K-step reductions)', PREC_NAMES[5])
    elif mode == 2 and nst == 3:
        sym = 'wgrad16_kernel<%d, %d, %s>' % (bm, bn, '4, 2' if bm == 256 else '2, 4')
        what = ('weight gradient on bf16 activations, persistent form: one block per CU walks (%dx%d tile, 64-pixel steps of a pixel range) units, four producer waves keep '
                'a three-stage LDS ring of pixel-major dy / x tiles filled by global_load_lds, fragments through ds_read_b64_tr_b16, fp32 split-K slabs reduced in fixed '
                'order; FLOPs = 2*M*N*K' % (bm, bn))
    else:
        sym = 'conv_igemm_kernel<%d, %d, %d, %s, %d, %d, %d>' % (mode, bm, bn, '4, 1' if bn == 32 else '2, 2', km, kprec, nst)
        what = '%s, %dx%dx32 tile, %s K-state, %s LDS, %s; direct convolutions and the batched Winograd F(4x4,3x3) / F(2x2,3x3) GEMMs; FLOPs = 2*M*N*K executed' % (
            ('forward', 'data gradient', 'weight gradient')[mode], bm, bn, ('wave-uniform', 'per-lane', 'per-lane looping')[km],
            'double-buffered' if nst == 2 else 'single-stage', PREC_NAMES[kprec])
    return r, sym, what


def host_enqueue_ms(step_fn, tries=2):
    """Host time to enqueue one step from an idle GPU (tools/cpu_enqueue_time.py's method): synchronize, call, stop the clock when the call returns (the GPU is still
    working), synchronize again. The minimum over `tries`. A step is GPU-bound while this stays below its wall time; above ~0.8 x the launch path is the bound."""
    import torch
    best = None
    for _ in range(tries):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step_fn()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        best = dt if best is None else min(best, dt)
    return round(best * 1e3, 3)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, child=None, env=None, timeout=None):
    """`python bench.py --gpus N` without a launcher environment: start N fresh rank processes (one per GPU, the reference's
    `torch.distributed.launch --nproc_per_node N` hop, train.py:63-76 reads RANK / WORLD_SIZE / LOCAL_RANK the same way), relay
    rank 0's JSON line and return the worst child exit code. This process has not touched the GPU (no torch import, no HIP call)
    and never re-execs: the ranks are children. `child` = the command to run per rank (tests stub it)."""
    import subprocess
    import threading
    base = dict(os.environ if env is None else env)
    base.update({'WORLD_SIZE': str(n), 'LOCAL_WORLD_SIZE': str(n), 'MASTER_ADDR': '127.0.0.1',
                 'MASTER_PORT': base.get('MASTER_PORT') or str(free_port()), 'PM_BENCH_LAUNCHED': '1'})
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = list(child) if child is not None else [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout carries the JSON line: captured and relayed; the other ranks print nothing on stdout by contract
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    lines = []

    def pump():
        for line in procs[0].stdout:
            lines.append(line)
            sys.stdout.write(line)
            sys.stdout.flush()
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    # a rank that dies leaves the others in a collective: once one has failed, give the rest a grace period, then end them
    import time as _t
    rcs = [None] * n
    failed_at = None
    t_start = _t.time()
    while any(rc is None for rc in rcs):
        if timeout is not None and _t.time() - t_start > timeout:      # a rank that hangs without exiting (a collective nobody answers) ends the whole run
            print('bench.py: ranks still running after %.0f s -- killing them' % timeout, file=sys.stderr)
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.kill()
                    p.wait()
                    rcs[i] = 124
            break
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
                if rcs[i] not in (None, 0) and failed_at is None:
                    failed_at = _t.time()
        if failed_at is not None and _t.time() - failed_at > 30:
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.kill()
                    rcs[i] = p.wait() or 1
        _t.sleep(0.05)
    t.join(timeout=10)
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst == 0 and not any(l.lstrip().startswith('{') for l in lines):
        print('bench.py: rank 0 printed no JSON line', file=sys.stderr)
        worst = 1
    return min(worst, 255)


def main():
    a = parse()
    if a.workload == 'config5':
        return config5(a)
    if a.workload == 'meminit':
        return meminit(a)
    if a.workload == 'mldg':
        return mldg(a)
    if a.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:], timeout=a.launch_timeout))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert a.gpus == world, 'bench.py --gpus %d but the launcher environment says WORLD_SIZE=%d' % (a.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the HIP path has no CPU fallback'
    # one process per GPU; PM_BENCH_BACKEND=gloo lets two ranks share one GPU to rehearse the N > 1 code path on a 1-GPU box
    backend = os.environ.get('PM_BENCH_BACKEND', 'nccl')
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    multi = world > 1 or os.environ.get('PM_DIST_FORCE', '0') == '1'      # PM_DIST_FORCE: one-rank RCCL rehearsal (pinthememory_amd/dist.py)
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device('cuda', local)

    from pinthememory_amd import dist as D, harness, synth
    from pinthememory_amd.hip import kernels as K
    from pinthememory_amd.network import deepv3plus, mynn
    crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
    K.set_conv_precision(a.dtype)
    bf16 = a.dtype != 'f32'
    peak = PEAK_TFLOPS_BF16_MFMA if bf16 else PEAK_TFLOPS_F32_MFMA
    if multi:
        mynn.set_bnfunc(torch.nn.SyncBatchNorm)        # train.py:95 converts to SyncBN under DDP
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).to(dev)
    if multi:                                          # train.py:95: every BatchNorm, incl. the two inside Memory_sup (memory.py:76,105)
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
        assert not any(type(m) is torch.nn.BatchNorm2d for m in net.modules())
    opt, sched = harness.make_optimizer(net)
    buckets = D.GradBuckets(net.parameters()) if multi else None
    x, y = synth.make_batch(a.batch, a.size, seed=304 + rank)          # rank r: its own 8 images (config 4)
    x, y = x.to(dev), y.to(dev)

    edge = None
    if a.input_edge:
        from pinthememory_amd import input_edge
        domains = 2 if a.batch % 2 == 0 else 1                                 # e.g. gtav + synthia (datasets/multi_loader.py:81-102)
        src = input_edge.SyntheticDomainSource(a.batch // domains, domains, a.size, n_buffers=3, seed=304 + rank, static=True)
        pf = input_edge.DevicePrefetcher(src, depth=1, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            pf.next()
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t0) / 5 * 1e3
        edge = {'source': 'synthetic uint8 [%d, %d, %d, %d, 3] + uint8 labels, pinned host ring of 3, side-stream H2D + u8->NHWC4/int64 kernels, depth 1' % (
                    a.batch // domains, domains, a.size, a.size),
                'h2d_bytes_per_step': src.bytes_per_batch(), 'edge_alone_ms': round(h2d_ms, 3), 'pcie_GBps_alone': round(src.bytes_per_batch() / h2d_ms / 1e6, 2),
                'fp32_int64_bytes_the_reference_ships': src.bytes_per_batch() // 4 * (12 + 8)}

    def step():
        if edge is not None:
            xb, yb = pf.next()
            return harness.agg_train_step(net, opt, xb, yb, sched=sched, buckets=buckets, truncate_second_forward=a.truncate_second_forward)
        return harness.agg_train_step(net, opt, x, y, sched=sched, buckets=buckets, truncate_second_forward=a.truncate_second_forward)

    for _ in range(a.warmup):
        step()
    prof = not a.no_profile
    from pinthememory_amd.hip import ops as _ops
    overlap_defaults = (_ops.OVERLAP_WGRAD, harness.COMMIT_OVERLAP)
    graphed = None
    timed_step = step
    if a.graph:
        assert not multi and edge is None, '--graph: single process, resident batch'
        graphed = harness.GraphedAggStep(net, opt, x, y, sched=sched, warmup=0 if a.warmup else 2, pipelined=True)
        timed_step = lambda: graphed.step(x, y)
        for _ in range(2):
            timed_step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = timed_step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # host-side cost of one step (two extra untimed steps from an idle GPU). Skipped with --no-profile: the counter / trace passes of tools/gpu_round*_profiles.sh count the
    # steps of the run (warm-up + timed) and must not see extra ones
    host_ms = {'timed_form': None if a.no_profile else host_enqueue_ms(timed_step)}
    if graphed is not None:
        losses = {k: v.clone() for k, v in losses.items()}
        graphed.close()
        graphed = None
        host_ms['eager'] = None if a.no_profile else host_enqueue_ms(step)
    per_rank_ms, ranks_seen = [round(dt / a.steps * 1e3, 3)], 1
    if multi:
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = dt
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(v / a.steps * 1e3, 3) for v in t.tolist()]
        dt = t.max().item()                              # the contract's MAX over ranks
        from pinthememory_amd import rccl as _rccl
        comm = _rccl.get(None)                           # the direct communicator every SyncBN / memory / gradient exchange of the step used
        ranks_seen = comm.count() if comm is not None else dist.get_world_size()
    # how the N > 1 exchanges travelled: the direct RCCL communicator on the compute stream, or torch.distributed's process group (and why)
    rccl_direct, rccl_reason, coll_per_step = None, 'single process: no collectives', 0
    if multi:
        rccl_direct = comm is not None
        rccl_reason = ('direct ncclComm on the compute stream (pinthememory_amd/rccl.py), %d ranks' % ranks_seen) if rccl_direct else D.direct_fallback_reason(backend)
        coll_per_step = D.count_collectives(lambda: step())
    roof = None
    if prof:
        # Per-kernel timing brackets every conv launch with two HIP events, which costs ~1.7 ms per step (measured: 70.2 vs 68.5 ms),
        # so it never runs inside the timed region: the numbers come from extra, untimed steps right after it.
        ov_steps = 2
        K.profile_enable(True)
        for _ in range(ov_steps):
            step()
        torch.cuda.synchronize()
        K.profile_enable(False)
        ov_ms, ov_fl, ov_n = K.profile_read(clear=True)        # as in the timed region: weight gradients overlapped on the side stream
        # Per-kernel durations are only meaningful when kernels do not share the GPU: repeat two steps, untimed, with the
        # weight-gradient side stream off (every launch serialised on one stream) and take the roofline numbers from those.
        from pinthememory_amd.hip import ops as _ops
        prof_steps = 2
        _ops.OVERLAP_WGRAD = False
        harness.COMMIT_OVERLAP = False             # ... and the commit forward back on the main stream, behind the SGD
        torch.cuda.synchronize()
        K.profile_enable(True)
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize()
        K.profile_enable(False)
        if os.environ.get('PM_PROFILE_DUMP'):
            K.profile_dump(os.environ['PM_PROFILE_DUMP'])
        # dominant kernel = the convolution-kernel instantiation with the largest total time in the serialised pass
        best = dominant_conv_kernel(K, bf16)
        tot_ms, tot_fl, tot_n = K.profile_read(clear=True)
        if best:
            (ms, fl, n), sym, what = best
            ach = fl / (ms * 1e-3) / 1e12
            traffic, traffic_src = None, None      # HBM bytes per launch of that symbol, from the committed PMC passes of this command
            traffic_stale = None
            try:
                cj, name, traffic_stale = counter_file('r*_bench_1gpu%s_hbm_counters.json' % ('_bf16' if bf16 else ''))      # this tier's latest committed PMC passes
                for kr in cj['kernels']:
                    if kr['kernel'].replace('void ', '').replace(', false>', '>').strip() == sym:      # the symbol's trailing STATS=false template argument
                        traffic = round((kr['read_MB_per_launch'] + kr['write_MB_per_launch']) * 1e6)
                        traffic_src = (name + ': rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of '
                                       'bench.py --steps 1 --warmup 1; read = 2 x FETCH_SIZE (gfx950), average over all launches of the symbol')
            except (OSError, KeyError, ValueError, IndexError, TypeError):
                pass
            split_dom = DOMINANT['split']
            # a split-operand kernel executes SIX bf16 MFMA products per fp32 multiply-add: `achieved` is what the matrix pipe really ran (6 x 2*M*N*K) against the bf16
            # peak; `effective_fp32_tflops` is the fp32 work it delivered (to set beside the 157.3 TF of v_mfma_f32_32x32x2_f32)
            roof = {'bound': 'mfma', 'kernel': '%s (%s)' % (sym, what), 'achieved': round(ach * (6 if split_dom else 1), 2),
                    'peak': PEAK_TFLOPS_BF16_MFMA if split_dom else peak, 'unit': 'TFLOP/s', 'frac': round(ach * (6 if split_dom else 1) / (PEAK_TFLOPS_BF16_MFMA if split_dom else peak), 4),
                    'effective_fp32_tflops': round(ach, 2), 'effective_vs_fp32_mfma_peak': round(ach / PEAK_TFLOPS_F32_MFMA, 4),
                    'traffic': traffic, 'traffic_source': traffic_src,
                    'traffic_stale': traffic_stale,
                    'algorithmic_bytes_per_launch': round(DOMINANT['bytes'] / n) if DOMINANT.get('bytes') else None,
                    'traffic_over_algorithmic': round(traffic / (DOMINANT['bytes'] / n), 3) if (traffic and DOMINANT.get('bytes')) else None,
                    'algorithmic_bytes_note': 'both operands once + the output (every slab of a split-K launch), averaged over the launches of the symbol in the serialised pass',
                    'achieved_GBps_algorithmic': round(DOMINANT['bytes'] / n / (ms / n * 1e-3) / 1e9, 1) if DOMINANT.get('bytes') else None,
                    'launches_per_step': n / prof_steps, 'avg_launch_ms': round(ms / n, 5), 'gflop_per_launch': round(fl / n / 1e9, 3),
                    'measured': '%d extra steps after the timed region (event timing costs ~1.7 ms/step, so the timed region runs without it), all launches '
                                'serialised on one stream (with the weight gradients on their side stream and the commit forward on its own every concurrent kernel\'s duration inflates)' % prof_steps,
                    'all_conv_kernels': {'achieved': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), 'ms_per_step': round(tot_ms / prof_steps, 3),
                                         'launches_per_step': tot_n / prof_steps,
                                         'timed_region_overlapped': {'achieved': round(ov_fl / (ov_ms * 1e-3) / 1e12, 2), 'ms_per_step': round(ov_ms / ov_steps, 3),
                                                                     'measured': '%d untimed steps with the stream overlaps of the timed region (fp32: weight gradients on a side stream; bf16: inline; commit forward of step t under the training forward of step t + 1)' % ov_steps}}}
    side = None
    if a.dtype == 'f32' and not multi and not a.no_side and edge is None:
        # BASELINE configs[2] in the driver's own run: the SAME process, model and batch switched to the bf16 tier after the timed region -- a few untimed steps, a
        # few timed ones (barrier-free single GPU: synchronize on both sides), then two serialised steps for its dominant kernel. Never part of `value`.
        # Any failure in here is recorded in the line, never raised: the fp32 metric has been measured and must be printed.
        _ops.OVERLAP_WGRAD, harness.COMMIT_OVERLAP = overlap_defaults
        K.set_conv_precision('bf16')
        side = {}
        try:
            s_warm, s_steps = 3, 5
            for _ in range(s_warm):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(s_steps):
                l16 = step()
            torch.cuda.synchronize()
            dt16 = time.perf_counter() - t1
            eager_ms = dt16 / s_steps * 1e3
            enq16 = host_enqueue_ms(step)
            what16 = ('configs[2]: the same model, batch and agg train step on the bf16 tier (bf16 activations and activation gradients between layers, '
                      'bf16-MFMA convolutions with fp32 accumulation, fp32 statistics / losses / memory / parameters)')
            side['bf16'] = {'workload': what16, 'ms_per_step': round(eager_ms, 3), 'value': round(a.batch * s_steps / dt16, 3), 'unit': 'imgs/sec', 'steps': s_steps,
                            'warmup': s_warm, 'final_loss': round(float(l16['total']), 5), 'dtype': 'bf16', 'form': 'eager launches (commit forward of step t on its own stream under the training forward of step t + 1)',
                            'host_enqueue_ms': enq16, 'measured': 'after the timed fp32 region of this run, same process'}
            if prof:
                _ops.OVERLAP_WGRAD, harness.COMMIT_OVERLAP = False, False
                torch.cuda.synchronize()
                K.profile_enable(True)
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                K.profile_enable(False)
                if os.environ.get('PM_PROFILE_DUMP'):
                    K.profile_dump(os.environ['PM_PROFILE_DUMP'] + '.bf16')
                b16 = dominant_conv_kernel(K, True)
                t16_ms, t16_fl, t16_n = K.profile_read(clear=True)
                _ops.OVERLAP_WGRAD, harness.COMMIT_OVERLAP = overlap_defaults
                if b16:
                    (ms, fl, n), sym, what = b16
                    side['bf16']['roofline'] = {'bound': 'mfma', 'kernel': '%s (%s)' % (sym, what), 'achieved': round(fl / (ms * 1e-3) / 1e12, 2), 'peak': PEAK_TFLOPS_BF16_MFMA,
                                                'unit': 'TFLOP/s', 'frac': round(fl / (ms * 1e-3) / 1e12 / PEAK_TFLOPS_BF16_MFMA, 4), 'launches_per_step': n / 2,
                                                'all_conv_kernels': {'achieved': round(t16_fl / (t16_ms * 1e-3) / 1e12, 2), 'ms_per_step': round(t16_ms / 2, 3)}}
            # the same step as ONE hipGraph launch (harness.GraphedAggStep, pipelined: the commit forward of step t - 1 beside the training forward of step t inside the graph)
            try:
                g16 = harness.GraphedAggStep(net, opt, x, y, sched=sched, warmup=0, pipelined=True)
                for _ in range(2):
                    g16.step(x, y)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(s_steps):
                    lg16 = g16.step(x, y)
                torch.cuda.synchronize()
                dtg = time.perf_counter() - t1
                enq_g = host_enqueue_ms(lambda: g16.step(x, y))
                final_g = round(float(lg16['total']), 5)
                g16.close()
                side['bf16_graphed'] = {'workload': what16, 'form': 'one hipGraph launch per step (harness.GraphedAggStep, pipelined: [commit forward of step t - 1 || training forward '
                                        'of step t] + backward + SGD; bit-identical to eager steps: tests/test_model_parity.py::test_graphed_agg_step_is_bit_identical_to_eager)',
                                        'ms_per_step': round(dtg / s_steps * 1e3, 3), 'value': round(a.batch * s_steps / dtg, 3), 'unit': 'imgs/sec', 'steps': s_steps, 'warmup': 2,
                                        'final_loss': final_g, 'dtype': 'bf16', 'host_enqueue_ms': enq_g}
                # when the eager step's host enqueue exceeds 0.8 x its wall time the launch path is (about to be) the bound: the captured step is then the form side.bf16 reports
                if enq16 > 0.8 * eager_ms:
                    eager_rec = {k: side['bf16'][k] for k in ('ms_per_step', 'value', 'form', 'host_enqueue_ms', 'final_loss')}
                    side['bf16'].update({k: side['bf16_graphed'][k] for k in ('ms_per_step', 'value', 'form', 'host_enqueue_ms', 'final_loss')})
                    side['bf16']['eager'] = eager_rec
                    side['bf16']['form_rule'] = 'host_enqueue_ms %.1f > 0.8 x eager ms_per_step %.1f: the hipGraph form is the one reported' % (enq16, eager_ms)
                else:
                    side['bf16']['form_rule'] = 'host_enqueue_ms %.1f <= 0.8 x eager ms_per_step %.1f: eager launches are GPU-bound and stay the reported form' % (enq16, eager_ms)
            except Exception as e:      # noqa: BLE001
                side['bf16_graphed'] = {'error': repr(e)}
        except Exception as e:      # noqa: BLE001 -- the fp32 line is always emitted
            side.setdefault('bf16', {})['error'] = repr(e)
        finally:
            K.set_conv_precision('f32')
            _ops.OVERLAP_WGRAD, harness.COMMIT_OVERLAP = overlap_defaults
        # the regime every pinmem script runs, on both tiers, eager and captured (never part of `value`; failures are recorded, not raised)
        for tier, key in (('f32', 'mldg'), ('bf16', 'mldg_bf16')):
            try:
                harness.finish_commit(net)
                torch.cuda.synchronize()
                side[key] = mldg_side(torch, harness, K, net, opt, sched, x, y, tier)
            except Exception as e:      # noqa: BLE001
                side[key] = {'error': repr(e)}
            finally:
                K.set_conv_precision('f32')
    if rank == 0:
        imgs = a.batch * world * a.steps
        gf_img = STEP_GFLOP_PER_IMG * (a.size / 768.0) ** 2      # conv FLOPs scale with the pixel count
        out = {'metric': 'train imgs/sec 768x768 bs=8 R50-DeepLabV3+ +mem', 'value': round(imgs / dt, 3), 'unit': 'imgs/sec', 'n_gpus': world,
               'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
               'config': {'workload': '%s: ResNet-50 DeepLabV3+ + memory, bs=%d/GPU %dx%d synthetic, %s reference-faithful agg train step '
                                      '(fwd + bwd + SGD + eval-mode memory-commit fwd%s)' % ('configs[2]' if bf16 else 'configs[1]', a.batch, a.size, a.size,
                                                                                            {'bf16': 'bf16 tier: bf16 activations and activation gradients between layers, bf16-MFMA convolutions with fp32 accumulation, fp32 statistics / losses / memory / parameters',
                                                                                             'bf16_operands': 'bf16-MFMA convolutions (bf16 operands in HBM and LDS, fp32 accumulation; fp32 activations between layers)',
                                                                                             'bf16_staged': 'bf16-MFMA (fp32 tiles staged, rounded per fragment)',
                                                                                             'f32': 'fp32 storage / accumulation / results; the contractions run on the bf16 matrix pipe: fp32 operands, 3-way exact bf16 split, 6 products, fp32 accumulate (the pointwise 64- / 128-channel reductions on the 192^2 / 96^2 maps in the wave-streamed form of csrc/pwstream.hip; other reductions <= 128 and outputs < 64 columns stay on v_mfma_f32_32x32x2_f32)' if os.environ.get('PM_SPLIT', '1') != '0' else 'fp32'}[a.dtype],
                                                                                            ', decoder skipped in the commit fwd' if a.truncate_second_forward else ''),
                          'global_batch': a.batch * world, 'crop': a.size, 'parallelism': 'dp%d' % world,
                          'ranks_seen': ranks_seen, 'ranks_seen_source': ('ncclCommCount of the direct RCCL communicator' if multi and backend == 'nccl' and ranks_seen == world and _rccl.get(None) is not None
                                                                           else ('torch.distributed world size (%s)' % backend if multi else 'single process')),
                          'ms_per_step_per_rank': per_rank_ms,
                          'step_form': ('one hipGraph launch per step (harness.GraphedAggStep pipelined; --graph)' if a.graph else 'eager launches'),
                          'host_enqueue_ms': host_ms['timed_form'], 'host_enqueue_ms_eager': host_ms.get('eager', host_ms['timed_form']),
                          'host_enqueue_note': 'host time to enqueue one step from an idle GPU (min of 2); GPU-bound while below ms_per_step',
                          'rccl_direct': rccl_direct, 'rccl_direct_reason': rccl_reason,
                          'collectives_per_step': coll_per_step,
                          'conv_tflop_per_step': round(gf_img * a.batch / 1e3, 3),
                          'conv_flop_convention': 'direct-algorithm FLOPs (SURVEY 8d); the Winograd F(4x4,3x3) / F(2x2,3x3) layers execute 4x / 2.25x fewer on the MFMA, so step_mfma_frac is a direct-equivalent rate, not MFMA utilisation',
                          'step_mfma_frac': round(gf_img * a.batch * world * a.steps / 1e3 / dt / (peak * world), 4),
                          'final_loss': round(float(losses['total']), 5)},
               'roofline': roof}
        if side is not None:
            out['side'] = side
        if edge is not None:
            out['config']['input_edge'] = edge
            out['data'] = 'synthetic, fresh uint8 batch from pinned host memory every step (PCIe-inclusive side measurement)'
        if roof is not None:
            # the whole step against both rooflines, on EXECUTED FLOPs (2*M*N*K of every launch: the Winograd layers count their reduced work)
            # and on the fabric traffic of the committed counter passes -- step_mfma_frac above is a direct-algorithm-equivalent rate, not utilisation
            ex_tf = tot_fl / prof_steps / 1e12
            step = {'executed_tflop': round(ex_tf, 3), 'mfma_frac_executed': round(ex_tf / (dt / a.steps) / peak, 4),
                    'hbm_GB': None, 'hbm_frac': None, 'hbm_source': None, 'hbm_stale': None}
            try:
                cj, name, hstale = counter_file('r*_bench_1gpu%s_hbm_counters.json' % ('_bf16' if bf16 else ''))
                gb = (cj['total_read_GB'] + cj['total_write_GB']) / float(cj.get('steps_counted', 2))
                step.update(hbm_GB=round(gb, 1), hbm_frac=round(gb / (dt / a.steps) / PEAK_HBM_GBPS, 4), hbm_stale=hstale,
                            hbm_source=name + ' (FETCH_SIZE x 2 + WRITE_SIZE over all kernels of the counted steps, separate --pmc passes), '
                                       'divided by this run\'s step time and the 8 TB/s peak; hbm_stale = the counter file comes from another build of the library')
            except (OSError, KeyError, ValueError, IndexError, TypeError):
                pass
            roof['step'] = step
            try:
                roof['memory_read'] = memory_path_roofline(a.batch, a.size)
            except Exception as e:      # noqa: BLE001 -- the metric line must not depend on the side measurement
                roof['memory_read'] = {'error': repr(e)}
        if not a.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(a.cpu_batch, a.size)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if multi:
        from pinthememory_amd import rccl
        torch.cuda.synchronize()
        dist.barrier()          # rank 0 is still producing its side measurements: nobody tears a communicator down under it
        torch.cuda.synchronize()
        rccl.shutdown()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
