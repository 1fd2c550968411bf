"""Synthetic Cityscapes-shaped inputs and RNG-free deterministic weights.

Used by bench.py, the parity tests and the golden-fixture generator so that the imported
reference (build container only), the CPU oracle and the HIP path all see identical bits.
Input recipe: SURVEY.md section 8(d); the reference's own seed is 304 (/root/reference/config.py:52).
"""
import math
import types

import torch

IGNORE = 255                                          # /root/reference/datasets/__init__.py:26
NUM_CLASSES = 19                                      # /root/reference/datasets/__init__.py:25


def model_args(**over):
    """The 10 model flags the reference's networks read (deepv3plus.py:314-317,459-472,580)."""
    a = dict(wt_layer=[0] * 7, relax_denom=0.0, clusters=50, memory=True, mem_slot=19, mem_dim=256,
             mem_momentum=0.8, mem_temp=1, gumbel_off=True, use_wtloss=False)
    a.update(over)
    return types.SimpleNamespace(**a)


def _hash_uniform(n, key):
    """Counter hash -> uniform(-0.5, 0.5) float64; pure int64 arithmetic, identical on every platform."""
    i = torch.arange(n, dtype=torch.int64)
    x = (i * 2654435761 + (key + 1) * 40503) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return x.to(torch.float64) / 4294967296.0 - 0.5


def det_tensor(shape, key, std=1.0, mean=0.0):
    n = 1
    for s in shape:
        n *= s
    return (_hash_uniform(n, key) * (std * math.sqrt(12.0)) + mean).to(torch.float32).reshape(shape)


RESIDUAL_GAIN = 0.1      # gamma of the last BatchNorm of every bottleneck (bn3) is centred here instead of 1


def det_state_dict(net, gain=1.0, residual_gain=RESIDUAL_GAIN):
    """Deterministic values for every state_dict entry of `net` (never committed: regenerate anywhere).

    conv/linear weights ~ U with std sqrt(2/fan_in)*gain (kaiming-like, keeps activations O(1));
    BN gamma in [0.8,1.2], beta small, running_mean small, running_var in [0.75,1.25]; biases small.
    The residual branches are damped (`bn3.weight` ~ residual_gain * [0.8,1.2], as in a trained ResNet): with gamma ~ 1 on all
    three BatchNorms of 16 stacked bottlenecks a train-mode network is chaotic at initialisation -- an fp32 round-off of 1e-7
    grows 1.35x per block to 4e-4 at the logits, ReLU masks flip, trunk gradients reach O(500) and the reference's OWN fp32
    gradients are 35 % (median, bs=2 128^2) off an fp64 run of the same code. Damped, the forward error stays at 5e-6 and the
    fp32 gradients within ~1e-3 of fp64 -- the floor left is sqrt(2 * p(0) * eps_fwd): ReLU units within round-off of zero that
    take the other branch. Measured with tools/grad_conditioning.py (CPU, oracle only).
    """
    out = {}
    for idx, (name, v) in enumerate(net.state_dict().items()):
        if name.endswith('num_batches_tracked'):
            out[name] = torch.zeros_like(v)
        elif name.endswith('running_var'):
            out[name] = det_tensor(v.shape, idx, std=0.25 / math.sqrt(3.0), mean=1.0).clamp_min(0.5)
        elif name.endswith('running_mean'):
            out[name] = det_tensor(v.shape, idx, std=0.05)
        elif v.dim() == 1 and name.endswith('weight'):
            out[name] = det_tensor(v.shape, idx, std=0.2 / math.sqrt(3.0), mean=1.0) * (residual_gain if name.endswith('bn3.weight') else 1.0)
        elif v.dim() == 1:
            out[name] = det_tensor(v.shape, idx, std=0.02)
        else:
            fan_in = v[0].numel()
            out[name] = det_tensor(v.shape, idx, std=gain * math.sqrt(2.0 / fan_in))
    return out


def det_memory(slots=19, dim=256, key=7777):
    return torch.nn.functional.normalize(det_tensor((slots, dim), key), dim=1)


def load_det_weights(net, gain=1.0):
    net.load_state_dict(det_state_dict(net, gain))
    if getattr(net, 'memory', None) is not None:
        m = det_memory(net.memory.memory_size, net.memory.feature_dim)
        net.memory.m_items = m.to(net.memory.m_items.device) if torch.is_tensor(net.memory.m_items) else m
    return net


def make_batch(batch, size, seed=304, block=64, classes=NUM_CLASSES, ignore_rows=None, ignore_frac=0.01):
    """images randn(B,3,S,S) fp32; labels = block x block patches of randint(0,19), the top rows and
    ~1 % random pixels set to 255; int64. `size` may be an int or (H, W)."""
    h, w = (size, size) if isinstance(size, int) else size
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, h, w, generator=g, dtype=torch.float32)
    block = max(1, min(block, h // 4, w // 4))
    gh, gw = (h + block - 1) // block, (w + block - 1) // block
    coarse = torch.randint(0, classes, (batch, gh, gw), generator=g, dtype=torch.int64)
    y = coarse.repeat_interleave(block, 1).repeat_interleave(block, 2)[:, :h, :w].contiguous()
    rows = h // 16 if ignore_rows is None else ignore_rows
    y[:, :rows] = IGNORE
    y[torch.rand(batch, h, w, generator=g) < ignore_frac] = IGNORE
    return x, y
