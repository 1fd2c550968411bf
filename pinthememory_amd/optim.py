"""SGD with momentum and weight decay as ONE multi-tensor launch of the HIP library (pm_sgd_momentum_multi) instead of torch's three
foreach passes -- the optimizer of /root/reference/optimizer.py:11-32 (SGD(lr, weight_decay=5e-4, momentum, nesterov=False)).

A subclass of torch.optim.SGD: param_groups, `state[p]['momentum_buffer']`, state_dict() / load_state_dict() (the 'optimizer' entry of the
reference's checkpoints, utils/misc.py:195-216) and LR schedulers are torch's own; only step() is replaced. The update is torch's formula
    d = g + wd * p ;  buf = momentum * buf + d ;  p = p - lr * buf
with a fresh buffer starting at zero (0 * momentum + d == d, torch's first step). CPU parameters (the gloo tests) take torch's step."""
import torch

from .hip import kernels as K


def _dense_like(a, b):
    """Same element order in memory: strides agree on every dimension that has more than one element (a [Cout, Cin, 1, 1] weight gradient
    that arrives as a channels-last view has other strides on its unit dimensions and the same bytes)."""
    same = a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)
    return same and a.dtype == b.dtype == torch.float32 and a.is_cuda and b.is_cuda


class SGD(torch.optim.SGD):
    def __init__(self, params, lr=0.01, momentum=0.9, weight_decay=5e-4, nesterov=False, **kw):
        for k in ('foreach', 'fused', 'differentiable'):      # torch's implementation switches: step() here is its own implementation
            if kw.get(k):
                raise ValueError('pinthememory_amd.optim.SGD does not take %s=%r (step() is one multi-tensor launch of the HIP library)' % (k, kw[k]))
            kw.pop(k, None)
        super().__init__(params, lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, **kw)
        self.lr_device = None      # harness.GraphedAggStep: one-element CUDA tensor the fused update reads the learning rate from (single param group)
        self._register_filters()

    def _register_filters(self):
        """The transformed-filter cache (hip/kernels.py) keeps U = G w G^T only for weights whose owner bumps their version on every update."""
        K.register_filter_owners(p for g in self.param_groups for p in g['params'])

    def add_param_group(self, group):
        super().add_param_group(group)
        K.register_filter_owners(self.param_groups[-1]['params'])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from .hip import ops as _ops
        _ops.wait_commit()      # a memory-commit forward on its own stream (harness.agg_train_step) may still be reading the weights this step writes
        for group in self.param_groups:
            plain = (group['momentum'] != 0 and group['dampening'] == 0 and not group['nesterov'] and not group.get('maximize', False))
            fused, rest = [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if not (plain and p.is_cuda and p.dtype == torch.float32 and dense and not p.grad.is_sparse):
                    rest.append(p)       # CPU tensors (gloo tests), exotic layouts: torch's own arithmetic
                    continue
                g = p.grad
                if not _dense_like(p, g):      # e.g. a reducer that hands back row-major gradients for channels-last weights
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                st = self.state[p]
                buf = st.get('momentum_buffer')
                if buf is None or not _dense_like(p, buf):
                    new = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if buf is not None:
                        new.copy_(buf)
                    buf = st['momentum_buffer'] = new
                fused.append((p, g, buf))
            if fused:
                K.register_filter_owners(p for p, _, _ in fused)      # parameters moved to the GPU after the optimizer was built
                K.sgd_momentum_multi(fused, float(group['lr']), float(group['momentum']), float(group['weight_decay']), lr_device=getattr(self, 'lr_device', None))
                # the kernel writes through raw pointers: tell autograd (saved-tensor checks) and the Winograd filter cache (hip/kernels.py)
                torch.autograd.graph.increment_version([t for p, _, buf in fused for t in (p, buf)])
            if rest:
                self._torch_step(group, rest)
        K.refresh_filters()           # the kept filter transforms of the weights this step moved (bf16 copies / fp32 Winograd U), rewritten in one launch
        return loss

    def _torch_step(self, group, params):
        for p in params:
            g = -p.grad if group.get('maximize', False) else p.grad      # torch.optim.sgd: the gradient is negated first
            d = g.add(p, alpha=group['weight_decay']) if group['weight_decay'] != 0 else g
            if group['momentum'] != 0:
                st = self.state[p]
                buf = st.get('momentum_buffer')
                if buf is None:
                    buf = st['momentum_buffer'] = torch.clone(d).detach()
                else:
                    buf.mul_(group['momentum']).add_(d, alpha=1 - group['dampening'])
                d = d.add(buf, alpha=group['momentum']) if group['nesterov'] else buf
            p.add_(d, alpha=-group['lr'])
