"""Oracle (test infrastructure): the callers either side of the hot path, restated on CPU.

  agg_train_step      /root/reference/train.py:284-374 (train_memory_agg) + calculate_loss :213-244
  make_optimizer      /root/reference/optimizer.py:11-32  (SGD wd hard-coded 5e-4, exp LambdaLR)
  memory_initialize   /root/reference/train.py:1000-1042
  sliding_tiles       /root/reference/eval.py:148-194
  sliding_logits      /root/reference/eval.py:210-249,340-405 (logits summed over tiles / true count)
  fast_hist / miou    /root/reference/utils/misc.py:65-73,152-168
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOSS_W = dict(aux=0.4, read=0.02, div=0.4, cls=0.2)   # train.py:1213-1215 defaults (mem_readloss/divloss/clsloss)


def make_optimizer(net, lr=0.01, momentum=0.9, poly_exp=9):
    opt = torch.optim.SGD(list(p for _, p in net.named_parameters()), lr=lr, weight_decay=5e-4,
                          momentum=momentum, nesterov=False)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda it: math.exp(-1 * poly_exp * it / 120000))
    return opt, sched


def total_loss(outputs, w=LOSS_W):
    """calculate_loss (train.py:213-244) for the memory configuration without whitening."""
    main, aux = outputs[0], outputs[1]
    readloss, writeloss = outputs[-2], outputs[-3]
    return main + w['aux'] * aux + w['read'] * readloss + w['div'] * writeloss[0] + w['cls'] * writeloss[1]


def agg_train_step(net, opt, x, gts, aux_gts=None, sched=None):
    """One reference-faithful iteration. Returns dict of the five loss scalars (+ total)."""
    aux_gts = gts if aux_gts is None else aux_gts
    net.train()
    mem_t = net.memory.m_items.clone().detach()
    opt.zero_grad()
    outputs = net(x, gts=gts, aux_gts=aux_gts, memory_writing=True, writing_detach=False)
    loss = total_loss(outputs)
    loss.backward()
    opt.step()
    with torch.no_grad():
        net.eval()
        net.memory.m_items = mem_t
        net(x, gts=gts, aux_gts=aux_gts, memory_writing=True)
        net.train()
    if sched is not None:
        sched.step()
    return dict(loss1=outputs[0].detach(), loss2=outputs[1].detach(), readloss=outputs[-2].detach(),
                div=outputs[-3][0].detach(), cls=outputs[-3][1].detach(), total=loss.detach())


def memory_initialize(net, batches, epochs=2):
    """Class-prototype initialisation (train.py:1000-1042): eval/no-grad, bot_aspp features, no writenet."""
    mem = net.memory
    net.eval()
    with torch.no_grad():
        basket = torch.zeros_like(mem.m_items)
        count = torch.zeros(mem.memory_size, 1)
        for _ in range(epochs):
            for x, gt in batches:
                q = F.normalize(net(x, gts=gt, aux_gts=gt)[-1], dim=1)
                b, d, h, w = q.shape
                g = gt.clone()
                g[g == 255] = mem.memory_size
                g = F.one_hot(g, num_classes=mem.memory_size + 1)
                g = F.interpolate(g.permute(0, 3, 1, 2).contiguous().type(torch.float32), [h, w], mode='bilinear',
                                  align_corners=True).permute(0, 2, 3, 1).contiguous().view(b, -1, mem.memory_size + 1)
                den = g.sum(1).unsqueeze(1)
                nom = torch.matmul(q.view(b, d, -1), g)
                count += den[:, :, :mem.memory_size].sum(0).t()
                basket += nom[:, :, :mem.memory_size].sum(0).t()
        count[count == 0] = 1
        mem.m_items = F.normalize(basket / count, dim=1)
    net.train()
    return mem.m_items


def sliding_tiles(h, w, crop, overlap=1.0 / 3, scale=1.0):
    """Tile list [(x1,y1,x2,y2)] in the reference's order (eval.py:158-182)."""
    tile = int(crop * max(scale, 1.0))
    stride = math.ceil(tile * (1 - overlap))
    rows = int(math.ceil((w - tile) / stride) + 1)
    cols = int(math.ceil((h - tile) / stride) + 1)
    out = []
    for r in range(rows):
        for c in range(cols):
            x2, y2 = min(int(r * stride) + tile, w), min(int(c * stride) + tile, h)
            out.append((max(int(x2 - tile), 0), max(int(y2 - tile), 0), x2, y2))
    return out


def sliding_logits(net, img, crop, overlap=1.0 / 3, flips=(False, True)):
    """Single-scale sliding-window logits for one CHW image: tiles forwarded one by one, logits summed and
    divided by the per-pixel tile count, mean over flips (eval.py:340-405; the reference's count array is
    mis-indexed, eval.py:216-219 -- class-uniform, so argmax is unaffected; we divide by the true count)."""
    c, h, w = img.shape
    tiles = sliding_tiles(h, w, crop, overlap)
    net.eval()
    acc = None
    with torch.no_grad():
        for flip in flips:
            src = torch.flip(img, dims=[2]) if flip else img
            full = torch.zeros(0)
            cnt = torch.zeros(1, h, w, dtype=torch.float64)
            for (x1, y1, x2, y2) in tiles:
                lg = net(src[None, :, y1:y2, x1:x2].contiguous())[0][0].to(torch.float64)
                if full.numel() == 0:
                    full = torch.zeros(lg.shape[0], h, w, dtype=torch.float64)
                full[:, y1:y2, x1:x2] += lg
                cnt[:, y1:y2, x1:x2] += 1
            full = full / cnt
            if flip:
                full = torch.flip(full, dims=[2])
            acc = full if acc is None else acc + full
    return acc / len(flips)


def fast_hist(pred, gt, n=19):                        # utils/misc.py:65-70
    pred, gt = np.asarray(pred).reshape(-1), np.asarray(gt).reshape(-1)
    k = (gt >= 0) & (gt < n)
    return np.bincount(n * gt[k].astype(int) + pred[k].astype(int), minlength=n * n).reshape(n, n)


def miou(hist):                                       # utils/misc.py:152-168
    with np.errstate(divide='ignore', invalid='ignore'):
        iu = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))
    return float(np.nanmean(iu)), iu


# ---- train_memory_mldg (train.py:493-632) with get_updated_network / put_theta (train.py:246-277) -------------------------
def put_theta(model, theta):
    """Rewire every leaf module's _parameters with the tensors in `theta` (non-leaf, still attached to the graph)."""
    def walk(mod, name=None):
        if len(mod._modules) != 0:
            for k, v in mod._modules.items():
                walk(v, str(k) if name is None else str(name + '.' + k))
        else:
            for k, v in mod._parameters.items():
                if isinstance(v, torch.Tensor):
                    mod._parameters[k] = theta[str(name + '.' + k)]
    walk(model)
    return model


def get_updated_network(old, new, lr):
    """theta' = theta - lr * grad for every parameter with a gradient (first-order: grad is a constant), buffers copied."""
    sd, params = old.state_dict(), dict(old.named_parameters())
    theta = {k: (params[k] - lr * params[k].grad if k in params and params[k].grad is not None else sd[k]) for k in sd}
    return put_theta(new, theta)


def mldg_train_step(net, updated_net, updated_net2, opt, x_tr, y_tr, x_te, y_te, inner_lr=0.001, sched=None):
    """One iteration of train_memory_mldg for the memory configuration (no whitening). Returns inner/outer loss dicts.
    inner_lr: train.py:1208 default 0.001 (with --inner_lr_anneal the caller passes lr / 4 of the outer schedule, train.py:625-626)."""
    net.train()
    mem_t = net.memory.m_items.clone().detach()
    opt.zero_grad()
    out_in = net(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True, writing_detach=False)
    inner = total_loss(out_in)
    inner.backward(retain_graph=True)
    updated_net = get_updated_network(net, updated_net, inner_lr).train()
    updated_net2 = get_updated_network(net, updated_net2, inner_lr).train()
    updated_net2.memory.m_items = mem_t
    for k, v in updated_net2.named_parameters():               # freeze the encoder (train.py:549-552)
        if k.split('.')[0] != 'memory':
            v.detach_()
            v.requires_grad_(False)
    updated_net2(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True, writing_detach=False)   # seen-domain memory write
    updated_net.memory.m_items = updated_net2.memory.m_items.clone()
    out_te = updated_net(x_te, gts=y_te, aux_gts=y_te, memory_writing=False)                # meta-test: read only
    outer = out_te[0] + LOSS_W['aux'] * out_te[1] + LOSS_W['read'] * out_te[-2]            # writeloss is [0, 0]
    outer.backward()
    opt.step()
    with torch.no_grad():
        net.eval()
        net.memory.m_items = mem_t
        net(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True)
        net.train()
    if sched is not None:
        sched.step()
    return dict(inner=inner.detach(), outer=outer.detach(), inner_loss1=out_in[0].detach(), outer_loss1=out_te[0].detach(),
                outer_read=out_te[-2].detach())


# ---- inference_pool + MeanFusion (eval.py:133-145,277-337) ---------------------------------------------------------------
def inference_pool(net, imgs, orisize, no_flip=False):
    """imgs[flip][scale] -> (probs, preds): softmax of the half-pixel-upsampled logits, float64 running mean, max over classes."""
    net.eval()
    buf = torch.zeros(imgs[0][0].shape[0], 19, orisize[0], orisize[1], dtype=torch.float64)
    cnt = 0
    with torch.no_grad():
        for flip in range(1 if no_flip else 2):
            for img in imgs[flip]:
                y = F.interpolate(net(img)[0], size=orisize, mode='bilinear')
                if flip == 1:
                    y = torch.flip(y, dims=[3])
                cnt += 1
                buf.add_((F.softmax(y, dim=1) - buf) / cnt)
    return buf.max(1)
