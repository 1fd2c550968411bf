"""Input edge of the training step: what sits between the reference's DataLoader and `net(x, gts=...)`.

The reference's DomainUniformConcatDataset (datasets/multi_loader.py:81-102) stacks one sample per source domain, so a batch
arrives as float32 `[B, D, 3, H, W]` images and int64 `[B, D, H, W]` label maps; train.py:297-307 merges the domain axis and
copies both to the GPU. ToTensor + Normalize (transforms/transforms.py:95-97, train-time `mean_std` of datasets/__init__.py)
ran on the host before that. Here the host hands over what the decoder produced -- uint8 pixels and uint8 train ids, 4x / 8x
fewer PCIe bytes -- in pinned memory; the copy runs on a side stream while the previous step computes, and ToTensor / Normalize /
the int64 widening are two small kernels on that same stream (`pm_image_u8_to_nhwc4`, `pm_labels_u8_to_i64`).

`SyntheticDomainSource` is the stand-in for the loader (no datasets in this build); `DevicePrefetcher` is the product part.
"""
import torch

from .hip import kernels as K
from .hip import ops


class SyntheticDomainSource:
    """Endless `[B, D, H, W, 3]` uint8 images + `[B, D, H, W]` uint8 train ids (255 = ignore) in `n_buffers` rotating host buffers,
    pinned when a GPU is present. Deterministic in (seed, batch index): batch i is the same bytes on every run and every rank layout."""

    def __init__(self, batch, domains, size, n_buffers=3, seed=0, classes=19, static=False):
        h, w = (size, size) if isinstance(size, int) else size
        self.shape = (batch, domains, h, w)
        self.seed, self.classes, self.i, self.static = seed, classes, 0, static
        pin = torch.cuda.is_available()
        self.bufs = [(torch.empty((batch, domains, h, w, 3), dtype=torch.uint8, pin_memory=pin),
                      torch.empty((batch, domains, h, w), dtype=torch.uint8, pin_memory=pin)) for _ in range(n_buffers)]
        if static:                                                             # the ring is filled once and replayed (decode cost belongs to loader workers)
            for i, (img, lab) in enumerate(self.bufs):
                self.fill(i, img, lab)

    def bytes_per_batch(self):
        b, d, h, w = self.shape
        return b * d * h * w * 4

    def fill(self, i, img, lab):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        img.copy_(torch.randint(0, 256, img.shape, generator=g, dtype=torch.uint8))
        blocks = torch.randint(0, self.classes + 1, (self.shape[0], self.shape[1], (self.shape[2] + 15) // 16, (self.shape[3] + 15) // 16), generator=g)
        blocks[blocks == self.classes] = 255                                   # ignore_label regions (datasets/cityscapes_labels.py trainId 255)
        lab.copy_(blocks.repeat_interleave(16, 2).repeat_interleave(16, 3)[:, :, :self.shape[2], :self.shape[3]].to(torch.uint8))

    def __iter__(self):
        return self

    def __next__(self):
        img, lab = self.bufs[self.i % len(self.bufs)]
        if not self.static:
            self.fill(self.i, img, lab)
        self.i += 1
        return img, lab


class DevicePrefetcher:
    """Keeps `depth` batches in flight: H2D of the uint8 buffers and the u8 -> NHWC4 float / int64 kernels run on a side stream, an
    event hands the result to the compute stream (`next()` makes the current stream wait on it; the host never blocks on the GPU
    except to keep the source from refilling a pinned buffer whose copy is still in flight). The raw uint8 device slots are only
    touched by the side stream, so its own order protects them; the converted tensors are handed over with `record_stream`."""

    def __init__(self, source, depth=1, device=None):
        assert torch.cuda.is_available(), 'the input edge stages into HBM: needs a GPU'
        self.src, self.depth = iter(source), depth
        self.host_ring = max(1, len(getattr(source, 'bufs', [])) or 1)          # pinned buffers the source rotates through
        self.dev = device or torch.device('cuda', torch.cuda.current_device())
        self.side = torch.cuda.Stream(device=self.dev)
        self.slots, self.copied, self.ready = [None] * (depth + 1), [], []
        self.n = 0
        for _ in range(depth):
            self._issue()

    def _issue(self):
        if len(self.copied) >= self.host_ring:
            self.copied.pop(0).synchronize()                                   # the buffer the source is about to refill has left the host
        img_h, lab_h = next(self.src)
        k = self.n % len(self.slots)
        self.n += 1
        h, w = img_h.shape[-3:-1]
        with torch.cuda.stream(self.side):
            if self.slots[k] is None:
                self.slots[k] = (torch.empty(img_h.reshape(-1, h, w, 3).shape, dtype=torch.uint8, device=self.dev),
                                 torch.empty(lab_h.reshape(-1, h, w).shape, dtype=torch.uint8, device=self.dev))
            img_d, lab_d = self.slots[k]
            img_d.copy_(img_h.reshape(-1, h, w, 3), non_blocking=True)        # train.py:297-307: the domain axis merges into the batch
            lab_d.copy_(lab_h.reshape(-1, h, w), non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.side)
            x = ops.nchw(K.image_u8_to_nhwc4(img_d))
            gt = K.labels_u8_to_i64(lab_d)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.copied.append(done)
        self.ready.append((x, gt, ev))

    def next(self):
        """(x, gts) of the oldest batch in flight, valid on the current stream; issues the copy of the one after."""
        x, gt, ev = self.ready.pop(0)
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(ev)
        x.record_stream(cur)
        gt.record_stream(cur)
        self._issue()
        return x, gt
