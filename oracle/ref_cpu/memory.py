"""Oracle (test infrastructure): categorical memory read / write, stock torch ops on CPU.

Restates /root/reference/network/memory.py:
  initialize_weights :9-19    Writingnet :67-87    Memory_sup.__init__ :94-122
  get_score          :167-189 forward    :191-204  write               :206-257
  classification_loss:259-262 diversityloss :264-272  read             :317-336
Device-agnostic (the reference hard-codes .cuda()); m_items is a plain attribute,
re-assigned (not updated in place) by write(), exactly like the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _init(model):
    # memory.py:9-19 -- conv kaiming_normal(fan_in), BN weight 1 / bias 1e-4, Linear N(0,1e-4)/0
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight.data, nonlinearity='relu')
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1.)
            m.bias.data.fill_(1e-4)
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(0.0, 0.0001)
            m.bias.data.zero_()


class Writingnet(nn.Module):
    def __init__(self, input_feature_dim, feature_dim):
        super().__init__()
        assert input_feature_dim == feature_dim
        self.writefeat = nn.Sequential(nn.Conv2d(input_feature_dim, feature_dim, 1, bias=False),
                                       nn.BatchNorm2d(feature_dim))
        self.relu = nn.ReLU(inplace=True)
        _init(self)                                   # memory.py:81 (first of two inits)

    def forward(self, x):                             # memory.py:83-87
        return self.relu(x + self.writefeat(x))


class Memory_sup(nn.Module):
    def __init__(self, memory_size, input_feature_dim, feature_dim, momentum, temperature, gumbel_read):
        super().__init__()
        self.memory_size, self.feature_dim = memory_size, feature_dim
        self.momentum = self.initial_momentum = momentum
        self.temperature = temperature
        self.output = nn.Sequential(nn.Conv2d(feature_dim * 2, input_feature_dim, 1, bias=False),
                                    nn.BatchNorm2d(input_feature_dim), nn.ReLU(inplace=True))
        self.writenet = Writingnet(input_feature_dim, feature_dim)
        self.mem_cls = torch.arange(memory_size)
        self.clsfier = nn.Linear(feature_dim, memory_size, bias=True)
        self.celoss = nn.CrossEntropyLoss(ignore_index=255)
        self.gumbel_read = gumbel_read
        self.m_items = F.normalize(torch.rand((memory_size, feature_dim), dtype=torch.float), dim=1)
        _init(self)                                   # memory.py:122 (re-initialises writenet too)

    # -- read ---------------------------------------------------------------------------------
    def get_score(self, query, mask, mem):            # memory.py:167-189, query is NHWC
        bs, h, w, d = query.shape
        m = mem.shape[0]
        score = torch.matmul(query, mem.t())
        if mask is not None:
            s = (score / self.temperature).permute(0, 3, 1, 2).contiguous()
            s = F.interpolate(s, size=mask.shape[1:], mode='bilinear', align_corners=True)
            readloss = self.celoss(s, mask)
        else:
            readloss = 0
        score = score.view(bs * h * w, m)
        if self.gumbel_read:
            sq, sm = F.gumbel_softmax(score, dim=0), F.gumbel_softmax(score, dim=1)
        else:
            sq, sm = F.softmax(score, dim=0), F.softmax(score, dim=1)
        return sq, sm, readloss

    def read(self, query, mask, memory_writing):      # memory.py:317-336
        q = F.normalize(query.clone(), dim=1).permute(0, 2, 3, 1).contiguous()
        b, h, w, d = q.shape
        if memory_writing:
            self.m_items = self.m_items.detach()
        sq, sm, readloss = self.get_score(q, mask, self.m_items)
        agg = torch.matmul(sm, self.m_items)
        u = torch.cat((q.view(b * h * w, d), agg), dim=1).view(b, h, w, 2 * d).permute(0, 3, 1, 2).contiguous()
        u = self.output(u)
        return u, sq.view(b, h, w, self.memory_size), sm.view(b, h, w, self.memory_size), readloss

    # -- write --------------------------------------------------------------------------------
    def soft_labels(self, mask, h, w, dtype=torch.float32):
        """memory.py:220-225: 255 -> slot 19, one-hot(20) int64 -> float -> bilinear(align_corners) to h x w.
        dtype is float32 in the reference; the fp64 "truth" runs of the parity tests pass float64."""
        t = mask.clone().detach()
        t[t == 255] = self.memory_size
        t = F.one_hot(t, num_classes=self.memory_size + 1)
        t = F.interpolate(t.permute(0, 3, 1, 2).contiguous().type(dtype), [h, w],
                          mode='bilinear', align_corners=True).permute(0, 2, 3, 1).contiguous()
        return t.view(mask.shape[0], -1, self.memory_size + 1)

    def accumulate(self, zhat, mask):
        """memory.py:219-231: nominator [20,d] and denominator [20], summed over batch and pixels."""
        b, d, h, w = zhat.shape
        y = self.soft_labels(mask, h, w, zhat.dtype)
        den = y.sum(1).unsqueeze(1)
        nom = torch.matmul(zhat.view(b, d, -1), y)
        return nom.sum(0).t(), den.sum(0).squeeze()

    def update(self, nom, den):
        """memory.py:233-239: per-slot momentum update where the class occurs, then row-normalise."""
        upd = self.m_items.clone().detach()
        for slot in range(self.memory_size):
            if den[slot] != 0:
                upd[slot] = self.momentum * self.m_items[slot] + ((1 - self.momentum) * nom[slot] / den[slot])
        return F.normalize(upd, dim=1)

    def write(self, x, mask, writing_detach=True):    # memory.py:206-257
        z = F.normalize(self.writenet(x.clone()), dim=1)
        nom, den = self.accumulate(z, mask)
        upd = self.update(nom, den)
        losses = [self.diversityloss(upd), self.classification_loss(upd)]
        self.m_items = upd.detach() if writing_detach else upd
        return losses

    def classification_loss(self, mem):               # memory.py:259-262
        return self.celoss(self.clsfier(mem), self.mem_cls)

    def diversityloss(self, mem):                     # memory.py:264-272
        g = torch.matmul(mem, mem.t()) - 0
        g[g < 0] = 0
        return (torch.sum(g) - torch.trace(g)) / (self.memory_size * (self.memory_size - 1))

    def forward(self, query, mask=None, memory_writing=True, writing_detach=True):   # memory.py:191-204
        u, sq, sm, readloss = self.read(query, mask, memory_writing)
        writeloss = self.write(query, mask, writing_detach) if memory_writing else [0, 0]
        return u, sq, sm, readloss, writeloss
