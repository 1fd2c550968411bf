"""Norm selection, bilinear Upsample and weight init -- the surface of /root/reference/network/mynn.py,
with Upsample running the HIP resize kernel on channels-last memory."""
import torch
import torch.nn as nn

from ..hip import ops

BNFUNC = None   # None -> resolve lazily: the reference's cfg.MODEL.BNFUNC when its config module is importable, else BatchNorm2d


def set_bnfunc(layer):
    """config.py:109-119 equivalent: nn.BatchNorm2d (local statistics) or nn.SyncBatchNorm."""
    global BNFUNC
    BNFUNC = layer


def Norm2d(in_channels):                              # mynn.py:8-14
    layer = BNFUNC
    if layer is None:
        try:
            from config import cfg                    # dropped into the reference tree: honour its global knob
            layer = getattr(cfg.MODEL, 'BNFUNC')
        except Exception:
            layer = nn.BatchNorm2d
    return layer(in_channels)


def freeze_weights(*models):
    for model in models:
        for k in model.parameters():
            k.requires_grad = False


def unfreeze_weights(*models):
    for model in models:
        for k in model.parameters():
            k.requires_grad = True


def initialize_weights(*models):                      # mynn.py:27-44
    for model in models:
        for module in model.modules():
            if isinstance(module, (nn.Conv2d, nn.Linear, nn.Conv1d)):
                nn.init.kaiming_normal_(module.weight, nonlinearity='relu')
                if module.bias is not None:
                    module.bias.data.zero_()
            elif isinstance(module, (nn.BatchNorm2d, nn.BatchNorm1d, nn.GroupNorm, nn.SyncBatchNorm)):
                module.weight.data.fill_(1)
                module.bias.data.zero_()


def Upsample(x, size):                                # mynn.py:57-62, bilinear align_corners=True
    return ops.resize(x, size)


def forgiving_state_restore(net, loaded_dict):        # mynn.py:64-80
    net_state_dict = net.state_dict()
    new_loaded_dict = {}
    for k in net_state_dict:
        if k in loaded_dict and net_state_dict[k].size() == loaded_dict[k].size():
            new_loaded_dict[k] = loaded_dict[k]
        else:
            print("Skipped loading parameter", k)
    net_state_dict.update(new_loaded_dict)
    net.load_state_dict(net_state_dict)
    return net


def channels_last_weights(model):
    """Store every 4-D conv weight in channels_last memory == the KRSC layout the kernels read (free view, same
    state_dict shapes). load_state_dict / .cuda() / optimizers preserve it."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return model
