"""Oracle tooling (test infrastructure): import the reference's own hot-path modules from /root/reference.

Works ONLY in the build container (the reference never travels to the GPU box). Recipe of SURVEY.md
Appendix A: 5 sys.modules stubs for absent optional deps + identity .cuda() + no-network model_zoo +
cfg.MODEL.BNFUNC = BatchNorm2d. Nothing from the reference is copied; it is imported where it lies.
"""
import argparse
import os
import sys
import types

REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'network'))


def load():
    """Returns (deepv3plus, deepv2, memory) reference modules."""
    import torch
    if not available():
        raise RuntimeError('reference tree not present (expected only in the build container)')
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m
    if 'torchvision' not in sys.modules:
        tv = stub('torchvision')
        tv.models = stub('torchvision.models')
    stub('kmeans1d')
    stub('datasets', num_classes=19, ignore_label=255)
    tr = stub('transforms')
    tr.transforms = stub('transforms.transforms', HideAndSeek=object)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import torch.utils.model_zoo as mz
    mz.load_url = lambda *a, **k: {}
    from config import assert_and_infer_cfg, cfg
    if not cfg.is_immutable():
        assert_and_infer_cfg(argparse.Namespace(syncbn=False), train_mode=False)
    from network import deepv3plus, deepv2, memory
    return deepv3plus, deepv2, memory
