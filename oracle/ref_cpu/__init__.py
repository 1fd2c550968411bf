"""Oracle (test infrastructure): stock-torch CPU restatement of the reference hot path. See oracle/__init__.py."""
from . import deeplab, harness, memory, resnet  # noqa: F401
