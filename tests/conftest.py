import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope='session', autouse=True)
def _oracle_threads():
    """The CPU oracle legs of the -m gpu suite: 16-32 threads (on a 128-core host torch's default of one thread per core is 3.7 x slower than 16
    -- bench.py's own sweep -- and the driver's box took 750 s for a suite that ran 490 s elsewhere)."""
    try:
        import torch
        torch.set_num_threads(max(1, min(24, os.cpu_count() or 1)))
    except Exception:      # noqa: BLE001
        pass
    yield


@pytest.fixture(scope='session')
def oracle_cache():
    """Session-wide memo of CPU-oracle results keyed by (what, dtype, shapes, seeds ...): each oracle configuration is computed once per suite run."""
    return {}
