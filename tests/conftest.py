import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The library sends 3-D grids below ~52^3 cells to its direct kernel (one thread per cell: faster there).  This suite's small grids exist to
# exercise the TILED kernels against the oracle, so the switch is off here; test_small_grids_take_the_direct_kernel_by_default checks the
# default and that both kernels agree bit for bit.
os.environ.setdefault("HJ_DIRECT_BELOW", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name)))
        return cache[name]
    return load
