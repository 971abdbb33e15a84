import importlib
import os
import sys

import pytest

# the OpenMP oracle is fastest near 16 threads; a 256-thread default (GPU box) makes its small vectors crawl
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 8)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure)."""
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def shp():
    """The product package (ctypes mirror of libshielder_hip)."""
    return importlib.import_module("zkos-monorepo_amd")


@pytest.fixture(scope="session")
def gpu_ctx(shp):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = shp.Context(0)
    yield ctx
    ctx.close()


def golden_path(name):
    return os.path.join(GOLDEN, name)
