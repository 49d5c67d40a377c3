import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # as config.py does: before the HIP runtime initialises
os.environ.setdefault("CTM_ABORT_BACKTRACE", "1")   # csrc/ctm_runtime.hip: native stack on SIGABRT / SIGSEGV (the rare adjoint abort)
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "peps-torch_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import _native
    return _native.engine()


def golden(name):
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
