import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch (plumbing for device buffers in a few tests) bundles its own HIP runtime: when both it and
# libphi_amd.so live in one process, torch must initialise first (bench.py imports it first too).
try:
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
except ImportError:
    pass

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/): built on demand, test infrastructure only."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    from oracle import oracle as O
    return O


@pytest.fixture(scope="session")
def ctx_factory():
    """Contexts on cuda:0 through the C ABI; fails loudly when the HIP library is missing (a checkout
    without built libraries is built first: no fallback exists)."""
    import __graft_entry__
    __graft_entry__.ensure_built()
    import phi_amd
    made = []

    def make(**params):
        c = phi_amd.Context(0)
        if params:
            c.set_params(**params)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()
