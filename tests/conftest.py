import os
import sys

import pytest

# The CPU oracle is OpenMP code; a GPU box can show hundreds of CPUs in its affinity mask while its CPU share lets
# only a handful run (a team of that size thrashes: the same suite took 20 s on one box and 190 s on another).
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import cbv_oracle
    cbv_oracle.lib()
    return cbv_oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """The HIP context; fails loudly (no skip) when the library or device is missing."""
    from chessboard_vision_amd import _native as N
    return N.context()
