import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# torch's bundled HIP runtime must be the first one loaded in a process that also uses torch.cuda
# (see gdpathtracing_amd/capi.py); harmless on the CPU-only box.
try:
    import torch
    torch.cuda.is_available()
except Exception:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def hiplib():
    """The HIP library, built in-tree; GPU tests fail loudly if it is missing."""
    from gdpathtracing_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    return capi.lib()
