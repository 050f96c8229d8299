import os
import sys

import pytest

# One ROCm runtime per process. torch bundles its own libamdhip64 / libhsa-runtime64 / librccl (same SONAMEs as /opt/rocm's);
# whichever copy is loaded first serves every later library of the process. Tests that bring torch in late (the RCCL hook, the
# distributed helpers) would otherwise start a second, device-less runtime next to the one libgfbe.so already initialised —
# so torch goes first, as it does in bench.py and in a torch-hosted deployment.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gf():
    from _gfbe_import import gf as mod
    return mod
