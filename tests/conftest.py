import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu")


def load_oracle():
    """Build (if needed) and load the CPU oracle as a checker backend. Tests only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle.build import build_oracle
    from pasco_amd.me.backend import CBackend
    return CBackend(build_oracle(), "pho_", "cpu")


@pytest.fixture(scope="session")
def oracle():
    return load_oracle()


@pytest.fixture()
def oracle_registered(oracle):
    """Serve CPU tensors from the oracle for the duration of one test."""
    from pasco_amd.me import backend
    backend.register_checker_backend(oracle)
    yield oracle
    backend.register_checker_backend(None)


@pytest.fixture(scope="session")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pasco_amd.me.backend import hip_backend
    return hip_backend()
