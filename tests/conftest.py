import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """Binding over the CPU oracle (test infrastructure)."""
    from oracle_engine import oracle_binding
    return oracle_binding()


@pytest.fixture(scope="session")
def hip():
    """The product binding; raises if libmadarch_hip.so is missing (no fallback)."""
    from madarch_amd import _binding
    return _binding.hip_binding()
