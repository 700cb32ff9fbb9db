"""pytest configuration: markers, paths, and the oracle / HIP library fixtures.

``-m "not gpu"`` : oracle vs golden vectors and the reference's own statistical tests, host logic,
                   C-ABI symbol checks, world_size-2 gloo runs.  No GPU needed.
``-m gpu``       : parity tests proper -- every one calls the HIP path through the C ABI and
                   compares with the oracle / golden vectors.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~10 s on 8 host cores")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    oracle_lib.load()
    return oracle_lib


@pytest.fixture(scope="session")
def amc():
    """The product's ctypes binding; the .so must have been built (no fallback)."""
    from montecarlo_amd import _capi
    _capi.load()
    return _capi


@pytest.fixture(scope="session")
def gpu(amc):
    n = amc.device_count()
    if n < 1:
        pytest.fail("test is marked gpu but libamc.so sees no HIP device")
    return amc
