import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own C++ core (oracle/_ref), if it was built in this container."""
    from oracle.pyoracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libtrlda_ref.so not built (needs /root/reference)")
    return Reference()


@pytest.fixture(scope="session")
def hip_lib():
    """libtrlda_hip.so, built on demand (hipcc cross-compiles without a GPU)."""
    from trlda_amd import build
    build.build()
    from trlda_amd import _ffi
    return _ffi.lib()


GOLDEN = os.path.join(ROOT, "tests", "golden")
