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
    from oracle import qe_oracle
    qe_oracle.build()
    return qe_oracle


@pytest.fixture(scope="session")
def native_lib():
    """libqe_hip.so built in-tree (hipcc cross-compiles without a GPU)."""
    from queryengine_amd import native
    if not os.path.exists(native.LIB_PATH):
        native.build()
    return native.lib()


@pytest.fixture(scope="session")
def gpu_ctx(native_lib):
    """One device context for the whole GPU session (JIT cache stays warm)."""
    from queryengine_amd import engine
    ctx = engine.Context(device=0)
    yield ctx
    ctx.close()
