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


@pytest.fixture(scope="session")
def gpu_ctx_per_node(native_lib):
    """Same device, kernel-per-expression-node executor (QE_EXEC_PER_NODE)."""
    from queryengine_amd import engine, native
    ctx = engine.Context(device=0, exec_mode=native.EXEC_PER_NODE)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx_two_pass(native_lib):
    """Fused executor forced into its two-pass form (count, scan, direct ordered write) -- normally chosen only after a
    plan has shown high selectivity (debug bit 512 of qe_options.tuning[5])."""
    from queryengine_amd import engine
    ctx = engine.Context(device=0, tuning=[0, 0, 0, 0, 0, 512, 0, 0])
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx_dense(native_lib):
    """Fused executor forced into its dense single-pass form (workgroup tiles, the kept rows of a tile parked in LDS and
    resolved one tile later) -- normally chosen once a plan has kept >= 12 % of its rows (debug bit 16384 of tuning[5])."""
    from queryengine_amd import engine
    ctx = engine.Context(device=0, tuning=[0, 0, 0, 0, 0, 16384, 0, 0])
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx_local(native_lib):
    """Fused executor forced into its local form (dependency-free scan into per-chunk slots, then a scan over the counts and
    one move) -- normally chosen on large batches once a plan has kept <= 3 % of its rows (debug bit 262144 of tuning[5]).
    A chunk that keeps more rows than its slot holds makes the execution fall back to the single-pass kernel: the parity
    suite's high-selectivity cases run through exactly that fallback."""
    from queryengine_amd import engine
    ctx = engine.Context(device=0, tuning=[0, 0, 0, 0, 0, 262144, 0, 0])
    yield ctx
    ctx.close()


@pytest.fixture(params=["fused", "per_node", "fused_two_pass", "fused_dense", "fused_local"])
def any_ctx(request, gpu_ctx, gpu_ctx_per_node, gpu_ctx_two_pass, gpu_ctx_dense, gpu_ctx_local):
    """Every parity test runs through ALL execution forms, like the reference's
    @EnumSource(Mode::class) tests run through all three evaluators (CompilerTest.kt:13)."""
    return {"fused": gpu_ctx, "per_node": gpu_ctx_per_node, "fused_two_pass": gpu_ctx_two_pass,
            "fused_dense": gpu_ctx_dense, "fused_local": gpu_ctx_local}[request.param]
