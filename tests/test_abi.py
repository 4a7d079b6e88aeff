"""The C-ABI library loads and exports every symbol include/qe_hip.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

from queryengine_amd import native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "qe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qe_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(native_lib):
    names = declared_symbols()
    assert len(names) >= 35
    bound = {n for n, _, _ in N.SYMBOLS}
    for n in names:
        assert hasattr(native_lib, n), f"{n} declared in qe_hip.h but not exported by libqe_hip.so"
        assert n in bound, f"{n} has no ctypes binding"
    assert bound <= set(names), f"bindings without a declaration: {bound - set(names)}"


def test_no_cpu_fallback_without_device(native_lib):
    """In this container there is no GPU: context creation on a device must fail loudly."""
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    st = native_lib.qe_ctx_create(0, None, C.byref(h))
    assert st == 3 and b"no CPU fallback" in native_lib.qe_last_error(None)


def test_product_does_not_reference_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "queryengine_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "qe_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_product_does_not_reference_the_test_transport():
    """tests/transport (the stand-in for librccl of the multi-rank exchange tests) is test infrastructure: the product
    only knows the QE_RCCL_LIBRARY variable."""
    pkg = os.path.join(ROOT, "queryengine_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "qe_test_transport" not in text and "tests/transport" not in text, f
    # bench.py runs the exchange of an N-ranks-on-one-GPU rehearsal only when the CALLER names a transport (QE_RCCL_LIBRARY);
    # __graft_entry__.build() compiles the transport (building the checker is not using it)
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert "qe_test_transport" not in text and "tests/transport" not in text
