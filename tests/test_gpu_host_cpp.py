"""The C++ host mirror (queryengine_amd/host/qe_host.hpp) restating the reference's CompilerTest /
ByteCodeCompilerTest against the GPU through the C ABI, both execution modes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "queryengine_amd", "host")


@pytest.mark.gpu
def test_cpp_host_mirror_runs_reference_tests(native_lib):
    exe = os.path.join(HOST, "test_host")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", HOST], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host tests passed" in r.stdout


def test_cpp_host_mirror_builds_and_fails_loudly_without_gpu(native_lib):
    subprocess.run(["make", "-C", HOST], check=True)
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([os.path.join(HOST, "test_host")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no CPU fallback" in r.stdout
