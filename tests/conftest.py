"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path.

``-m "not gpu"`` runs here on CPU (oracle pins, host logic, C-ABI symbol checks,
gloo world_size-2 sharding); ``-m gpu`` runs on a real MI355X through the C-ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def product_asm(repo_root):
    """The compiler's assembly of the product kernels (hipcc -S --cuda-device-only: cross-compiles on the CPU, ~70 s), compiled
    once per session and cached under the temporary directory by the hash of source + header: what the ISA lint
    (tests/test_isa_lint.py) and the register-budget check (tests/test_host_logic_cpu.py) read."""
    import hashlib
    import shutil
    import subprocess
    import tempfile

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(repo_root, "t-svgp_amd", "csrc", "tsvgp_kernels.hip")
    hdr = os.path.join(repo_root, "include", "tsvgp_hip.h")
    h = hashlib.sha256(open(src, "rb").read() + open(hdr, "rb").read()).hexdigest()[:16]
    out = os.path.join(tempfile.gettempdir(), f"tsvgp_kernels_{h}.s")
    if not os.path.exists(out):
        tmp = out + f".{os.getpid()}.tmp"
        subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-I",
                        os.path.join(repo_root, "include"), src, "-o", tmp], check=True, capture_output=True, timeout=900)
        os.replace(tmp, out)
    return open(out).read()
