"""CPU checks of the C-ABI boundary: the in-tree library loads and exports every symbol include/tsvgp_hip.h declares,
the ctypes prototypes cover exactly those symbols, and argument validation works without a GPU (no compute calls)."""
import os
import re

import pytest

from tests.helpers import pkg


def _declared_symbols(header):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tsvgp_\w+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol(repo_root):
    B = pkg()._backend
    path = pkg().build_library()
    assert os.path.exists(path)
    lib = B.lib()
    declared = _declared_symbols(os.path.join(repo_root, "include", "tsvgp_hip.h"))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tsvgp_hip.h but not exported by {path}"
    assert sorted(B.exported_symbols()) == declared  # the ctypes table and the header agree
    assert lib.tsvgp_version().decode().startswith("tsvgp_hip gfx950")


def test_argument_validation_needs_no_gpu():
    lib = pkg()._backend.lib()
    # null pointers / unpadded sizes are rejected before any launch
    assert lib.tsvgp_trmm_f64(None, None, None, 128, 128, 0, None) == 1
    assert lib.tsvgp_trmm_f32(None, None, None, 100, 128, 0, None) == 1
    assert lib.tsvgp_moments_f64(None, None, None, None, 1.0, 0, 0.0, None, None, None, None, None, None, 1, 128, 128, 1, 0, None) == 1
    assert lib.tsvgp_site_accum_f64(None, None, None, None, None, None, 128, 128, 1, 1, None) == 1
    assert lib.tsvgp_potrf_f64(None, 128, 128, 1, 0, None, None, 0, None) == 1
    # diagonal tiles get ceil(20 ns / 32) slices in fp64 (syrk1_kernel) and ceil(23 ns / 32) in fp32 (syrk_kernel)
    assert lib.tsvgp_site_accum_work_bytes_f64(1024, 1, 15) == (28 * 15 + 8 * 10) * 128 * 128 * 8 + 10 * 1024 * 8
    assert lib.tsvgp_site_accum_work_bytes_f32(1024, 1, 15) == (28 * 15 + 8 * 11) * 128 * 128 * 4 + 11 * 1024 * 4
    assert lib.tsvgp_site_accum_work_bytes_f32(1000, 1, 15) == -1


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: with no ROCm device the model refuses to run the E-step (it never routes through the oracle)."""
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = pkg()
    m = p.t_SVGP(p.SquaredExponential(), p.Gaussian(0.1), np.zeros((4, 2)) + np.arange(4)[:, None])
    for call in (lambda: m.natgrad_step((np.zeros((5, 2)), np.zeros((5, 1)))), lambda: m.elbo((np.zeros((5, 2)), np.zeros((5, 1)))),
                 lambda: m.predict_f(np.zeros((5, 2)))):
        with pytest.raises(p.HipExtensionError):
            call()
    w = p.t_SVGP_white(p.SquaredExponential(), p.Gaussian(0.1), np.zeros((4, 2)) + np.arange(4)[:, None])
    data = (np.zeros((5, 2)), np.zeros((5, 1)))
    for call in (lambda: w.natgrad_step(data), lambda: w.elbo(data), lambda: w.predict_f(data[0]), lambda: m.elbo_and_grads(data)):
        with pytest.raises(p.HipExtensionError):
            call()
    assert lib_validation_extra()


def lib_validation_extra():
    lib = pkg()._backend.lib()
    ok = lib.tsvgp_potrf_inv_f64(None, 128, 128, 1, 0, None, None, None, None, None, 0, None) == 1
    ok &= lib.tsvgp_kernel_fill_f64(7, None, None, None, 1.0, None, 10, 4, 2, 128, None) == 1
    ok &= lib.tsvgp_kernel_grad_f64(0, None, None, None, 1.0, None, 128, None, None, 1, None, 1, 10, 4, 2, None, None, None, None) == 1
    ok &= lib.tsvgp_kernel_grad_rows() == 1024 and lib.tsvgp_kernel_grad_dpad(5) == 8
    return bool(ok)


def test_package_never_imports_the_oracle(repo_root):
    """oracle/ is test infrastructure: nothing under t-svgp_amd/ may import or reference it."""
    for dirpath, _, files in os.walk(os.path.join(repo_root, "t-svgp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, f"{f} mentions the oracle"
