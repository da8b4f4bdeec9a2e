"""ctypes binding of the C-ABI HIP library (include/tsvgp_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (or ``build_library()`` below) as
``t-svgp_amd/csrc/libtsvgp_hip.so``.  There is NO fallback: if the library is missing or a tensor is not
on a ROCm device the product path raises -- it never silently computes on the CPU.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libtsvgp_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "tsvgp_hip.h")
SOURCES = [os.path.join(CSRC, "tsvgp_kernels.hip"), os.path.join(CSRC, "tsvgp_chol.hip")]
HEADERS = [HEADER_PATH, os.path.join(CSRC, "tsvgp_chol.h")]
# per-source compiler switches: the small-matrix kernels keep their MFMA accumulators in VGPRs (the AGPR form the compiler
# picks for a one-wave-per-SIMD kernel ran the panel kernel's dependent MFMA chains ~1.5x slower, tools/diag2_lab.hip)
SOURCE_FLAGS = {"tsvgp_chol.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}

TILE = 128
MAX_BATCH = 32  # TSVGP_MAX_BATCH: latents per launch of the *_batched entry points
LIK_NONE, LIK_GAUSSIAN, LIK_BERNOULLI = 0, 1, 2
LIK_NOCROP = 0x100
LIK_MEANONLY = 0x200
KERNEL_SE, KERNEL_MATERN32, KERNEL_MATERN52 = 0, 2, 3
TRI_LOWER, TRI_UPPER, TRI_DENSE = 0, 1, 2
POTRF_SUBST = 1  # TSVGP_POTRF_SUBST
POTRF_RHS_UPPER = 2  # TSVGP_POTRF_RHS_UPPER
POTRF_DIAG_V1 = 4  # TSVGP_POTRF_DIAG_V1
POTRF_DIAG_V2 = 8  # TSVGP_POTRF_DIAG_V2
POTRF_FUSE = 16  # TSVGP_POTRF_FUSE
ABI_VERSION = 4  # TSVGP_ABI_VERSION of include/tsvgp_hip.h these prototypes were written for

_lib = None


class HipExtensionError(RuntimeError):
    """The HIP extension is missing, failed to load, or a kernel launch was rejected."""


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU): one object per source, compiled side
    by side, then linked into the one shared library the C-ABI lives in."""
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(p) for p in SOURCES + HEADERS)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I", os.path.join(_ROOT, "include"), "-I", CSRC]
    flags += os.environ.get("TSVGP_HIPCC_FLAGS", "").split()  # experiment builds (tools/): extra -D switches
    # build beside the target and rename: a concurrent loader (several ranks of one job) never maps a half-written file
    tag = f"{os.getpid()}.tmp"
    objs = [f"{src}.{tag}.o" for src in SOURCES]
    tmp = f"{LIB_PATH}.{tag}"
    procs = []
    try:
        for src, obj in zip(SOURCES, objs):
            cmd = [hipcc, *flags, *SOURCE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate()[0] for p in procs]
        if any(p.returncode != 0 for p in procs):
            raise HipExtensionError("hipcc failed:\n" + "\n".join(outs))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
        if verbose:
            print(" ".join(link))
        res = subprocess.run(link, capture_output=True, text=True)
        if res.returncode != 0:
            raise HipExtensionError("hipcc (link) failed:\n" + res.stdout + res.stderr)
        os.replace(tmp, LIB_PATH)
    finally:
        for f in objs + [tmp]:
            if os.path.exists(f):
                os.remove(f)
    return LIB_PATH


_PROTOTYPES = {
    # name: (restype, argtypes)
    "tsvgp_version": (c_char_p, []),
    "tsvgp_abi_version": (c_int, []),
    "tsvgp_site_accum_slots_f64": (c_int, []),
    "tsvgp_site_accum_slots_f32": (c_int, []),
    "tsvgp_se_fill_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p]),
    "tsvgp_se_fill_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p]),
    "tsvgp_kernel_fill_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p]),
    "tsvgp_kernel_fill_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p]),
    "tsvgp_kernel_fill_batched_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                              c_int64, c_int, c_void_p]),
    "tsvgp_kernel_fill_batched_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                              c_int64, c_int, c_void_p]),
    "tsvgp_gram_to_kernel_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_int64, c_int, c_int64, c_void_p]),
    "tsvgp_gram_to_kernel_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_int64, c_void_p]),
    "tsvgp_gram_to_gradw_parts": (c_int64, [c_int64, c_int]),
    "tsvgp_gram_to_gradw_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                                        c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "tsvgp_gram_to_gradw_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                                        c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "tsvgp_kernel_grad_rows": (c_int, []),
    "tsvgp_kernel_grad_dpad": (c_int, [c_int]),
    "tsvgp_kernel_grad_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                                      c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tsvgp_kernel_grad_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                                      c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tsvgp_trmm_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "tsvgp_trmm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "tsvgp_trmm_batched_f64": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_trmm_batched_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_moments_batched_f64": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_moments_batched_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_site_accum_batched_f64": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_site_accum_batched_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_moments_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_double, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_moments_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_double, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_lik_map_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                  c_int, c_void_p]),
    "tsvgp_lik_map_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                  c_int, c_void_p]),
    "tsvgp_site_accum_work_bytes_f64": (c_int64, [c_int, c_int, c_int]),
    "tsvgp_site_accum_work_bytes_f32": (c_int64, [c_int, c_int, c_int]),
    "tsvgp_site_accum_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_site_accum_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "tsvgp_potrf_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    "tsvgp_potrf_solve_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "tsvgp_flip_transpose_f64": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "tsvgp_potrf_inv_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_void_p]),
    "tsvgp_tri_copy_f64": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int, c_int, c_double, c_int, c_void_p]),
    "tsvgp_tri_copy_shift_f64": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int, c_int, c_double, c_double, c_int,
                                         c_void_p]),
    "tsvgp_site_target_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_double, c_void_p,
                                      c_double, c_void_p]),
    "tsvgp_site_update_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_double, c_double, c_void_p, c_double, c_void_p]),
    "tsvgp_site_beta_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "tsvgp_gemv_f64": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "tsvgp_keeper_run": (c_int, [c_void_p, c_double, c_int, c_void_p]),
    "tsvgp_keeper_signal": (c_int, [c_void_p, c_int, c_void_p]),
    "tsvgp_step_status_f64": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "tsvgp_sym_pack_f64": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "tsvgp_sym_unpack_f64": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "tsvgp_selftest_mfma_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "tsvgp_selftest_mfma_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
}


def exported_symbols():
    """Names every entry point of include/tsvgp_hip.h (kept in sync by tests/test_cabi.py)."""
    return sorted(_PROTOTYPES)


def lib():
    """Load (once) and return the ctypes handle.  Raises HipExtensionError if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("TSVGP_HIP_LIB", LIB_PATH)  # experiments (tools/): an alternative build of the same sources
    if not os.path.exists(path):
        raise HipExtensionError(
            f"HIP extension not built: {path} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the E-step."
        )
    try:
        handle = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the box
        raise HipExtensionError(f"cannot load {path}: {e}") from e
    # An older or newer build of the library (tools/ab_builds.sh builds other revisions; TSVGP_HIP_LIB points at them) may take
    # different argument lists under the same symbol names: refuse it rather than hand a kernel a shifted stream or pointer.
    try:
        handle.tsvgp_abi_version.restype = c_int
        abi = int(handle.tsvgp_abi_version())
    except AttributeError:
        abi = None
    if abi != ABI_VERSION:
        raise HipExtensionError(f"{path}: ABI version {abi}, these bindings are written for {ABI_VERSION} "
                                "(include/tsvgp_hip.h: TSVGP_ABI_VERSION); rebuild the library from this tree")
    missing = [name for name in _PROTOTYPES if not hasattr(handle, name)]
    if missing:
        raise HipExtensionError(f"{path} does not export {missing}: not a build of this tree (include/tsvgp_hip.h); rebuild it")
    for name, (restype, argtypes) in _PROTOTYPES.items():
        fn = getattr(handle, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def check(status: int, what: str):
    if status != 0:
        msg = {1: "invalid argument", 2: "kernel launch failure"}.get(status, f"status {status}")
        raise HipExtensionError(f"{what}: {msg}")


def suffix(dtype) -> str:
    import torch

    if dtype == torch.float64:
        return "f64"
    if dtype == torch.float32:
        return "f32"
    raise TypeError(f"unsupported compute dtype {dtype}; use torch.float64 or torch.float32")


def round_up(n: int, m: int = TILE) -> int:
    return (int(n) + m - 1) // m * m
