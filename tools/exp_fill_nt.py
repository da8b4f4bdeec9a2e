#!/usr/bin/env python3
"""A/B of the K(X, Z) fill with plain vs non-temporal stores (-DTSVGP_FILL_STREAM=0 / 1; the library picks by output size), alone and followed by the moments kernel
(does bypassing the caches on the way out cost the reader?).  usage: exp_fill_nt.py   (GPU box)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp, i64 = ctypes.c_void_p, ctypes.c_int64
dev, N, M, D = "cuda:0", 1_000_000, 1024, 8
Np = (N + 127) // 128 * 128
X = torch.randn(N, D, dtype=torch.float64, device=dev); Z = X[:M].clone()
il = torch.ones(D, dtype=torch.float64, device=dev)
K = torch.empty(Np, M, dtype=torch.float64, device=dev)
T = torch.triu(torch.randn(1, M, M, dtype=torch.float64, device=dev)) / 32
gam = torch.randn(M, 1, dtype=torch.float64, device=dev); Y = torch.randn(N, 1, dtype=torch.float64, device=dev)
g0 = torch.empty(Np, 1, dtype=torch.float64, device=dev); g1 = torch.empty_like(g0)
vep = torch.empty(Np // 128, dtype=torch.float64, device=dev); npp = torch.empty(Np // 128, dtype=torch.int32, device=dev)
for flags in (["-DTSVGP_FILL_STREAM=0"], ["-DTSVGP_FILL_STREAM=1"]):
    so = "/tmp/libtsvgp_fillnt_%d.so" % len(flags)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", *flags,
                           "-I", root + "/include", root + "/t-svgp_amd/csrc/tsvgp_kernels.hip", "-o", so])
    lib = ctypes.CDLL(so)
    lib.tsvgp_se_fill_f64.argtypes = [vp, vp, vp, ctypes.c_double, vp, i64, ctypes.c_int, ctypes.c_int, i64, vp]
    lib.tsvgp_moments_f64.argtypes = [vp, vp, vp, vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, vp, vp, vp, vp, vp, vp, i64, i64,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    fill = lambda: lib.tsvgp_se_fill_f64(X.data_ptr(), Z.data_ptr(), il.data_ptr(), 1.0, K.data_ptr(), N, M, D, M, None)
    mom = lambda: lib.tsvgp_moments_f64(K.data_ptr(), T.data_ptr(), gam.data_ptr(), Y.data_ptr(), 1e9, 1, 0.1, None, None, g0.data_ptr(),
                                        g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), N, Np, M, 1, 1, None)
    def timeit(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_f = timeit(fill); t_m = timeit(mom, 5); t_fm = timeit(lambda: (fill(), mom()), 5)
    print(f"flags {flags}: fill {t_f:.3f} ms ({Np * M * 8 / t_f / 1e9:.2f} TB/s)  moments {t_m:.3f} ms  fill+moments {t_fm:.3f} ms", flush=True)
