#!/usr/bin/env python3
"""Builds the kernel library with extra -D flags into /tmp and times tsvgp_moments (UPPER, Gaussian) back to back at a
few row counts (GPU box): separates the tail effect (rows = 983040 is exactly 15 rounds of 512 resident workgroups)
and the mean-GEMV pre-pass (-DTSVGP_EXP_NOGEMV).   usage: exp_moments.py "<flags>" rows [rows...]
Environment: EXP_DTYPE=f32 times tsvgp_moments_f32 (e.g. with -DTSVGP_PANEL_KC_F32=16), EXP_LIK=2 the Bernoulli map."""
import ctypes, os, subprocess, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = sys.argv[1].split()
so = "/tmp/libtsvgp_expm_%d.so" % (abs(hash(sys.argv[1])) % 100000)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", *flags,
                       "-I", root + "/include", root + "/t-svgp_amd/csrc/tsvgp_kernels.hip", "-o", so])
lib = ctypes.CDLL(so)
vp = ctypes.c_void_p
dev = "cuda:0"
M = 1024
dt = torch.float32 if os.environ.get("EXP_DTYPE", "f64") == "f32" else torch.float64
fn = lib.tsvgp_moments_f32 if dt == torch.float32 else lib.tsvgp_moments_f64
lik = int(os.environ.get("EXP_LIK", "1"))
for rows in [int(r) for r in sys.argv[2:]]:
    Np = (rows + 127) // 128 * 128
    A = torch.randn(Np, M, dtype=dt, device=dev) * 0.01
    T = torch.triu(torch.randn(M, M, dtype=dt, device=dev)) / 32
    gam = torch.randn(M, 1, dtype=dt, device=dev)
    Y = (torch.randn(rows, 1, dtype=dt, device=dev) > 0).to(dt) if lik == 2 else torch.randn(rows, 1, dtype=dt, device=dev)
    g0 = torch.empty(Np, 1, dtype=dt, device=dev); g1 = torch.empty_like(g0)
    vep = torch.empty(Np // 128, dtype=torch.float64, device=dev); npp = torch.empty(Np // 128, dtype=torch.int32, device=dev)
    def run():
        assert fn(vp(A.data_ptr()), vp(T.data_ptr()), vp(gam.data_ptr()), vp(Y.data_ptr()), ctypes.c_double(1e9), lik,
                                     ctypes.c_double(0.1), None, None, vp(g0.data_ptr()), vp(g1.data_ptr()), vp(vep.data_ptr()),
                                     vp(npp.data_ptr()), ctypes.c_int64(rows), ctypes.c_int64(Np), M, 1, 1, None) == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{os.environ.get('EXP_DTYPE', 'f64')} lik={lik} flags={sys.argv[1]!r:24s} rows={rows:8d} ({Np / 128 / 512:6.2f} rounds) moments {ms:8.3f} ms  {rows * M * (M + 1) / ms / 1e9:6.2f} TFLOP/s", flush=True)
