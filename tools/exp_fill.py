#!/usr/bin/env python3
"""Builds the kernel library with extra -D flags into /tmp and times tsvgp_se_fill_f64 at N=1e6, M=1024, D=8 (GPU box).
usage: exp_fill.py "<flags>" """
import ctypes, os, subprocess, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = sys.argv[1].split() if len(sys.argv) > 1 else []
so = "/tmp/libtsvgp_expf_%d.so" % (abs(hash(" ".join(flags))) % 100000)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", *flags,
                       "-I", root + "/include", root + "/t-svgp_amd/csrc/tsvgp_kernels.hip", "-o", so])
lib = ctypes.CDLL(so)
vp = ctypes.c_void_p
N, M, D = 1_000_000, 1024, 8
Np = (N + 127) // 128 * 128
X = torch.randn(N, D, dtype=torch.float64, device="cuda:0"); Z = X[:M].clone().contiguous()
inv_ls = torch.ones(D, dtype=torch.float64, device="cuda:0")
K = torch.empty(Np, M, dtype=torch.float64, device="cuda:0")
def run():
    assert lib.tsvgp_se_fill_f64(vp(X.data_ptr()), vp(Z.data_ptr()), vp(inv_ls.data_ptr()), ctypes.c_double(1.0), vp(K.data_ptr()),
                                 ctypes.c_int64(N), M, D, ctypes.c_int64(M), None) == 0
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"flags={' '.join(flags)!r:24s} se_fill {ms:7.3f} ms  {Np * M * 8 / ms / 1e6:7.1f} GB/s written")
if not flags:
    ref = torch.exp(-0.5 * torch.cdist(X[:4096], Z) ** 2)
    print("max rel err vs torch (first 4096 rows):", float(((K[:4096] - ref).abs() / ref).max()))
