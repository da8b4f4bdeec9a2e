#!/usr/bin/env python3
"""Diagnostic build: in-kernel cycle stamps of potrf_diag_kernel's phases (GPU box)."""
import ctypes, os, subprocess, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = "/tmp/libtsvgp_diag_potrf.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DTSVGP_DIAG_POTRF",
                       "-I", root + "/include", root + "/t-svgp_amd/csrc/tsvgp_kernels.hip", "-o", so])
lib = ctypes.CDLL(so)
M = 128
A = torch.randn(M, M, dtype=torch.float64, device="cuda:0"); A = A @ A.T / M + torch.eye(M, dtype=torch.float64, device="cuda:0")
info = torch.zeros(1, dtype=torch.int32, device="cuda:0"); work = torch.zeros(128 * 128, dtype=torch.float64, device="cuda:0")
vp = ctypes.c_void_p
for _ in range(3):
    W = A.clone()
    assert lib.tsvgp_potrf_f64(vp(W.data_ptr()), M, M, 1, ctypes.c_int64(M * M), vp(info.data_ptr()), vp(work.data_ptr()), None) == 0
torch.cuda.synchronize()
st = work[:6].view(torch.int64).cpu().numpy()
names = ["load", "phase A (factor)", "store L + dinv", "phase B (inverse)", "store inverse"]
for i, n in enumerate(names):
    print(f"{n:22s} {int(st[i + 1] - st[i]):8d} cycles  {(st[i + 1] - st[i]) / 2.39e3:7.1f} us")
