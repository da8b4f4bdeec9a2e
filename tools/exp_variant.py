#!/usr/bin/env python3
"""Builds the kernel library with extra -D flags into /tmp and times trmm/moments at a few row counts (GPU box).
usage: exp_variant.py "<flags>" rows [rows...]"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = sys.argv[1].split()
so = "/tmp/libtsvgp_exp_%d.so" % (abs(hash(sys.argv[1])) % 100000)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", *flags,
                       "-I", root + "/include", root + "/t-svgp_amd/csrc/tsvgp_kernels.hip", "-o", so])
lib = ctypes.CDLL(so)
vp = ctypes.c_void_p
dev = "cuda:0"
M = 1024
for rows in [int(r) for r in sys.argv[2:]]:
    A = torch.randn(rows, M, dtype=torch.float64, device=dev)
    T = torch.tril(torch.randn(M, M, dtype=torch.float64, device=dev)) / 32
    C = torch.empty_like(A)
    def run():
        assert lib.tsvgp_trmm_f64(vp(A.data_ptr()), vp(T.data_ptr()), vp(C.data_ptr()), ctypes.c_int64(rows), M, 0, None) == 0
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"flags={sys.argv[1]!r:28s} rows={rows:8d} trmm {ms:8.3f} ms  {rows * M * (M + 1) / ms / 1e9:6.2f} TFLOP/s")
