#!/usr/bin/env python3
"""Diagnostic build: in-kernel cycle stamps of potrf_diag_kernel's phases (GPU box).  usage: diag_potrf.py [k]: the block
column whose factoring workgroup is stamped (default 0)."""
import ctypes, os, subprocess, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = "/tmp/libtsvgp_diag_potrf.so"
kblk = int(sys.argv[1]) if len(sys.argv) > 1 else 0
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DTSVGP_DIAG_POTRF", f"-DTSVGP_DIAG_POTRF_K={kblk}", *sys.argv[2:],
                       "-I", root + "/include", root + "/t-svgp_amd/csrc/tsvgp_kernels.hip", "-o", so])
lib = ctypes.CDLL(so)
M = 128 * (kblk + 2)  # one block column more than the stamped one: it also builds its inverse
A = torch.randn(M, M, dtype=torch.float64, device="cuda:0"); A = A @ A.T / M + torch.eye(M, dtype=torch.float64, device="cuda:0")
info = torch.zeros(1, dtype=torch.int32, device="cuda:0"); work = torch.zeros(128 * 128, dtype=torch.float64, device="cuda:0")
vp = ctypes.c_void_p
# the stamped build overwrites the head of inv(L_00) in `work` with its stamps, so the factor itself is wrong here
for _ in range(3):
    W = A.clone()
    assert lib.tsvgp_potrf_f64(vp(W.data_ptr()), M, M, 1, ctypes.c_int64(M * M), vp(info.data_ptr()), vp(work.data_ptr()), 0, None) == 0
torch.cuda.synchronize()
raw = work[:128].view(torch.int64).cpu().numpy()
st = raw[:40]
n = int(st[0]); st = st[1:1 + n]
names = ["load"]
nsb = (n - 5) // 2  # sub-blocks per 128-block (TSVGP_CHOL_SB = 16 -> 8)
for s_ in range(nsb):
    names += [f"s{s_}: factor + row solves + inverse || previous update", f"s{s_}: update of the next block column"]
names += ["store L", "assemble inverse (3 levels)", "store inverse"]
for i in range(n - 1):
    print(f"{names[i] if i < len(names) else '?':52s} {int(st[i + 1] - st[i]):8d} ticks  {(st[i + 1] - st[i]) / 2.3e3:7.2f} us")
print(f"{'total':32s} {int(st[-1] - st[0]):8d} ticks  {(st[-1] - st[0]) / 2.3e3:7.2f} us   (s_memtime ticks at ~2.3 GHz)")

# stamps inside the factor pass of sub-blocks 0 and 5 (wave 0): start, then per 4-column step: pivots done, own row done,
# remaining columns updated
for name, off in (("sub-block 0 (3 waves)", 64), ("sub-block 5 (1 wave)", 96)):
    f = raw[off:off + 14]
    print(name + ": " + " ".join(str(int(f[i + 1] - f[i])) for i in range(12)) + "  ticks between stamps")
