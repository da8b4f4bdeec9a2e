#!/usr/bin/env python3
"""Where the moments kernel's MFMA-idle time is: per-workgroup start / end stamps (s_memrealtime, 100 MHz) and placement
(XCC, SE/CU from HW_ID) from a -DTSVGP_DIAG_PANEL build, turned into (a) workgroups resident per CU over time, (b) the
distribution of workgroup durations by dispatch round.  usage: diag_panel.py [rows] [extra -D flags ...]   (GPU box)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
flags = sys.argv[2:]
so = "/tmp/libtsvgp_diag_panel_%d.so" % (abs(hash(" ".join(flags))) % 100000)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DTSVGP_DIAG_PANEL",
                       *flags, "-I", root + "/include", root + "/t-svgp_amd/csrc/tsvgp_kernels.hip", "-o", so])
lib = ctypes.CDLL(so)
vp, i64 = ctypes.c_void_p, ctypes.c_int64
dev, M = "cuda:0", 1024
Np = (rows + 127) // 128 * 128
nwg = Np // 128
A = torch.randn(Np, M, dtype=torch.float64, device=dev) / 32
T = torch.triu(torch.randn(1, M, M, dtype=torch.float64, device=dev)) / 32
gam = torch.randn(M, 1, dtype=torch.float64, device=dev)
Y = torch.randn(rows, 1, dtype=torch.float64, device=dev)
g0 = torch.empty(Np, 1, dtype=torch.float64, device=dev); g1 = torch.empty_like(g0)
vep = torch.empty(nwg, dtype=torch.float64, device=dev); npp = torch.empty(nwg, dtype=torch.int32, device=dev)
# stamp buffer: an 8-word header whose first word is the number of 8-word slots behind it (the kernel checks its slot index
# against it), then one slot per workgroup
dbg = torch.zeros(8 + 2 * nwg * 8, dtype=torch.int64, device=dev)  # header, one bank of per-workgroup slots, a second bank
dbg[0] = nwg
lib.tsvgp_moments_f64.argtypes = [vp, vp, vp, vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, vp, vp, vp, vp, vp, vp, i64, i64,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
def run(d):
    assert lib.tsvgp_moments_f64(A.data_ptr(), T.data_ptr(), gam.data_ptr(), Y.data_ptr(), 1e9, 1, 0.1, d, None, g0.data_ptr(),
                                 g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), rows, Np, M, 1, 1, None) == 0
for _ in range(3): run(None)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run(None)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"flags {flags} rows {rows}: moments {ms:.3f} ms  {rows * M * (M + 1) / ms / 1e9:.2f} TFLOP/s")
run(dbg.data_ptr()); torch.cuda.synchronize()
d2 = dbg.cpu().numpy()[8 + nwg * 8:].reshape(nwg, 8)
d = dbg.cpu().numpy()[8:8 + nwg * 8].reshape(nwg, 8)
assert (d[:, 1] > 0).all(), "a workgroup left no stamp"
# wave 0's shader cycles by phase (median over workgroups), against the cycles its MFMAs need when two workgroups share a CU
# (2 x 64 cycles per v_mfma_f64_16x16x4): per workgroup 28 full k-tiles x 8 chunks x 64 MFMAs, 8 diagonal k-tiles x 288 MFMAs
ph = np.median(d[:, 4:8].astype(np.float64), axis=0)
tot = np.median(d[:, 3].astype(np.float64))
nt = M // 128
full_mfma, diag_mfma = (nt * (nt - 1) // 2) * 8 * 64, nt * 288
print("wave-0 cycles by phase (median workgroup): full chunks %.0f (MFMA x2: %.0f, use %.3f) | diagonal chunks %.0f (MFMA x2: %.0f, use %.3f) | "
      "tile prologues %.0f | tile epilogues %.0f | rest %.0f | total %.0f" % (
          ph[0], 2 * 64 * full_mfma, 2 * 64 * full_mfma / max(ph[0], 1), ph[1], 2 * 64 * diag_mfma, 2 * 64 * diag_mfma / max(ph[1], 1),
          ph[2], ph[3], tot - ph.sum(), tot))
print("   rest = before the first tile %.0f + after the last tile %.0f (median cycles)" % (np.median(d2[:, 0]), np.median(d2[:, 1])))
t0, t1 = d[:, 0].astype(np.float64), d[:, 1].astype(np.float64)
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0  # microseconds
hw, xcc = d[:, 2] & 0xFFFFFFFF, (d[:, 2] >> 32) & 0xF
clk = d[:, 3].astype(np.float64) / ((d[:, 1] - d[:, 0]).astype(np.float64) / 100.0) / 1e3  # GHz: shader cycles / wall
print("in-kernel clock (GHz) over workgroups: p10 %.3f median %.3f p90 %.3f" % tuple(np.percentile(clk, [10, 50, 90])))
mfma_cycles_pair = 2 * 16640 * 64  # two resident workgroups' MFMAs per SIMD (8 tiles, M = 1024, upper form), 64 cycles each
print("steady-state MFMA pipe use = cycles of two workgroups' MFMAs / median workgroup cycles: %.3f" % (mfma_cycles_pair / np.median(d[:, 3])))
cu_key = xcc * 65536 + ((hw >> 8) & 0xFF)  # (xcc, se/sh/cu bits of HW_ID)
keys = np.unique(cu_key)
print(f"kernel span {t1.max():.0f} us; {len(keys)} distinct (xcc, HW_ID[15:8]) keys; workgroups {nwg}")
dur = t1 - t0
order = np.argsort(t0)
print("workgroup duration (us): min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (dur.min(), *np.percentile(dur, [10, 50, 90]), dur.max()))
# residency per CU key over time
grid = np.linspace(0, t1.max(), 2001)
occ_hist = np.zeros(8)
for k in keys:
    sel = cu_key == k
    ev = np.concatenate([np.stack([t0[sel], np.ones(sel.sum())], 1), np.stack([t1[sel], -np.ones(sel.sum())], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    occ, last = 0, 0.0
    for tt, dlt in ev:
        occ_hist[min(int(occ), 7)] += tt - last
        last, occ = tt, occ + dlt
    occ_hist[0] += t1.max() - last
tot = occ_hist.sum()
print("time share by resident workgroups per CU key: " + "  ".join(f"{i}: {occ_hist[i] / tot:.3f}" for i in range(5)))
# by dispatch round (512 slots)
for r in range(0, (nwg + 511) // 512):
    sel = order[r * 512:(r + 1) * 512]
    print(f"round {r:2d}: start {t0[sel].min():8.0f}..{t0[sel].max():8.0f} us  dur median {np.median(dur[sel]):7.0f} us  end {t1[sel].min():8.0f}..{t1[sel].max():8.0f}")
