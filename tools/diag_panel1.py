#!/usr/bin/env python3
"""Where the one-workgroup-per-CU moments kernel (panel1_kernel) spends the time its MFMAs do not: per-workgroup stamps from a
-DTSVGP_DIAG_PANEL1 build (s_memrealtime at entry / exit, placement from HW_ID / XCC_ID, shader cycles of the prologue, the column
tiles' epilogues and the tail) turned into (a) the share of a workgroup's cycles outside the chunk stream, (b) the gap a CU sits
empty between one workgroup's exit and the next one's entry (the turnover a persistent row-panel loop would remove), (c) the in-kernel
clock and the last dispatch round.     usage: diag_panel1.py [rows] [M] [f64|f32] [extra -D flags ...]     (GPU box)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prec = sys.argv[3] if len(sys.argv) > 3 else "f64"
flags = sys.argv[4:]
so = "/tmp/libtsvgp_diag_panel1_%d.so" % (abs(hash(" ".join(flags))) % 100000)
if not os.path.exists(so):
    # (two objects, as _backend.build_library: the factorisation's second source has its own code-generation flag)
    cc = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I", root + "/include"]
    o1, o2 = so + ".kernels.o", so + ".chol.o"
    subprocess.check_call(cc + ["-DTSVGP_DIAG_PANEL1", *flags, "-c", root + "/t-svgp_amd/csrc/tsvgp_kernels.hip", "-o", o1])
    subprocess.check_call(cc + ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-c", root + "/t-svgp_amd/csrc/tsvgp_chol.hip", "-o", o2])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", o1, o2, "-o", so])
lib = ctypes.CDLL(so)
vp, i64 = ctypes.c_void_p, ctypes.c_int64
dev = "cuda:0"
dt = torch.float64 if prec == "f64" else torch.float32
Np = (rows + 127) // 128 * 128
nwg = Np // 128
A = (torch.randn(Np, M, dtype=torch.float64, device=dev) / 32).to(dt)
T = (torch.triu(torch.randn(1, M, M, dtype=torch.float64, device=dev)) / 32).to(dt)
gam = torch.randn(M, 1, dtype=torch.float64, device=dev).to(dt)
Y = torch.randn(rows, 1, dtype=torch.float64, device=dev).to(dt)
g0 = torch.empty(Np, 1, dtype=dt, device=dev); g1 = torch.empty_like(g0)
vep = torch.empty(nwg, dtype=torch.float64, device=dev); npp = torch.empty(nwg, dtype=torch.int32, device=dev)
fn = getattr(lib, "tsvgp_moments_" + prec)
fn.argtypes = [vp, vp, vp, vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, vp, vp, vp, vp, vp, vp, i64, i64,
               ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
lib.tsvgp_diag_panel1_stamps.argtypes = [vp]
def run():
    assert fn(A.data_ptr(), T.data_ptr(), gam.data_ptr(), Y.data_ptr(), 1e9, 1, 0.1, None, None, g0.data_ptr(),
              g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), rows, Np, M, 1, 1, None) == 0
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
peak = 78.6 if prec == "f64" else 157.3
print(f"{prec} rows {rows} M {M} flags {flags}: moments {ms:.3f} ms  {rows * M * (M + 1) / ms / 1e9:.2f} TFLOP/s = {rows * M * (M + 1) / ms / 1e9 / peak:.3f} of {peak}")
dbg = torch.zeros(8 + nwg * 8, dtype=torch.int64, device=dev)
dbg[0] = nwg
assert lib.tsvgp_diag_panel1_stamps(dbg.data_ptr()) == 0
run(); torch.cuda.synchronize()
assert lib.tsvgp_diag_panel1_stamps(None) == 0
d = dbg.cpu().numpy()[8:].reshape(nwg, 8)
assert (d[:, 1] > 0).all(), "a workgroup left no stamp"
t0, t1 = d[:, 0].astype(np.float64), d[:, 1].astype(np.float64)
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0  # microseconds (s_memrealtime: 100 MHz)
cyc, pro, epi, tail = (d[:, i].astype(np.float64) for i in (3, 4, 5, 6))
dur = t1 - t0
clk = cyc / dur / 1e3
print("in-kernel clock (GHz) over workgroups: p10 %.3f median %.3f p90 %.3f" % tuple(np.percentile(clk, [10, 50, 90])))
print("workgroup duration (us): min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f;  kernel span %.0f us" % (
    dur.min(), *np.percentile(dur, [10, 50, 90]), dur.max(), t1.max()))
med = np.median(cyc)
print("wave-0 shader cycles, median workgroup: total %.0f | prologue (entry -> first fragments) %.0f = %.2f %% | tile epilogues %.0f = %.2f %% | "
      "tail (row sums, likelihood map, stores) %.0f = %.2f %%" % (med, np.median(pro), 100 * np.median(pro) / med, np.median(epi),
                                                                  100 * np.median(epi) / med, np.median(tail), 100 * np.median(tail) / med))
# MFMA cycles of a workgroup's wave: upper form, nt column tiles; tile it: (nt - it - 1) full k-tiles of 8 chunks x 64 MFMAs + the diagonal
# k-tile's 288 MFMAs (fp64: 16 passes of 4 cycles = 64 cycles each... v_mfma_f64_16x16x4: 32 cycles at 4 per pass on gfx950; stated as measured)
nt = M // 128
n_mfma = (nt * (nt - 1) // 2) * 8 * 64 + nt * 288
print("MFMAs per wave and workgroup: %d;  median cycles per MFMA slot: %.2f" % (n_mfma * (1 if prec == "f64" else 2), med / (n_mfma * (1 if prec == "f64" else 2))))
hw, xcc = d[:, 2] & 0xFFFFFFFF, (d[:, 2] >> 32) & 0xF
cu_key = xcc * 65536 + ((hw >> 8) & 0xFF)
keys = np.unique(cu_key)
gaps, first_start, n_per = [], [], []
empty_time = 0.0
for k in keys:
    sel = np.where(cu_key == k)[0]
    o = sel[np.argsort(t0[sel])]
    n_per.append(len(o))
    first_start.append(t0[o[0]])
    # several CUs may share a key: treat the key as a pool and measure the time no workgroup of the key is resident
    ev = sorted([(t0[i], 1) for i in o] + [(t1[i], -1) for i in o])
    occ, last, peak_occ = 0, ev[0][0], 0
    for tt, dl in ev:
        if occ == 0 and tt > last: gaps.append(tt - last)
        occ += dl; peak_occ = max(peak_occ, occ); last = tt
gaps = np.array(gaps) if gaps else np.zeros(1)
print(f"{len(keys)} distinct (xcc, HW_ID[15:8]) keys, workgroups per key min {min(n_per)} max {max(n_per)}; first entry per key: median {np.median(first_start):.1f} us, max {max(first_start):.1f} us")
# exit -> next entry on the same CU: match each exit with the nearest later entry on its key
turn = []
for k in keys:
    sel = np.where(cu_key == k)[0]
    starts = np.sort(t0[sel])
    for i in sel:
        j = np.searchsorted(starts, t1[i])
        if j < len(starts): turn.append(starts[j] - t1[i])
turn = np.array(turn)
print("turnover, exit of a workgroup -> the next entry on its key (us): p10 %.2f  median %.2f  p90 %.2f  max %.2f  (n = %d)" % (
    *np.percentile(turn, [10, 50, 90]), turn.max(), len(turn)))
print("turnover share of a CU's time: median turnover / (median duration + median turnover) = %.2f %%" % (
    100 * np.median(turn) / (np.median(dur) + np.median(turn))))
order = np.argsort(t0)
ncu = 256
nr = (nwg + ncu - 1) // ncu
for r in sorted(set([0, 1, nr // 2, nr - 2, nr - 1])):
    sel = order[r * ncu:(r + 1) * ncu]
    if len(sel) == 0: continue
    print(f"dispatch round {r:2d} ({len(sel):3d} workgroups): entry {t0[sel].min():8.1f}..{t0[sel].max():8.1f} us  duration median {np.median(dur[sel]):6.1f} us  exit {t1[sel].min():8.1f}..{t1[sel].max():8.1f}")
