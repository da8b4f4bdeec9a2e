#!/usr/bin/env python3
"""Diagnostic build (-DTSVGP_DIAG_STAMPS): where does a syrk workgroup spend its cycles?  Run on the GPU box."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "STAMPS"
so = f"/tmp/libtsvgp_diag_{mode}.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DTSVGP_DIAG_" + mode,
                       "-I", root + "/include", root + "/t-svgp_amd/csrc/tsvgp_kernels.hip", "-o", so])
lib = ctypes.CDLL(so)
rows, Mp, P = 262144, 1024, 1
dev = "cuda:0"
Bm = torch.randn(rows, Mp, dtype=torch.float64, device=dev)
g0 = torch.randn(rows, P, dtype=torch.float64, device=dev); g1 = -torch.rand(rows, P, dtype=torch.float64, device=dev) - 0.1
nsplit = 15
lib.tsvgp_site_accum_work_bytes_f64.restype = ctypes.c_int64
nb = lib.tsvgp_site_accum_work_bytes_f64(Mp, P, nsplit)
nwg = 28 * nsplit + 8 * ((23 * nsplit + 31) // 32)
work = torch.zeros(nb + nwg * 4 * 8 * 8 + 4096, dtype=torch.uint8, device=dev)
acc2 = torch.empty(P, Mp, Mp, dtype=torch.float64, device=dev); acc1 = torch.empty(P, Mp, dtype=torch.float64, device=dev)
vp = ctypes.c_void_p
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for rep in range(4):
    if rep == 1: e0.record()
    st = lib.tsvgp_site_accum_f64(vp(Bm.data_ptr()), vp(g0.data_ptr()), vp(g1.data_ptr()), vp(acc2.data_ptr()), vp(acc1.data_ptr()), vp(work.data_ptr()),
                                  ctypes.c_int64(rows), Mp, P, nsplit, None)
    assert st == 0
e1.record(); torch.cuda.synchronize()
print("mode", mode, "launch ms (syrk + reduce):", e0.elapsed_time(e1) / 3)
per_p = 28 * nsplit + 8 * ((23 * nsplit + 31) // 32)
off = per_p * 128 * 128 * 8 + P * ((23 * nsplit + 31) // 32) * Mp * 8
dbg = work[off: off + nwg * 4 * 8 * 8].view(torch.int64).cpu().numpy().reshape(nwg, 4, 8)
for kind in (0, 1):
    sel = dbg[dbg[:, 0, 7] == kind]
    if len(sel) == 0: continue
    ch = sel[:, :, 6].mean()
    seg = sel[:, :, :4].mean(axis=(0, 1)) / ch
    tot = sel[:, :, 4].mean() / ch
    clk = (sel[:, :, 4] / (sel[:, :, 5] * 10.0)).mean()
    print(f"{'diag' if kind else 'off-diag'} WGs={len(sel)} chunks/WG={ch:.0f}  cycles per chunk: load-issue {seg[0]:.0f}  mfma {seg[1]:.0f}  stage(vmcnt+ds_write) {seg[2]:.0f}  barrier {seg[3]:.0f}  total/chunk {tot:.0f}  in-kernel clock {clk:.3f} GHz")
    print("   per-wave mfma segment:", (sel[:, :, 1].mean(axis=0) / ch).round(0), " barrier:", (sel[:, :, 3].mean(axis=0) / ch).round(0))

if mode == "CLOCK":
    st0 = dbg[:, 0, 0].astype(np.float64); en = dbg[:, 0, 1].astype(np.float64)
    t0 = st0.min()
    st_us = (st0 - t0) / 100.0; en_us = (en - t0) / 100.0
    print("WG start times (us): min %.1f  median %.1f  p90 %.1f  max %.1f" % (st_us.min(), np.median(st_us), np.percentile(st_us, 90), st_us.max()))
    print("WG end   times (us): min %.1f  median %.1f  p90 %.1f  max %.1f" % (en_us.min(), np.median(en_us), np.percentile(en_us, 90), en_us.max()))
    late = st_us > 100
    print("WGs starting later than 100 us:", int(late.sum()), "of", len(st_us))
    hw = dbg[:, 0, 2]; xcc = dbg[:, 0, 3] & 0xF
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    import collections
    cnt = collections.Counter(key.tolist())
    print("distinct (xcc,se,sh,cu):", len(cnt), " WGs per CU histogram:", sorted(collections.Counter(cnt.values()).items()))
    print("WGs per XCC:", sorted(collections.Counter(xcc.tolist()).items()))
    dur = en_us - st_us
    print("WG duration (us): min %.0f median %.0f max %.0f" % (dur.min(), np.median(dur), dur.max()))
if mode == "CLOCK":
    kind = dbg[:, 0, 7]
    for kd in (0, 1):
        d = dur[kind == kd]
        print(("diag" if kd else "off-diag"), "durations us: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f  (n=%d)" % (d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max(), len(d)))
    # pairing: which kinds share a CU
    import collections
    bycu = collections.defaultdict(list)
    for i, kk in enumerate(key.tolist()): bycu[kk].append(i)
    pair_d = collections.defaultdict(list)
    for kk, idxs in bycu.items():
        kinds = tuple(sorted(int(kind[i]) for i in idxs))
        for i in idxs: pair_d[(kinds, int(kind[i]))].append(dur[i])
    for (kinds, kd), v in sorted(pair_d.items()):
        print("CU holds kinds", kinds, "-> WG kind", kd, "median %.0f max %.0f n=%d" % (np.median(v), max(v), len(v)))
    xs = collections.defaultdict(list)
    for i in range(len(dur)): xs[int(xcc[i])].append(dur[i])
    print("per XCC max duration:", {k: int(max(v)) for k, v in sorted(xs.items())})
