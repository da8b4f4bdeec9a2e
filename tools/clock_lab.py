#!/usr/bin/env python3
"""Round 5, verdict item 4: the clock the chip holds under the moments kernel, as a function of what runs between its launches.
Per-launch HIP-event times of tsvgp_moments_f64 (product library) in a loop of 60, with between two launches: nothing; one wave
asleep for G microseconds (the latency-bound M x M chain, as far as power goes); a kernel chaining fp64 MFMAs on registers on
`nwg` workgroups for G microseconds (tools/heater_lab.hip).     usage: clock_lab.py [rows] [gap_us]     (GPU box)"""
import ctypes, importlib, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
M = 1024
so = "/tmp/heater_lab.so"
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", root + "/tools/heater_lab.hip", "-o", so])
lab = ctypes.CDLL(so)
B = importlib.import_module("t-svgp_amd._backend")
lib = B.lib()
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lab.lab_sleep.argtypes = [ctypes.c_double, vp]
lab.lab_heat.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, vp]
lab.lab_duty.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
dev = "cuda:0"
dt = torch.float64
Np = (rows + 127) // 128 * 128
nwg = Np // 128
A = torch.randn(Np, M, dtype=dt, device=dev) / 32
T = torch.triu(torch.randn(1, M, M, dtype=dt, device=dev)) / 32
gam = torch.randn(M, 1, dtype=dt, device=dev)
Y = torch.randn(rows, 1, dtype=dt, device=dev)
g0 = torch.empty(Np, 1, dtype=dt, device=dev); g1 = torch.empty_like(g0)
vep = torch.empty(nwg, dtype=torch.float64, device=dev); npp = torch.empty(nwg, dtype=torch.int32, device=dev)
out = torch.zeros(8, dtype=dt, device=dev)
fn = lib.tsvgp_moments_f64
def moments():
    assert fn(A.data_ptr(), T.data_ptr(), gam.data_ptr(), Y.data_ptr(), 1e9, 1, 0.1, None, None, g0.data_ptr(),
              g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), rows, Np, M, 1, 1, None) == 0
def between(kind):
    if kind == "none":
        return
    if kind == "sleep":
        lab.lab_sleep(gap, None)
    elif kind.startswith("mfma"):
        lab.lab_heat(gap, int(kind[4:]), 0, out.data_ptr(), None)
    elif kind.startswith("fma"):
        lab.lab_heat(gap, int(kind[3:]), 1, out.data_ptr(), None)
    elif kind.startswith("duty"):  # duty:<nwg>:<threads>:<burst>:<nap>
        _, a, b, c, d = kind.split(":")
        lab.lab_duty(gap, int(a), int(b), int(c), int(d), out.data_ptr(), None)
    elif kind == "keeper":  # the product's keeper, flag never raised: leaves after `gap`
        assert lib.tsvgp_keeper_run(flag.data_ptr(), gap, 0, None) == 0
    elif kind == "sleep+heat":  # the last third of the gap heated
        lab.lab_sleep(gap * 2 / 3, None); lab.lab_heat(gap / 3, 256, 1, out.data_ptr(), None)
    elif kind == "heat+sleep":
        lab.lab_heat(gap * 2 / 3, 256, 1, out.data_ptr(), None); lab.lab_sleep(gap / 3, None)
    elif kind == "fill+sleep":  # what the step's M x M section looks like to the power management: a short fill, then little
        fill(); lab.lab_sleep(gap - 250, None)
    elif kind == "fill+keeper":
        fill(); assert lib.tsvgp_keeper_run(flag.data_ptr(), gap - 250, 0, None) == 0
flag = torch.zeros(16, dtype=torch.int32, device=dev)
estep = importlib.import_module("t-svgp_amd.estep")
eng = estep.EStepEngine(torch.float64, dev)
Xr = torch.randn(rows, 8, dtype=dt, device=dev); Zr = torch.randn(M, 8, dtype=dt, device=dev); ils = torch.ones(8, dtype=dt, device=dev)
def fill():
    eng.se_fill(Xr, Zr, ils, 0.001, A, B.KERNEL_SE)
K = 60
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
print(f"moments fp64, rows {rows}, M {M}; {gap:.0f} us between launches; ms per launch (HIP events), loop of {K} behind a synchronisation")
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ("none", "sleep", "mfma256", "mfma1024", "mfma128", "mfma64", "fma256", "fma1024", "none", "sleep")
for kind in kinds:
    torch.cuda.synchronize()
    for i in range(K):
        between(kind)
        ev[i][0].record()
        moments()
        ev[i][1].record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    print(f"{kind:18s}: first 3 {ms[0]:.3f} {ms[1]:.3f} {ms[2]:.3f} | launches 10-19 mean {ms[10:20].mean():.3f} | last 20 mean {ms[-20:].mean():.3f} min {ms[-20:].min():.3f} max {ms[-20:].max():.3f}", flush=True)
