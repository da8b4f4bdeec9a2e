#!/usr/bin/env python3
"""Experiment: hide the tail of the moments launch behind the head of the site-sum launch.  The moments kernel's 7813 equal
workgroups are 15.26 dispatch rounds; its last, partial round leaves most CUs idle.  Variant: moments over the first R1 rows
(whole rounds) -> [moments over the rest || site sums over the first R1 rows (side stream)] -> site sums over the rest ->
two partial results added.  Prints the time of the plain sequence and of the variant.   (GPU box)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("t-svgp_amd")
estep = importlib.import_module("t-svgp_amd.estep")
B = pkg._backend
dev = torch.device("cuda", 0)
eng = estep.EStepEngine(torch.float64, dev)
N, M, P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 1024, 1
Np = B.round_up(N)
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(Np, M, generator=g, device=dev, dtype=torch.float64) / 32
Tm = torch.triu(torch.randn(1, M, M, generator=g, device=dev, dtype=torch.float64)) / 32
gam = torch.randn(M, 1, generator=g, device=dev, dtype=torch.float64)
Y = torch.randn(N, 1, generator=g, device=dev, dtype=torch.float64)
g0 = torch.empty(Np, 1, dtype=torch.float64, device=dev); g1 = torch.empty_like(g0)
vep = torch.empty(Np // 128, dtype=torch.float64, device=dev); npp = torch.empty(Np // 128, dtype=torch.int32, device=dev)
mom, syrk, wb = eng._fn("tsvgp_moments"), eng._fn("tsvgp_site_accum"), eng._fn("tsvgp_site_accum_work_bytes")
slots = eng.slots()
side = torch.cuda.Stream(dev)

def moments(r0, r1, stream):
    n, npad = min(r1, N) - r0, r1 - r0
    assert mom(A[r0:].data_ptr(), Tm.data_ptr(), gam.data_ptr(), Y[r0:].data_ptr(), 1e9, 1, 0.1, None, None, g0[r0:].data_ptr(),
               g1[r0:].data_ptr(), vep[r0 // 128:].data_ptr(), npp[r0 // 128:].data_ptr(), n, npad, M, P, 1, stream.cuda_stream) == 0

bufs = {}
def sums(r0, r1, stream, key):
    rows = r1 - r0
    ns = max(1, min(eng.choose_nsplit(M, P, rows), rows // 16))
    if key not in bufs:
        bufs[key] = (torch.empty(int(wb(M, P, ns)), dtype=torch.uint8, device=dev), torch.empty(P, M, M, dtype=torch.float64, device=dev),
                     torch.empty(P, M, dtype=torch.float64, device=dev))
    w, a2, a1 = bufs[key]
    assert syrk(A[r0:].data_ptr(), g0[r0:].data_ptr(), g1[r0:].data_ptr(), a2.data_ptr(), a1.data_ptr(), w.data_ptr(), rows, M, P, ns,
                stream.cuda_stream) == 0
    return a2, a1

main = torch.cuda.current_stream(dev)
def plain():
    moments(0, Np, main)
    return sums(0, Np, main, "all")

def variant(R1):
    moments(0, R1, main)
    evA = torch.cuda.Event(); evA.record(main)
    moments(R1, Np, main)
    with torch.cuda.stream(side):
        side.wait_event(evA)
        a2, a1 = sums(0, R1, side, "A")
        evS = torch.cuda.Event(); evS.record(side)
    b2, b1 = sums(R1, Np, main, "B")
    main.wait_event(evS)
    return a2 + b2, a1 + b1

def timeit(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out

t0, ref = timeit(plain)
print(f"N = {N}: moments + site sums, plain sequence: {t0:.3f} ms", flush=True)
nwg = Np // 128
for rounds_back in (0, 1, 2):
    R1 = ((nwg // slots) - rounds_back) * slots * 128
    if R1 <= 0 or R1 >= Np: continue
    t1, out = timeit(lambda: variant(R1))
    err = float((out[0] - ref[0]).abs().max() / ref[0].abs().max())
    print(f"   split at row {R1} ({R1 // 128} panels = {R1 // 128 / slots:.2f} rounds; rest {nwg - R1 // 128} panels): {t1:.3f} ms  ({t1 - t0:+.3f})  rel diff of acc2 {err:.1e}", flush=True)
