"""Experiment (historic: the runtime switch TSVGP_FILL_GRID it drove is now the compile-time macro TSVGP_FILL_GRID_CAP; build one
library per cap with -DTSVGP_FILL_GRID_CAP=<n> and point TSVGP_HIP_LIB at it): the K(X,Z) fill as FEW looping workgroups on a side stream, overlapped with the real
M x M prelude of the E-step (t_SVGP._site_operands: K_uu fill, GEMMs, batched Cholesky + inverse) on the main stream.
A saturating grid starves whatever the other queue holds (tools/exp_overlap.py); a grid that is resident at once leaves
wave slots and LDS on every CU.   usage: python tools/exp_overlap2.py [grid caps ...]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = importlib.import_module("t-svgp_amd")
dev = torch.device("cuda:0")
N, M, D = 1_000_000, 1024, 8
rng = np.random.RandomState(0)
X = rng.randn(N, D); Y = np.sin(X @ rng.randn(D, 1)) + np.sqrt(0.1) * rng.randn(N, 1); Z = X[:M].copy()
model = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z, num_data=N, device=dev)
Xd, Yd = torch.as_tensor(X).to(dev), torch.as_tensor(Y).to(dev)
for _ in range(2):
    model.natgrad_step((Xd, Yd), lr=0.8)
eng = model._get_engine()
kern = model.kernel
inv_ls = kern.inv_lengthscales(D, torch.float64, dev)
Kfu = eng._get("Kfu", ((N + 127) // 128 * 128, M), torch.float64)
Zd = model._Z()
routes = model._routes(1e-9)

def prelude():
    return model._site_operands(whiten_jitter=1e-9, routes=routes)

def fill():
    eng.se_fill(Xd, Zd, inv_ls, 1.0, Kfu, kern.kind)

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

main = torch.cuda.current_stream(dev)
prio = int(os.environ.get("OVL_PRIO", "0"))
side = torch.cuda.Stream(dev, priority=prio)
print("stream priority range (torch):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None,
      "side priority", side.priority)
print(f"routes {routes}")
print(f"prelude alone          {timeit(prelude):.3f} ms")
os.environ.pop("TSVGP_FILL_GRID", None)
print(f"fill alone, full grid  {timeit(fill):.3f} ms")
print(f"sequential, full grid  {timeit(lambda: (prelude(), fill())):.3f} ms")
for cap in [int(a) for a in sys.argv[1:]] or [256]:
    if cap > 0:
        os.environ["TSVGP_FILL_GRID"] = str(cap)
    else:
        os.environ.pop("TSVGP_FILL_GRID", None)
    def both():
        e0 = torch.cuda.Event(); e0.record(main); side.wait_event(e0)
        if os.environ.get("OVL_SWAP"):  # the prelude on the (high-priority) side stream, the fill on the main one
            with torch.cuda.stream(side):
                prelude()
                e1 = torch.cuda.Event(); e1.record(side)
            fill()
        else:
            with torch.cuda.stream(side):
                fill()
                e1 = torch.cuda.Event(); e1.record(side)
            prelude()
        main.wait_event(e1)
    print(f"cap {cap:5d} (x2 column tiles): fill alone {timeit(fill):.3f} ms   overlapped with the prelude {timeit(both):.3f} ms", flush=True)
