"""Experiment: K_fu fill on a CU-masked side stream, overlapped with the latency-bound M x M factorisations.
usage: python tools/exp_overlap.py [reserved_cus ...]"""
import ctypes, importlib, os, sys, time
import torch
sys.path.insert(0, "/root/repo")
p = importlib.import_module("t-svgp_amd")
from importlib import import_module
E = import_module("t-svgp_amd.estep"); K_ = import_module("t-svgp_amd.kernels")
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
N, M, D = 1_000_000, 1024, 8
eng = E.EStepEngine(torch.float64, dev)
g = torch.Generator().manual_seed(0)
X = torch.rand(N, D, generator=g, dtype=torch.float64).to(dev); Z = X[:M].clone()
kern = K_.SquaredExponential(variance=1.0, lengthscales=1.0)
inv_ls = kern.inv_lengthscales(D, torch.float64, dev)
Kfu = torch.empty((N + 127) // 128 * 128, M, dtype=torch.float64, device=dev)
A = eng.kuu(Z, kern) + 1e-6 * torch.eye(M, dtype=torch.float64, device=dev)
A2 = torch.stack([A, A + torch.eye(M, dtype=torch.float64, device=dev)])

def masked_stream(reserve):
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(ncu - reserve):
        mask[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)

MODE = os.environ.get("OVL_MODE", "potrf")
def prelude():
    if MODE == "gemm":
        B_ = A
        for _ in range(12): B_ = A @ B_ * 1e-3
    else:
        eng.cholesky(A2); eng.cholesky(A)

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

main = torch.cuda.current_stream(dev)  # the (null) stream everything else runs on
print("CUs", torch.cuda.get_device_properties(dev).multi_processor_count)
print(f"fill alone (main)      {timeit(lambda: eng.se_fill(X, Z, inv_ls, 1.0, Kfu)):.3f} ms")
print(f"2 x potrf alone        {timeit(prelude):.3f} ms")
print(f"sequential             {timeit(lambda: (eng.se_fill(X, Z, inv_ls, 1.0, Kfu), prelude())):.3f} ms")
for reserve in [int(a) for a in sys.argv[1:]] or [0]:
    side = masked_stream(reserve) if reserve else torch.cuda.Stream(dev)
    def fill_side():
        e0 = torch.cuda.Event(); e0.record(main); side.wait_event(e0)
        with torch.cuda.stream(side):
            eng.se_fill(X, Z, inv_ls, 1.0, Kfu)
            e1 = torch.cuda.Event(); e1.record(side)
        return e1
    def both():
        e1 = fill_side(); prelude(); main.wait_event(e1)
    print(f"reserve {reserve:3d}: fill alone on side {timeit(lambda: main.wait_event(fill_side())):.3f} ms   overlapped with 2 x potrf {timeit(both):.3f} ms")
