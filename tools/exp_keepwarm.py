#!/usr/bin/env python3
"""Does keeping the matrix pipes busy during the low-activity M x M section spare the moments kernel its slow start?
tsvgp_moments_f64 at the headline sizes after (a) nothing, (b) 3 ms of a one-workgroup spin, (c) the same spin with an fp64
MFMA burner (G workgroups x 1 wave per SIMD, no memory traffic) beside it on a second stream, (d) the K(X,Z) fill + 300 us spin,
(e) the same with the burner beside the fill.  The burner ends before the moments kernel starts (event).  GPU box.
usage: exp_keepwarm.py [G ...]"""
import ctypes, importlib, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
SRC = r'''
#include <hip/hip_runtime.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void burn(double* out, int iters) {
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[threadIdx.x] = s;
}
extern "C" int launch_burn(double* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}
'''
d = tempfile.mkdtemp(); open(d + "/burn.hip", "w").write(SRC)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", d + "/burn.hip", "-o", d + "/libburn.so"])
burnlib = ctypes.CDLL(d + "/libburn.so")
p = importlib.import_module("t-svgp_amd")
E = importlib.import_module("t-svgp_amd.estep"); K_ = importlib.import_module("t-svgp_amd.kernels")
dev = torch.device("cuda:0")
N, M, D = 1_000_000, 1024, 8
eng = E.EStepEngine(torch.float64, dev)
g = torch.Generator().manual_seed(0)
X = torch.randn(N, D, generator=g, dtype=torch.float64).to(dev); Z = X[:M].clone()
kern = K_.SquaredExponential(variance=1.0, lengthscales=1.0)
inv_ls = kern.inv_lengthscales(D, torch.float64, dev)
Np = (N + 127) // 128 * 128
Kfu = torch.empty(Np, M, dtype=torch.float64, device=dev)
eng.se_fill(X, Z, inv_ls, 1.0, Kfu)
T = torch.triu(torch.randn(M, M, dtype=torch.float64, device=dev)) / 32
gam = torch.randn(M, 1, dtype=torch.float64, device=dev); Y = torch.randn(N, 1, dtype=torch.float64, device=dev)
g0 = torch.empty(Np, 1, dtype=torch.float64, device=dev); g1 = torch.empty_like(g0)
vep = torch.empty(Np // 128, dtype=torch.float64, device=dev); npp = torch.empty(Np // 128, dtype=torch.int32, device=dev)
fn = eng._fn("tsvgp_moments")
sink = torch.zeros(256, dtype=torch.float64, device=dev)
side = torch.cuda.Stream(dev)
def moments():
    assert fn(Kfu.data_ptr(), T.data_ptr(), gam.data_ptr(), Y.data_ptr(), 1e9, 1, 0.1, None, None, g0.data_ptr(), g1.data_ptr(),
              vep.data_ptr(), npp.data_ptr(), N, Np, M, 1, 1, eng._stream()) == 0
def burn_ms(G, iters):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        e0.record(); assert burnlib.launch_burn(ctypes.c_void_p(sink.data_ptr()), G, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0; e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1)
def with_burner(G, iters, work):
    """`work` on the main stream with the burner beside it on the side stream; the main stream then waits for the burner."""
    side.wait_stream(torch.cuda.current_stream())
    assert burnlib.launch_burn(ctypes.c_void_p(sink.data_ptr()), G, iters, ctypes.c_void_p(side.cuda_stream)) == 0
    work()
    torch.cuda.current_stream().wait_stream(side)
def timed(pre, reps=8):
    for _ in range(2): pre(); moments()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        pre()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); moments(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps
cyc = 2100
spin = lambda us: torch.cuda._sleep(us * cyc)
print(f"back to back:                                   {timed(lambda: moments()):.3f} ms")
print(f"after 3000 us of a one-workgroup spin:          {timed(lambda: (moments(), spin(3000))):.3f} ms")
print(f"after the fill + 300 us spin:                   {timed(lambda: (moments(), eng.se_fill(X, Z, inv_ls, 1.0, Kfu), spin(300))):.3f} ms")
for G in [int(a) for a in sys.argv[1:]] or [64, 256, 1024]:
    it1 = 2000; ms1 = burn_ms(G, it1)
    for target in (2.0, 2.9):
        iters = max(int(it1 * target / ms1), 1)
        print(f"G={G:5d}: burner alone {burn_ms(G, iters):.2f} ms; moments after 3000 us spin + burner: "
              f"{timed(lambda: (moments(), with_burner(G, iters, lambda: spin(3000)))):.3f} ms;  after fill + 300 us spin + burner: "
              f"{timed(lambda: (moments(), with_burner(G, iters, lambda: (eng.se_fill(X, Z, inv_ls, 1.0, Kfu), spin(300))))):.3f} ms")
