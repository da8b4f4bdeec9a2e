import torch, time
dev="cuda:0"
A=torch.randn(1024,1024,dtype=torch.float64,device=dev); Bm=torch.randn(1024,1024,dtype=torch.float64,device=dev)
def timeit(fn,n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for lib in ("default","hipblaslt","hipblas"):
    try:
        if lib!="default": torch.backends.cuda.preferred_blas_library(lib)
        print(lib, torch.backends.cuda.preferred_blas_library())
        print("  mm NN   %.1f us"%timeit(lambda: torch.mm(A,Bm)))
        print("  mm NT   %.1f us"%timeit(lambda: torch.mm(A,Bm.T)))
        print("  mm TN   %.1f us"%timeit(lambda: torch.mm(A.T,Bm)))
        Cb=torch.empty(1,1024,1024,dtype=torch.float64,device=dev); I=torch.eye(1024,dtype=torch.float64,device=dev)
        print("  baddbmm %.1f us"%timeit(lambda: torch.baddbmm(I.expand(1,1024,1024), A.T[None], Bm[None], out=Cb)))
        A5=torch.randn(512,512,dtype=torch.float64,device=dev)
        print("  mm 512  %.1f us"%timeit(lambda: torch.mm(A5,A5)))
    except Exception as ex:
        print(lib,"failed:",str(ex)[:200])
