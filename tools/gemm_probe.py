import torch, time
dev="cuda:0"; M=1024; P=1
torch.manual_seed(0)
L=torch.tril(torch.randn(P,M,M,dtype=torch.float64,device=dev))
K6=torch.randn(M,M,dtype=torch.float64,device=dev); K6=K6@K6.T
batch=torch.empty((P+1,M,M),dtype=torch.float64,device=dev)
def timeit(fn,reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
X=K6@L
print("K6 @ L                         %.1f us" % timeit(lambda: K6@L))
print("bmm(L^T, X, out=batch[:P])     %.1f us" % timeit(lambda: torch.bmm(L.transpose(-1,-2), X, out=batch[:P])))
print("bmm(L^T, X)                    %.1f us" % timeit(lambda: torch.bmm(L.transpose(-1,-2), X)))
Lt=L.transpose(-1,-2).contiguous()
print("transpose copy                 %.1f us" % timeit(lambda: L.transpose(-1,-2).contiguous()))
print("bmm(Lt, X, out)                %.1f us" % timeit(lambda: torch.bmm(Lt, X, out=batch[:P])))
print("mm(L[0].t(), X[0])             %.1f us" % timeit(lambda: torch.mm(L[0].t(), X[0])))
print("mm(X[0].t(), L[0]) (=W^T)      %.1f us" % timeit(lambda: torch.mm(X[0].t(), L[0])))
print("(L^T K6) @ L: mm(L.t(),K6)     %.1f us" % timeit(lambda: torch.mm(L[0].t(), K6)))
A=torch.randn(M,M,dtype=torch.float64,device=dev)
print("mm(A, A)                       %.1f us" % timeit(lambda: torch.mm(A, A)))
print("mm(A.t(), A)                   %.1f us" % timeit(lambda: torch.mm(A.t(), A)))
print("mm(A, A.t())                   %.1f us" % timeit(lambda: torch.mm(A, A.t())))
print("mm(A.t(), A.t())               %.1f us" % timeit(lambda: torch.mm(A.t(), A.t())))
v=torch.randn(M,1,dtype=torch.float64,device=dev)
print("K6 @ v                         %.1f us" % timeit(lambda: K6@v))
print("clone 8MB                      %.1f us" % timeit(lambda: K6.clone()))
print("diagonal add                   %.1f us" % timeit(lambda: batch[:P].diagonal(dim1=-2,dim2=-1).add_(1.0)))
