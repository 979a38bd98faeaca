import os, sys, time, torch, torch.nn.functional as F
dev='cuda:0'
shapes=[(100000,400,256),(20000,400,256),(20000,256,256),(100000,256,256),(100000,256,16)]
def timeit(f,n=30):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
xs=[(torch.randn(m,k,device=dev),torch.randn(n,k,device=dev),torch.randn(n,device=dev)) for m,k,n in shapes]
base=[timeit(lambda: F.linear(x,w)) for x,w,b in xs]
print('default      ', [round(t,1) for t in base], 'sum', round(sum(base),1))
try:
    torch.backends.cuda.preferred_blas_library('hipblaslt')
    t2=[timeit(lambda: F.linear(x,w)) for x,w,b in xs]
    print('hipblaslt    ', [round(t,1) for t in t2], 'sum', round(sum(t2),1))
except Exception as e: print('hipblaslt n/a', e)
try:
    torch.backends.cuda.preferred_blas_library('default')
    import torch.cuda.tunable as tn
    tn.enable(True); tn.set_max_tuning_duration(200); tn.set_max_tuning_iterations(50)
    t0=time.time()
    t3=[timeit(lambda: F.linear(x,w)) for x,w,b in xs]
    print('tunable      ', [round(t,1) for t in t3], 'sum', round(sum(t3),1), 'tuning took', round(time.time()-t0,1),'s')
except Exception as e: print('tunable n/a', e)
# round 2: other formulations of the big projection (M=100k, K=400, N=256)
try:
    import torch.cuda.tunable as tn
    tn.enable(False)
except Exception:
    pass
x, w, b = xs[0]
wt = w.t().contiguous()
print('x @ Wt (contig)', round(timeit(lambda: x @ wt), 1))
xp = F.pad(x, (0, 16)); wp = F.pad(w, (0, 16))
print('K padded to 416', round(timeit(lambda: F.linear(xp, wp)), 1))
xp = F.pad(x, (0, 48)); wp = F.pad(w, (0, 48))
print('K padded to 448', round(timeit(lambda: F.linear(xp, wp)), 1))
xa = torch.randn(120000, 400, device=dev)
print('genes+cells in one GEMM (120k x 400 x 256)', round(timeit(lambda: F.linear(xa, w)), 1))
out = torch.empty(100000, 256, device=dev)
print('addmm out=', round(timeit(lambda: torch.mm(x, wt, out=out)), 1))
x16 = x.half(); w16 = w.half()
print('fp16 inputs (reference only, NOT the product dtype)', round(timeit(lambda: F.linear(x16, w16)), 1))
