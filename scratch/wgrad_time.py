import sys, torch
sys.path.insert(0, '/root/repo')
from scdeepsort_amd import ops
dev='cuda:0'
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for M, N, K in ((100000, 256, 400), (20000, 256, 400), (100000, 256, 256), (20000, 256, 256), (100000, 16, 256)):
    g = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    t1 = timeit(lambda: g.t() @ x); t2 = timeit(lambda: ops.linear_wgrad(g, x))
    print(f"dW {N}x{K} over M={M}: torch {t1:.1f} us ({2*M*N*K/t1/1e6:.1f} TF)   wgnn {t2:.1f} us ({2*M*N*K/t2/1e6:.1f} TF)")
