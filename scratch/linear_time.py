"""wgnn_linear_fwd (hand-written fp32 MFMA GEMM) vs torch F.linear (default BLAS and hipBLASLt) on the forward's shapes."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from scdeepsort_amd import ops
dev = 'cuda:0'
shapes = [(100000, 400, 256), (20000, 400, 256), (20000, 256, 256), (100000, 256, 256), (100000, 256, 16), (100000, 400, 200)]
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
xs = [(torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)) for m, k, n in shapes]
for lib in ("default", "hipblaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "n/a", e); continue
    ts = [timeit(lambda: F.linear(x, w)) for x, w in xs]
    print(f"{lib:10s}", [round(t, 1) for t in ts], "us; TF/s", [round(2 * m * k * n / t / 1e6, 1) for (m, k, n), t in zip(shapes, ts)])
ts = [timeit(lambda: ops.linear_fwd(x, w)) for x, w in xs]
print(f"{'wgnn_mfma':10s}", [round(t, 1) for t in ts], "us; TF/s", [round(2 * m * k * n / t / 1e6, 1) for (m, k, n), t in zip(shapes, ts)])
for (x, w) in xs[:1]:
    print("max |wgnn - torch|", float((ops.linear_fwd(x, w) - F.linear(x, w)).abs().max()))
