"""wgnn_linear_fwd: 128-row vs 64-row tiles vs the kernel's own choice vs hipBLASLt, on the forward's shapes."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from scdeepsort_amd import ops
dev = 'cuda:0'
try: torch.backends.cuda.preferred_blas_library("hipblaslt")
except Exception: pass
shapes = [(100_000, 256, 400), (20_000, 256, 400), (100_000, 256, 256), (20_000, 256, 256), (100_000, 16, 256), (12_500, 256, 400), (50_000, 256, 400), (764_741, 256, 400)]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    ref = F.linear(x, w)
    row = [f"hipblaslt {timeit(lambda: F.linear(x, w))*1e3:7.1f}"]
    for tr in (128, 64, None):
        out = ops.linear_fwd(x, w, tile_rows=tr)
        err = (out - ref).abs().max().item()
        t = timeit(lambda: ops.linear_fwd(x, w, tile_rows=tr))
        row.append(f"wgnn[{tr}] {t*1e3:7.1f} us {2*M*N*K/t/1e9:5.1f} TF err {err:.1e}")
    print(f"{M:7d}x{N:4d}x{K:4d}: " + " | ".join(row), flush=True)
