"""Do the library GEMMs of the forward get faster under PyTorch TunableOp (per-shape pick among rocBLAS / hipBLASLt solutions)?
Shapes of cfg3's projections; F.linear(x [M,K], w [N,K]).  Also wgnn_linear_fwd on the same shapes."""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from scdeepsort_amd import ops
dev = 'cuda:0'
shapes = [(100_000, 256, 400), (20_000, 256, 400), (100_000, 256, 256), (20_000, 256, 256), (100_000, 16, 256), (12_500, 256, 400)]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
res = {}
for lib in ("default", "hipblaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print("cannot select", lib, e); continue
    for (M, N, K) in shapes:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
        t = timeit(lambda: F.linear(x, w))
        res[(lib, M, N, K)] = t
        print(f"{lib:10s} {M:7d}x{N:4d}x{K:4d}: {t*1e3:7.1f} us  {2*M*N*K/t/1e9:6.1f} TF", flush=True)
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    t = timeit(lambda: ops.linear_fwd(x, w))
    print(f"{'wgnn':10s} {M:7d}x{N:4d}x{K:4d}: {t*1e3:7.1f} us  {2*M*N*K/t/1e9:6.1f} TF", flush=True)
import torch.cuda.tunable as T
T.enable(True); T.tuning_enable(True)
T.set_max_tuning_duration(3000); T.set_max_tuning_iterations(50)
T.set_filename('/root/repo/gpurun_out/tunableop_results.csv')
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    t0 = time.time(); F.linear(x, w); torch.cuda.synchronize(); tt = time.time() - t0
    t = timeit(lambda: F.linear(x, w))
    print(f"{'tunable':10s} {M:7d}x{N:4d}x{K:4d}: {t*1e3:7.1f} us  {2*M*N*K/t/1e9:6.1f} TF   (tuning took {tt:.1f} s)", flush=True)
T.write_file()
print(open('/root/repo/gpurun_out/tunableop_results.csv').read()[:3000])
