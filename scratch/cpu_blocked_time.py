"""Where the cache-blocked CPU baseline spends its time on the GPU box's host (no GPU needed for the timing itself)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import c_oracle as CO
import scipy.sparse as sp
from scdeepsort_amd import synthetic as S
C, G, D = 100_000, 20_000, 256
rp, col, val = S.synth_expression(C, G, 0.04, device='cpu')
X = sp.csr_matrix((val.numpy(), col.numpy(), rp.numpy()), shape=(C, G)); X.sort_indices()
XT = sp.csr_matrix(X.T); XT.sort_indices()
rng = np.random.default_rng(0)
alpha = (rng.random(G + 2) + 0.5).astype(np.float32)
hg = rng.standard_normal((G, D)).astype(np.float32); hc = rng.standard_normal((C, D)).astype(np.float32)
print("threads", CO.num_threads(), "cpus", os.cpu_count(), flush=True)
def best(f, n=3):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts)
for name, A, mode, si, hs, hself in (("cells<-genes", X, 0, G + 1, hg, hc), ("genes<-cells", XT, 1, G, hc, hg)):
    t = best(lambda: CO.aggregate(A.indptr, A.indices, A.data, alpha, mode, si, hs, hself), 2)
    print(f"{name}: row-wise {t*1e3:.1f} ms = {2*A.nnz*D/t/1e9:.0f} GF/s", flush=True)
    for tr, br in ((256, 256), (128, 256), (64, 256), (64, 512), (32, 512), (128, 1024), (16, 2048)):
        t = best(lambda: CO.aggregate_blocked(A.indptr, A.indices, A.data, alpha, mode, si, hs, hself, tr, br))
        print(f"   blocked tile {tr:4d} x block {br:5d}: {t*1e3:8.1f} ms = {2*A.nnz*D/t/1e9:7.0f} GF/s", flush=True)
x = torch.from_numpy(rng.standard_normal((C, 400)).astype(np.float32)); w = torch.from_numpy(rng.standard_normal((256, 400)).astype(np.float32))
t = best(lambda: torch.nn.functional.linear(x, w), 5)
print(f"F.linear 100k x 400 x 256 on the host: {t*1e3:.1f} ms = {2*C*400*256/t/1e9:.0f} GF/s")
