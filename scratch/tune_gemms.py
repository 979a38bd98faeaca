"""Round 4: TunableOp picks for every library GEMM of the BASELINE workloads -> scdeepsort_amd/
tuned_gemms_gfx950.csv (tracked).  Workloads: cfg3 forward (N = 1 and rank 0's shard at N = 2 / 4 / 8), cfg2 forward, the
cfg3 full-batch training step, cfg3 with the reference's default hidden_dim = 200.  Then times the cfg3 forward with the
libraries' own heuristics vs the tuned picks in one process."""
import sys, json, time, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, dist as D, tuning, ops
from scdeepsort_amd.sharded import ShardedWgnn
dev = torch.device('cuda:0')
try:
    torch.backends.cuda.preferred_blas_library("hipblaslt")
except Exception:
    pass
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / n, 4)
cfg = S.CONFIGS['cfg3']; G = cfg.genes
rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, device=dev)
feats_g = S.synth_features(G, cfg.dense_dim, seed=7, device=dev); feats_c = S.synth_features(cfg.cells, cfg.dense_dim, seed=100, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
torch.manual_seed(1234)
model = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, 2, G, activation=F.relu).to(dev).eval()
def fwd():
    with torch.no_grad():
        return model(g, (feats_g, feats_c))
before = timeit(fwd)
gdeg, gsum = ShardedWgnn.gene_stats(col, val, G)
engines = {}
for N in (2, 4, 8):
    lo, hi = D.shard_range(cfg.cells, 0, N); b, e = int(rp[lo]), int(rp[hi])
    engines[N] = (ShardedWgnn.build(model, (rp[lo:hi + 1] - rp[lo]).clone(), col[b:e].clone(), val[b:e].clone(), G, global_stats=(gdeg, gsum)),
                  feats_c[lo:hi].contiguous())
shard_before = {N: timeit(lambda: eng.forward(feats_g, fc, gather_logits=False) if not torch.is_grad_enabled() else None, 10)
                for N, (eng, fc) in engines.items()} if False else {}
def shard_fwd(N):
    eng, fc = engines[N]
    with torch.no_grad():
        return eng.forward(feats_g, fc, gather_logits=False)
for N in engines: shard_before[N] = timeit(lambda: shard_fwd(N), 10)
mt = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, 2, G, activation=F.relu, dropout=0.1).to(dev)
opt = torch.optim.Adam(mt.parameters(), lr=1e-3, weight_decay=5e-4, fused=True)
y = torch.arange(cfg.cells, device=dev) % cfg.n_classes
def train():
    loss = sda.cross_entropy_sum(mt(g, (feats_g, feats_c)), y); opt.zero_grad(); loss.backward(); opt.step()
train_before = timeit(train, 5)
m200 = sda.GNN(cfg.dense_dim, 200, cfg.n_classes, 2, G, activation=F.relu).to(dev).eval()
def fwd200():
    with torch.no_grad():
        return m200(g, (feats_g, feats_c))
c2 = S.CONFIGS['cfg2']
rp2, col2, val2 = S.synth_expression(c2.cells, c2.genes, c2.density, device=dev)
g2 = sda.CellGeneGraph.from_device_csr(rp2, col2, val2, c2.genes)
m2 = sda.GNN(c2.dense_dim, c2.hidden, c2.n_classes, 2, c2.genes, activation=F.relu).to(dev).eval()
f2 = S.synth_features(c2.genes + c2.cells, c2.dense_dim, device=dev)
def fwd2():
    with torch.no_grad():
        return m2(g2, f2)
def workload():
    for _ in range(2):
        fwd(); fwd200(); fwd2(); train()
        for N in engines: shard_fwd(N)
t0 = time.time()
path = tuning.tune_gemms(workload)
print(f"tuned in {time.time() - t0:.1f} s -> {path}", flush=True)
print(open(path).read())
after = timeit(fwd); shard_after = {N: timeit(lambda: shard_fwd(N), 10) for N in engines}; train_after = timeit(train, 5)
import shutil; shutil.copy(path, '/root/repo/gpurun_out/tuned_gemms_gfx950.csv')
print(json.dumps({"cfg3_forward_ms": {"library_heuristics": before, "tuned_picks": after},
                  "shard_forward_ms": {str(N): {"library_heuristics": shard_before[N], "tuned_picks": shard_after[N]} for N in engines},
                  "cfg3_train_step_ms": {"library_heuristics": train_before, "tuned_picks": train_after}}))
