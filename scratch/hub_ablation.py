"""Round 4: price a dense treatment of the hub sources inside agg_tiled_flat4.  cfg3 graph with
UNSHUFFLED gene ids (gene id = popularity rank, so the first LDS blocks of the cells<-genes pass hold the hub genes);
timing-only ablation: the entry pipeline of the first 2n LDS blocks (156 n source rows) is skipped (wrong results) ->
upper bound of what taking those edges out of the sparse pipeline can save, BEFORE the cost of computing them densely."""
import sys, json, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C, H = cfg.genes, cfg.cells, 256
rp, col, val = S.synth_expression(C, G, device=dev, shuffle_genes=False)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
cnt = torch.bincount(col.long(), minlength=G).double()
share = {n: round(float(cnt[:156 * n].sum() / cnt.sum()), 4) for n in (1, 2, 3, 4, 6)}
dens = {n: round(float(cnt[:156 * n].sum() / (156 * n * C)), 4) for n in (1, 2, 3, 4, 6)}
print("edge share of the first 156 n genes:", share, "density:", dens, flush=True)
alpha = torch.rand(G + 2, device=dev) + 0.5
hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
tp = g.cg.tile_plan(78)
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
res = {}
for rep in range(3):
    for n in (0, 1, 2, 3):
        ops.DEBUG_FLAGS = (n << 21) | ((1 << 24) if n == 0 else 0)      # bit 24: nothing reads it - same (DBG) instantiation for n = 0
        t = timeit(lambda: ops.agg_fwd_tiled(g.cg, tp, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
        res.setdefault(n, []).append(round(t, 4))
        print(f"rep {rep} skip first {2 * n} blocks: {t:.4f} ms", flush=True)
ops.DEBUG_FLAGS = 0
print(json.dumps({"edge_share": share, "density": dens, "ms": res}))
