"""Round 4: what would the SPARSE remainder of the cells<-genes pass cost if the H most popular genes were handled elsewhere
(densely)?  cfg3 graph (shuffled gene ids, the bench's graph); the edges of the top-H genes are dropped from the CSR (all
20 000 source rows still stream through LDS), the tile pass is timed.  With the ablation of scratch/hub_ablation.py (sorted ids)
this brackets the gain of a dense hub treatment before its own cost."""
import sys, json, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C, H = cfg.genes, cfg.cells, 256
rp, col, val = S.synth_expression(C, G, device=dev)
cnt = torch.bincount(col.long(), minlength=G)
order = torch.argsort(cnt, descending=True)
alpha = torch.rand(G + 2, device=dev) + 0.5
hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / n, 4)
rows = torch.repeat_interleave(torch.arange(C, device=dev), (rp[1:] - rp[:-1]))
res = {}
for nh in (0, 156, 312, 468, 780):
    hub = torch.zeros(G, dtype=torch.bool, device=dev); hub[order[:nh]] = True
    keep = ~hub[col.long()]
    c2, v2, r2 = col[keep], val[keep], rows[keep]
    rp2 = torch.zeros(C + 1, dtype=torch.int64, device=dev); rp2[1:] = torch.cumsum(torch.bincount(r2, minlength=C), 0)
    g = sda.CellGeneGraph.from_device_csr(rp2, c2, v2, G)
    tp = g.cg.tile_plan(78)
    ts = [timeit(lambda: ops.agg_fwd_tiled(g.cg, tp, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)) for _ in range(3)]
    res[nh] = {"edges_left": int(c2.shape[0]), "share_removed": round(1 - c2.shape[0] / col.shape[0], 4), "ms": ts,
               "tiles": f"{tp.n_row_tiles}x{tp.n_col_splits} L{tp.n_loaders}"}
    print(nh, res[nh], flush=True)
    del g, tp
print(json.dumps(res))
