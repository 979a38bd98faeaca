"""Gene side (genes<-cells, 85 x 3 tiles) with ONE dedicated loader wave: needs <= 240 rows per tile, i.e. fewer virtual rows."""
import sys, json, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C, H = cfg.genes, cfg.cells, 256
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
alpha = torch.rand(G + 2, device=dev) + 0.5
hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
plans = {}
for share, L, geom in ((0.5, 0, (None, None)), (0.8, 0, (None, None)), (0.8, 1, (85, 3)), (1.2, 1, (85, 3)), (0.8, 2, (92, 3)), (0.8, 2, (128, 2))):
    GR.VIRTUAL_ROW_SHARE = share
    tp = GR.build_tile_plan(g.gc, geom[0], geom[1], block_rows=78, n_loaders=L)
    plans[(share, L, geom)] = tp
    print((share, L, geom), "tiles", tp.n_row_tiles, "x", tp.n_col_splits, "loaders", tp.n_loaders, "partial rows", tp.n_partials, flush=True)
ref = ops.agg_fwd_tiled(g.gc, plans[(0.5, 0, (None, None))], alpha, sda.DST_IS_GENE, G, hc, hg)
for rep in range(3):
    for k, tp in plans.items():
        out = ops.agg_fwd_tiled(g.gc, tp, alpha, sda.DST_IS_GENE, G, hc, hg)
        t = timeit(lambda: ops.agg_fwd_tiled(g.gc, tp, alpha, sda.DST_IS_GENE, G, hc, hg))
        print(rep, k, f"{t:.4f} ms  max|diff| {(out - ref).abs().max().item():.1e}", flush=True)
