"""Round 4: tile geometry at cfg2 size (10k cells x 5k genes, 2.0 M edges, D = 128) - one pass of the tile kernel (incl. scale_rows /
agg_finalize) for a sweep of (row tiles, column splits), loader waves on / off."""
import sys, json, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
cfg = S.CONFIGS['cfg2']; G, C, D = cfg.genes, cfg.cells, 128
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
alpha = torch.rand(G + 2, device=dev) + 0.5
hg = S.synth_features(G, D, device=dev); hc = S.synth_features(C, D, seed=3, device=dev)
def timeit(f, n=50):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / n * 1e3, 1)
kb = ops.tiled_block_rows(D)
res = {}
for name, csr, mode, sidx, src, slf in (("cells", g.cg, sda.SRC_IS_GENE, G + 1, hg, hc), ("genes", g.gc, sda.DST_IS_GENE, G, hc, hg)):
    R = csr.n_rows
    for L in (0, 1):
        for rt, cs in ((None, None), (40, 6), (42, 3), (50, 5), (64, 4), (84, 3), (128, 2), (250, 1), (20, 12), (32, 8)):
            try:
                tp = GR.build_tile_plan(csr, rt, cs, block_rows=kb, n_loaders=L)
            except Exception as e:
                continue
            t = timeit(lambda: ops.agg_fwd_tiled(csr, tp, alpha, mode, sidx, src, slf))
            res[f"{name} L{L} req {rt}x{cs} -> {tp.n_row_tiles}x{tp.n_col_splits} loaders {tp.n_loaders} partial rows {tp.n_partials}"] = t
for k, v in sorted(res.items(), key=lambda kv: (kv[0][:5], kv[1])):
    print(v, k)
