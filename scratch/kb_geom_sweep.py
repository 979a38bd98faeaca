"""cfg3: LDS block height x geometry re-check for both passes with the round-2 plan (one round on the gene side)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C, H = cfg.genes, cfg.cells, cfg.hidden
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
alpha = torch.rand(G + 2, device=dev) + 0.5
hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for rep in range(2):
    for kb in (64, 72, 78):
        tpc = GR.build_tile_plan(g.cg, None, None, block_rows=kb); tpg = GR.build_tile_plan(g.gc, None, None, block_rows=kb)
        tc = timeit(lambda: ops.agg_fwd_tiled(g.cg, tpc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
        tg = timeit(lambda: ops.agg_fwd_tiled(g.gc, tpg, alpha, sda.DST_IS_GENE, G, hc, hg))
        print(f"kb={kb} cells {tpc.n_row_tiles}x{tpc.n_col_splits} {tc:.3f} ms   genes {tpg.n_row_tiles}x{tpg.n_col_splits} {tg:.3f} ms", flush=True)
for geom in ((85, 3), (64, 4), (51, 5), (42, 6)):
    try:
        tpg = GR.build_tile_plan(g.gc, geom[0], geom[1], block_rows=78)
    except Exception as e:
        print(geom, "n/a", e); continue
    tg = timeit(lambda: ops.agg_fwd_tiled(g.gc, tpg, alpha, sda.DST_IS_GENE, G, hc, hg))
    print(f"genes {tpg.n_row_tiles}x{tpg.n_col_splits} partial_MB={tpg.n_partials*H*4/1e6:.0f} {tg:.3f} ms", flush=True)
