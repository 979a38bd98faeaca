"""Same-run A/B of the number of dedicated loader waves (0-3) on the cfg3 operands (round 5: grouped loader loop)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C = cfg.genes, cfg.cells; H = 256
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
alpha = torch.rand(G + 2, device=dev) + 0.5
hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
kb = ops.tiled_block_rows(H)
def timeit(f, n=20):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
plans = {L: (GR.build_tile_plan(g.cg, None, None, block_rows=kb, n_loaders=L), GR.build_tile_plan(g.gc, None, None, block_rows=kb, n_loaders=L)) for L in (0, 1, 2, 3)}
for L, (pc, pg) in plans.items():
    print(L, 'cells', pc.n_row_tiles, 'x', pc.n_col_splits, 'loaders', pc.n_loaders, '| genes', pg.n_row_tiles, 'x', pg.n_col_splits, 'loaders', pg.n_loaders)
for rep in range(3):
    for L, (pc, pg) in plans.items():
        tc = timeit(lambda: ops.agg_fwd_tiled(g.cg, pc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
        tg = timeit(lambda: ops.agg_fwd_tiled(g.gc, pg, alpha, sda.DST_IS_GENE, G, hc, hg))
        print(f'rep {rep} loaders {L}: cells<-genes {tc:.3f} ms, genes<-cells {tg:.3f} ms', flush=True)
