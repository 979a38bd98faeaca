"""Same-run A/B of the two tile geometries on the cfg3 operands: flat (16 waves x 16 rows, loader wave) vs tall (8 x 50)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C = cfg.genes, cfg.cells; H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
alpha = torch.rand(G + 2, device=dev) + 0.5
hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
kb = ops.tiled_block_rows(H)
def timeit(f, n=20):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
tm = ops.TILED_MIN_WORK; ops.TILED_MIN_WORK = None
ref_c = ops.agg_fwd(g.cg, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
ref_g = ops.agg_fwd(g.gc, alpha, sda.DST_IS_GENE, G, hc, hg)
ops.TILED_MIN_WORK = tm
plans = {}
t0 = time.perf_counter()
plans['flat'] = (GR.build_tile_plan(g.cg, None, None, block_rows=kb, n_loaders=1), GR.build_tile_plan(g.gc, None, None, block_rows=kb, n_loaders=1))
torch.cuda.synchronize(); t1 = time.perf_counter()
plans['tall'] = (GR.build_tile_plan(g.cg, None, None, block_rows=kb, geom=GR.GEOM_TALL), GR.build_tile_plan(g.gc, None, None, block_rows=kb, geom=GR.GEOM_TALL))
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'plan build: flat {t1 - t0:.2f} s, tall {t2 - t1:.2f} s')
for k, (pc, pg) in plans.items():
    print(k, 'cells side', pc.n_row_tiles, 'x', pc.n_col_splits, 'loaders', pc.n_loaders, 'entries', pc.entries.shape[0],
          '| gene side', pg.n_row_tiles, 'x', pg.n_col_splits, 'loaders', pg.n_loaders, 'partials', pg.n_partials)
for rep in range(3):
    for k, (pc, pg) in plans.items():
        out = []
        for nm, fl in [('full', 0), ('nofill', 1 << 16), ('nofill+nobar', (1 << 16) | (1 << 18))]:
            ops.DEBUG_FLAGS = fl
            tc = timeit(lambda: ops.agg_fwd_tiled(g.cg, pc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
            tg = timeit(lambda: ops.agg_fwd_tiled(g.gc, pg, alpha, sda.DST_IS_GENE, G, hc, hg))
            out.append(f'{nm} {tc:.3f}/{tg:.3f}')
        ops.DEBUG_FLAGS = 0
        ec = float((ops.agg_fwd_tiled(g.cg, pc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc) - ref_c).abs().max())
        eg = float((ops.agg_fwd_tiled(g.gc, pg, alpha, sda.DST_IS_GENE, G, hc, hg) - ref_g).abs().max())
        print(f'{k:5s} err {ec:.1e}/{eg:.1e} | ' + ' | '.join(out), flush=True)
