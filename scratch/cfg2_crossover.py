"""cfg2 (10k cells x 5k genes, 2.0 M edges, hidden 128): row-wave K1 vs the tile kernel per pass, and whole forward."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
def timeit(f, n=30):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name in ("cfg2",):
    cfg = S.CONFIGS[name]; G, C, H = cfg.genes, cfg.cells, cfg.hidden
    rp, col, val = S.synth_expression(C, G, device=dev)
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
    alpha = torch.rand(G + 2, device=dev) + 0.5
    for D in (128, 256):
        hg = S.synth_features(G, D, device=dev); hc = S.synth_features(C, D, seed=3, device=dev)
        ops.TILED_MIN_WORK = None
        k1c = timeit(lambda: ops.agg_fwd(g.cg, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)); k1g = timeit(lambda: ops.agg_fwd(g.gc, alpha, sda.DST_IS_GENE, G, hc, hg))
        res = []
        for geom in ((None, None), (40, 6), (40, 3), (20, 12), (40, 1)):
            try:
                tpc = GR.build_tile_plan(g.cg, geom[0], geom[1], block_rows=78); tpg = GR.build_tile_plan(g.gc, geom[0], geom[1], block_rows=78)
            except Exception as e:
                res.append((geom, "n/a")); continue
            tc = timeit(lambda: ops.agg_fwd_tiled(g.cg, tpc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)); tg = timeit(lambda: ops.agg_fwd_tiled(g.gc, tpg, alpha, sda.DST_IS_GENE, G, hc, hg))
            res.append((f"{tpc.n_row_tiles}x{tpc.n_col_splits}/{tpg.n_row_tiles}x{tpg.n_col_splits}", round(tc, 1), round(tg, 1)))
        print(f"{name} D={D}: K1 cells {k1c:.1f} us genes {k1g:.1f} us | tiled {res}", flush=True)
    m = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, 2, G, activation=F.relu).to(dev).eval()
    feats = S.synth_features(G + C, cfg.dense_dim, device=dev)
    for thr in (500_000_000, 1):
        ops.TILED_MIN_WORK = thr
        with torch.no_grad():
            t = timeit(lambda: m(g, feats))
        print(f"{name} forward with TILED_MIN_WORK={thr}: {t:.1f} us", flush=True)
