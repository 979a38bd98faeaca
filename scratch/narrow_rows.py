"""Round 4: LDS rows of the flat tile kernel packed to D*4 bytes rounded up to 256.  Per-pass times of the
tile kernel with the packed block height vs the round-3 geometry (78 rows per block) vs the row-wave kernel, at cfg2 size and at
a cfg3-size graph for D = 128 / 192 / 200 / 256; whole cfg2 forward replayed as a hipGraph under each dispatch."""
import sys, json, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
from scdeepsort_amd.graphed import GraphedForward
dev = 'cuda:0'
def timeit(f, n=30):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / n * 1e3, 1)
out = {}
for name, widths in (("cfg2", (128, 200)), ("cfg3", (128, 192, 200, 256))):
    cfg = S.CONFIGS[name]; G, C = cfg.genes, cfg.cells
    rp, col, val = S.synth_expression(C, G, device=dev)
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
    alpha = torch.rand(G + 2, device=dev) + 0.5
    for D in widths:
        hg = S.synth_features(G, D, device=dev); hc = S.synth_features(C, D, seed=3, device=dev)
        rec = {}
        if name == "cfg2":
            ops.TILED_MIN_WORK = None
            rec["row_wave"] = (timeit(lambda: ops.agg_fwd(g.cg, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)),
                               timeit(lambda: ops.agg_fwd(g.gc, alpha, sda.DST_IS_GENE, G, hc, hg)))
        for kb in sorted({78, ops.tiled_block_rows(D)}):
            tpc = GR.build_tile_plan(g.cg, None, None, block_rows=kb, n_loaders=GR.TILE_LOADER_WAVES)
            tpg = GR.build_tile_plan(g.gc, None, None, block_rows=kb, n_loaders=GR.TILE_LOADER_WAVES)
            rec[f"tiled_kb{kb}"] = (timeit(lambda: ops.agg_fwd_tiled(g.cg, tpc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)),
                                    timeit(lambda: ops.agg_fwd_tiled(g.gc, tpg, alpha, sda.DST_IS_GENE, G, hc, hg)),
                                    f"{tpc.n_row_tiles}x{tpc.n_col_splits} L{tpc.n_loaders} / {tpg.n_row_tiles}x{tpg.n_col_splits} L{tpg.n_loaders}")
        out[f"{name} D={D} (cells<-genes us, genes<-cells us)"] = rec
        print(name, D, rec, flush=True)
    if name == "cfg2":
        for hidden in (128, 200):
            torch.manual_seed(1)
            m = sda.GNN(cfg.dense_dim, hidden, cfg.n_classes, 2, G, activation=F.relu).to(dev).eval()
            feats = S.synth_features(G + C, cfg.dense_dim, device=dev)
            for label, thr, narrow in (("row_wave", None, True), ("tiled_packed", 1, True), ("tiled_kb78", 1, False)):
                ops.TILED_MIN_WORK, ops.NARROW_LDS_ROWS = thr, narrow
                g.cg._tile_plan = None; g.gc._tile_plan = None
                gf = GraphedForward(m, g, feats)
                out[f"cfg2 hidden={hidden} forward, hipGraph replay, {label} (us)"] = timeit(lambda: gf(), 50)
                print(hidden, label, out[f"cfg2 hidden={hidden} forward, hipGraph replay, {label} (us)"], flush=True)
            ops.TILED_MIN_WORK, ops.NARROW_LDS_ROWS = 500_000_000, True
    del g
print(json.dumps(out))
