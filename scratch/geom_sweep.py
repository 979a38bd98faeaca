"""Tile geometry sweep of the genes<-cells pass (cfg3, SURVEY 8d's graph): (row tiles, column splits) around the heuristic's pick."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = "cuda:0"
cfg = S.CONFIGS["cfg3"]; G, C, H = cfg.genes, cfg.cells, cfg.hidden
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
alpha = torch.rand(G + 2, device=dev) + 0.5
hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
kb = ops.tiled_block_rows(H)
def t(csr, tp, mode, sidx, src, slf, n=12):
    for _ in range(3): ops.agg_fwd_tiled(csr, tp, alpha, mode, sidx, src, slf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): ops.agg_fwd_tiled(csr, tp, alpha, mode, sidx, src, slf)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
auto = g.gc.tile_plan(kb)
print("gene side heuristic: %d x %d tiles, loaders %d: %.4f ms" % (auto.n_row_tiles, auto.n_col_splits, auto.n_loaders, t(g.gc, auto, sda.DST_IS_GENE, G, hc, hg)), flush=True)
tp85 = GR.build_tile_plan(g.gc, 85, 3, block_rows=kb, n_loaders=1)
for rep in range(4):
    print("  rep %d: heuristic %.4f ms (virtual rows -> partial rows %d)   explicit 85 x 3 %.4f ms (partial rows %d)" % (
        rep, t(g.gc, auto, sda.DST_IS_GENE, G, hc, hg), auto.n_partials, t(g.gc, tp85, sda.DST_IS_GENE, G, hc, hg), tp85.n_partials), flush=True)
for share in (0.4, 0.5, 0.65, 0.8, 1.0):
    saved, GR.VIRTUAL_ROW_SHARE = GR.VIRTUAL_ROW_SHARE, share
    try:
        tp = GR.build_tile_plan(g.gc, 85, 3, block_rows=kb, n_loaders=1)
        print("  share %.2f: loaders %d, partial rows %d: %.4f ms" % (share, tp.n_loaders, tp.n_partials, t(g.gc, tp, sda.DST_IS_GENE, G, hc, hg)), flush=True)
    finally:
        GR.VIRTUAL_ROW_SHARE = saved
auto = g.cg.tile_plan(kb)
print("cells side heuristic: %d x %d tiles, loaders %d: %.4f ms" % (auto.n_row_tiles, auto.n_col_splits, auto.n_loaders, t(g.cg, auto, sda.SRC_IS_GENE, G + 1, hg, hc)), flush=True)
