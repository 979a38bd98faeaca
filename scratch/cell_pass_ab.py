"""cells<-genes pass at cfg3: row-tile count sweep (one launch = n_row_tiles workgroups, no column split)."""
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
kb = ops.tiled_block_rows(H)
for rt, sp in ((512, 1), (400, 1), (448, 1), (768, 1), (256, 2), (512, 2), (391, 1)):
    tp = GR.build_tile_plan(g.cg, rt, sp, block_rows=kb)
    f = lambda: ops.agg_fwd_tiled(g.cg, tp, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"row_tiles={tp.n_row_tiles:4d} splits={tp.n_col_splits} rows/tile={C / tp.n_row_tiles:.0f} ms={e0.elapsed_time(e1) / 10:.3f}", flush=True)
