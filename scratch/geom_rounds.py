"""genes<-cells pass: one round of <= 256 tiles vs the round-1 heuristic (~nnz/50k tiles), over operand sizes."""
import sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
H = 256
for C, G in ((20_000, 15_000), (50_000, 20_000), (100_000, 20_000), (200_000, 20_000)):
    rp, col, val = S.synth_expression(C, G, device=dev)
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
    alpha = torch.rand(G + 2, device=dev) + 0.5
    hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
    kb = ops.tiled_block_rows(H)
    for name, thr in (("one_round", 0), ("old", 10 ** 12)):
        GR.ONE_ROUND_MIN_NNZ = thr
        tp = GR.build_tile_plan(g.gc, None, None, block_rows=kb)
        f = lambda: ops.agg_fwd_tiled(g.gc, tp, alpha, sda.DST_IS_GENE, G, hc, hg)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"C={C} G={G} nnz={g.gc.nnz/1e6:.1f}M {name:9s} tiles={tp.n_row_tiles}x{tp.n_col_splits} ms={e0.elapsed_time(e1)/10:.3f}", flush=True)
    del g, rp, col, val
