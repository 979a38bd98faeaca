"""ring3 (3 x 48-row LDS buffers, DMA two blocks ahead) vs the production 2 x 78 rows, cfg3, both passes; + parity."""
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
ref = {}
for rep in range(2):
    for name, kb, fl in (("2x78", 78, 0), ("2x48", 48, 0), ("ring3 3x48", 48, 1 << 22)):
        ops.DEBUG_FLAGS = fl
        tpc = GR.build_tile_plan(g.cg, None, None, block_rows=kb); tpg = GR.build_tile_plan(g.gc, None, None, block_rows=kb)
        fc = lambda: ops.agg_fwd_tiled(g.cg, tpc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
        fg = lambda: ops.agg_fwd_tiled(g.gc, tpg, alpha, sda.DST_IS_GENE, G, hc, hg)
        oc, og = fc(), fg()
        if "c" not in ref: ref["c"], ref["g"] = oc, og
        tc, tg = timeit(fc), timeit(fg)
        print(f"{name:11s} cells {tc:.3f} ms  genes {tg:.3f} ms   max|diff| {float((oc-ref['c']).abs().max()):.1e} {float((og-ref['g']).abs().max()):.1e}", flush=True)
ops.DEBUG_FLAGS = 0
