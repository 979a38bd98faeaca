"""Ablations of the tile kernel with 0 / 2 dedicated loader waves (cfg3 cells<-genes): what is left when the stream, the
barrier or the entry pipeline is switched off (debug instantiation; results are wrong when a switch is set)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C, H = cfg.genes, cfg.cells, 256
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
alpha = torch.rand(G + 2, device=dev) + 0.5
hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
B = lambda *bits: sum(1 << b for b in bits)
plans = {L: GR.build_tile_plan(g.cg, None, None, block_rows=78, n_loaders=L) for L in (0, 2, 3)}
for rep in range(2):
    for nm, fl in [('full', 0), ('nofill', B(16)), ('nobarrier(racy)', B(18)), ('nofill+nobar = compute only', B(16, 18)), ('nocompute', B(17)), ('nocompute+nobar', B(17, 18))]:
        row = []
        for L in (0, 2, 3):
            ops.DEBUG_FLAGS = fl
            row.append(timeit(lambda: ops.agg_fwd_tiled(g.cg, plans[L], alpha, sda.SRC_IS_GENE, G + 1, hg, hc)))
        ops.DEBUG_FLAGS = 0
        print(f"{nm:30s} " + "  ".join(f"L={L}: {t:.3f}" for L, t in zip((0, 2, 3), row)), flush=True)
