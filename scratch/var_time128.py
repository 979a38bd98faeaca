"""Per-pass time of the flat tile kernel at D = 128 (LDS row stride 512 B, 156-row blocks) for a variant library:
cfg2 operands (10k x 5k) and a cfg3-size graph.  usage: var_time128.py LIB"""
import sys, torch
sys.path.insert(0, '/root/repo')
from pathlib import Path
from scdeepsort_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = Path(sys.argv[1]).resolve()
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev = 'cuda:0'; D = 128
def timeit(f, n=30):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / n * 1e3, 1)
out = []
for name in ("cfg2", "cfg3"):
    cfg = S.CONFIGS[name]; G, C = cfg.genes, cfg.cells
    rp, col, val = S.synth_expression(C, G, device=dev)
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
    alpha = torch.rand(G + 2, device=dev) + 0.5
    hg = S.synth_features(G, D, device=dev); hc = S.synth_features(C, D, seed=3, device=dev)
    kb = ops.tiled_block_rows(D)
    tpc, tpg = g.cg.tile_plan(kb), g.gc.tile_plan(kb)
    saved, ops.TILED_MIN_WORK = ops.TILED_MIN_WORK, None
    ref = ops.agg_fwd(g.cg, alpha, sda.SRC_IS_GENE, G + 1, hg, hc); ops.TILED_MIN_WORK = saved
    err = float((ref - ops.agg_fwd_tiled(g.cg, tpc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)).abs().max())
    tc = timeit(lambda: ops.agg_fwd_tiled(g.cg, tpc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
    tg = timeit(lambda: ops.agg_fwd_tiled(g.gc, tpg, alpha, sda.DST_IS_GENE, G, hc, hg))
    out.append(f"{name} D=128 cells/genes {tc}/{tg} us err {err:.1e}")
    del g
print(Path(sys.argv[1]).stem if len(sys.argv) > 1 else 'default', ' | '.join(out), flush=True)
