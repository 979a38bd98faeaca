"""Row-wave K1 at cfg2 (10k x 5k, D = 128) and on a 100k-edge seed batch: gathers in flight per lane (variant builds)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from pathlib import Path
from scdeepsort_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = Path(sys.argv[1]).resolve()
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev = 'cuda:0'
def timeit(f, n=50):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
cfg = S.CONFIGS['cfg2']; G, C = cfg.genes, cfg.cells
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
alpha = torch.rand(G + 2, device=dev) + 0.5
ops.TILED_MIN_WORK = None
out = []
for D in (128, 256, 64):
    hg = S.synth_features(G, D, device=dev); hc = S.synth_features(C, D, seed=3, device=dev)
    tc = timeit(lambda: ops.agg_fwd(g.cg, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)); tg = timeit(lambda: ops.agg_fwd(g.gc, alpha, sda.DST_IS_GENE, G, hc, hg))
    out.append(f"D={D}: cells {tc:.1f} genes {tg:.1f} us")
print(Path(sys.argv[1]).stem if len(sys.argv) > 1 else 'default', ' | '.join(out), flush=True)
