"""cells<-genes pass at cfg3 operands through the flat tile kernel at native widths vs the round-1 zero-padding to 256."""
import sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C = cfg.genes, cfg.cells
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
alpha = torch.rand(G + 2, device=dev) + 0.5
tpc = g.cg.tile_plan(78); tpg = g.gc.tile_plan(78)
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for D in (256, 208, 200, 128, 64, 256):
    hg = S.synth_features(G, D, device=dev); hc = S.synth_features(C, D, seed=3, device=dev)
    tc = timeit(lambda: ops.agg_fwd_tiled(g.cg, tpc, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
    tg = timeit(lambda: ops.agg_fwd_tiled(g.gc, tpg, alpha, sda.DST_IS_GENE, G, hc, hg))
    print(f"D={D:4d}  cells<-genes {tc:.3f} ms   genes<-cells {tg:.3f} ms", flush=True)
