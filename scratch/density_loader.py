"""Dedicated loader waves vs symmetric waves over edge density (cfg3 node counts): where is the crossover?"""
import sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
G, C, H = 20000, 100000, 256
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
GR.LOADER_MIN_ENTRIES = {1: 0.0, 2: 0.0, 3: 0.0}          # no guard: measure both sides of it
for dens in (0.005, 0.01, 0.02, 0.03, 0.04, 0.08):
    rp, col, val = S.synth_expression(C, G, dens, device=dev)
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
    alpha = torch.rand(G + 2, device=dev) + 0.5
    hg = S.synth_features(G, H, device=dev); hc = S.synth_features(C, H, seed=3, device=dev)
    row = []
    for name, csr, mode, si, src, slf in (("cells", g.cg, sda.SRC_IS_GENE, G + 1, hg, hc), ("genes", g.gc, sda.DST_IS_GENE, G, hc, hg)):
        for L in (0, 1, 2):
            tp = GR.build_tile_plan(csr, None, None, block_rows=78, n_loaders=L)
            nblk = tp.nblk_max
            e = csr.nnz / (tp.n_tiles * nblk * (16 - tp.n_loaders))
            t = min(timeit(lambda: ops.agg_fwd_tiled(csr, tp, alpha, mode, si, src, slf)) for _ in range(2))
            row.append(f"{name} L={tp.n_loaders} ({tp.n_row_tiles}x{tp.n_col_splits}, {e:.0f}/wave-block): {t:.3f}")
    print(f"density {dens}: nnz {g.cg.nnz/1e6:.1f} M | " + " | ".join(row), flush=True)
    del g
