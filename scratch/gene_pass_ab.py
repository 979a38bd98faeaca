"""genes<-cells pass at cfg3: launch order (XCD-aware vs round-1 split-major) x column splits.  Times the tile kernel +
its finalize with HIP events; run under rocprofv3 --pmc for the traffic of a chosen variant (WGNN_AB_ONLY=order:splits)."""
import os, sys, time, torch
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
ref = None
only = os.environ.get("WGNN_AB_ONLY")
variants = [(o, rt, sp) for o in ("split_major", "xcd") for (rt, sp) in ((85, 3), (80, 16), (None, None))]
if only:
    o, rt, sp = only.split(":"); variants = [(o, int(rt) if rt != "None" else None, int(sp) if sp != "None" else None)]
for order, rt, sp in variants:
    GR.TILE_ORDER = order
    tp = GR.build_tile_plan(g.gc, rt, sp, block_rows=kb)
    f = lambda: ops.agg_fwd_tiled(g.gc, tp, alpha, sda.DST_IS_GENE, G, hc, hg)
    out = f(); torch.cuda.synchronize()
    if ref is None:
        ref = out
    err = float((out - ref).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 3 if only else 10
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"order={order:11s} row_tiles={tp.n_row_tiles:4d} splits={tp.n_col_splits:3d} tiles={tp.n_tiles:5d} partial_MB={tp.n_partials * H * 4 / 1e6:7.1f} "
          f"ms={e0.elapsed_time(e1) / n:.3f} max|diff|={err:.2e}", flush=True)
