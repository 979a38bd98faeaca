"""Round 3: shared pairs of agg_tiled_flat4 (graph.TILE_SHARED_PAIRS: two entries of a (wave, block) segment on the same
source row take one LDS read) vs the slot-sorted entries of before - cfg3, both directions, same process, interleaved
repetitions; parity of each against the row-wave kernel."""
import sys, json, os, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import _lib
if os.environ.get('WGNN_LIB'):
    from pathlib import Path
    _lib.LIB_PATH = Path(os.environ['WGNN_LIB'])              # A/B against another build of the library
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
cfg = S.CONFIGS[os.environ.get('CFG', 'cfg3')]; G, C, H = cfg.genes, cfg.cells, 256
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
kb = 78
res = {}
for name, csr, kind, self_idx, src, slf in (("cells<-genes", g.cg, sda.SRC_IS_GENE, G + 1, hg, hc),
                                             ("genes<-cells", g.gc, sda.DST_IS_GENE, G, hc, hg)):
    plans = {}
    for pairs in (False, True):
        GR.TILE_SHARED_PAIRS = pairs
        tp = GR.build_tile_plan(csr, None, None, block_rows=kb, n_loaders=GR.TILE_LOADER_WAVES)
        plans[pairs] = tp
        m = tp.entries[:, 0]
        print(name, "pairs" if pairs else "plain", "entries", tp.entries.shape[0], "in pairs", int((m < 0).sum()),
              "pads", int(((m & GR.TILE_PAD_FLAG) != 0).sum()), "loaders", tp.n_loaders, flush=True)
    tmw, ops.TILED_MIN_WORK = ops.TILED_MIN_WORK, None                       # the reference: the row-wave kernel (no tile dispatch)
    ref = ops.agg_fwd(csr, alpha, kind, self_idx, src, slf)
    ops.TILED_MIN_WORK = tmw
    for rep in range(int(os.environ.get('REPS', '4'))):
        for pairs in (False, True):
            tp = plans[pairs]
            out = ops.agg_fwd_tiled(csr, tp, alpha, kind, self_idx, src, slf)
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            t = timeit(lambda: ops.agg_fwd_tiled(csr, tp, alpha, kind, self_idx, src, slf))
            res.setdefault(f"{name} {'pairs' if pairs else 'plain'}", []).append(round(t, 4))
            print(f"rep {rep} {name} {'pairs' if pairs else 'plain'}: {t:.4f} ms  rel max|diff vs row-wave| {err:.2e}", flush=True)
print(json.dumps(res))
