"""Round 3: dedicated loader waves of agg_tiled_flat4 (graph.TILE_LOADER_WAVES / plan n_loaders) vs every wave streaming its
share - cfg3 cells<-genes pass, same process, interleaved repetitions; results must equal the symmetric launch bit for bit."""
import sys, json, torch
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
kb = 78
GR.LOADER_MIN_ENTRIES = {1: 0.0, 2: 0.0, 3: 0.0}          # no guard: measure every requested loader count
Ls = [x for x in (sys.argv[1:] or ['0', '1', '2'])]
plans = {}
for L in Ls:
    plans[L] = GR.build_tile_plan(g.cg, None, None, block_rows=kb, n_loaders=int(str(L).rstrip('ps')))
    seg = plans[L].seg_ptr.long(); cnt = (seg[1:] - seg[:-1]).reshape(-1, 16).float().sum(0)
    print(L, "share of the edges per wave:", [round(x, 3) for x in (cnt / cnt.sum()).tolist()],
          "per SIMD group:", [round(float(cnt[q::4].sum() / cnt.sum()), 3) for q in range(4)], flush=True)
ref = ops.agg_fwd_tiled(g.cg, plans[Ls[0]], alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
res = {}
for rep in range(int(__import__('os').environ.get('REPS', '5'))):
    for L in Ls:
        tp = plans[L]
        ops.LOADER_PRIO = 1 if str(L).endswith('p') else 0
        out = ops.agg_fwd_tiled(g.cg, tp, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
        err = (out - ref).abs().max().item()
        t = timeit(lambda: ops.agg_fwd_tiled(g.cg, tp, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
        ops.DEBUG_FLAGS = 1 << 17                                    # no compute: stream + barriers only
        tf = timeit(lambda: ops.agg_fwd_tiled(g.cg, tp, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
        ops.DEBUG_FLAGS = 0
        res.setdefault(str(L), []).append((round(t, 4), round(tf, 4)))
        print(f"rep {rep} loaders {L} (plan says {tp.n_loaders}): {t:.4f} ms  (no-compute {tf:.4f})  max|diff| {err:.2e}", flush=True)
print(json.dumps(res))
