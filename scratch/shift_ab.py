"""Round 3: out-of-phase wave halves of agg_tiled_flat4 (graph.TILE_H0_SHARE / WGNN_FLAG_TILE_SHIFT) vs the lock-step kernel,
cfg3 cells<-genes pass, same process / same box, interleaved repetitions.  Results must agree with the lock-step launch."""
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
shares = [None] + [float(x) for x in (sys.argv[1:] or ['0.35', '0.41', '0.47'])]
plans = {}
for sh in shares:
    plans[sh] = GR.build_tile_plan(g.cg, None, None, block_rows=kb, h0_share=sh)
ref = ops.agg_fwd_tiled(g.cg, plans[None], alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
res = {}
for rep in range(3):
    for sh in shares:
        tp = plans[sh]
        out = ops.agg_fwd_tiled(g.cg, tp, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
        err = (out - ref).abs().max().item()
        t = timeit(lambda: ops.agg_fwd_tiled(g.cg, tp, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
        seg = tp.seg_ptr.long(); cnt = (seg[1:] - seg[:-1]).reshape(-1, 16).float()
        h0 = cnt[:, :8].sum().item() / max(1.0, cnt.sum().item())
        res.setdefault(str(sh), []).append(round(t, 4))
        print(f"rep {rep} share {sh}: {t:.4f} ms  max|diff| {err:.2e}  realised h0 share {h0:.3f}  shift={tp.shift}", flush=True)
print(json.dumps(res))
