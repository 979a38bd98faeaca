"""Round 4: ONE rank's shard of the strong-scaling cfg3 job (rank 0 of N; collectives skipped - the 1-GPU lease has no peer),
eager and as a captured hipGraph; run under rocprofv3 --kernel-trace --stats for the per-kernel table of the shard."""
import os, sys, json, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, dist as D
from scdeepsort_amd.sharded import ShardedWgnn
from scdeepsort_amd import tuning
TUNED = os.environ.get('TUNED', '0') == '1' and tuning.use_tuned_gemms()
if os.environ.get('LT', '0') == '1':
    torch.backends.cuda.preferred_blas_library("hipblaslt")
N = int(os.environ.get('N', '8')); steps = int(os.environ.get('STEPS', '50'))
dev = torch.device('cuda:0')
cfg = S.CONFIGS['cfg3']; G = cfg.genes
rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, seed=S.REFERENCE_SEED, device=dev)
feats_g = S.synth_features(G, cfg.dense_dim, seed=7, device=dev)
feats_c = S.synth_features(cfg.cells, cfg.dense_dim, seed=100, device=dev)
torch.manual_seed(1234)
model = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, 2, G, activation=F.relu).to(dev).eval()
gdeg, gsum = ShardedWgnn.gene_stats(col, val, G)
lo, hi = D.shard_range(cfg.cells, 0, N)
b, e = int(rp[lo]), int(rp[hi])
eng = ShardedWgnn.build(model, (rp[lo:hi + 1] - rp[lo]).clone(), col[b:e].clone(), val[b:e].clone(), G, global_stats=(gdeg, gsum))
fc = feats_c[lo:hi].contiguous()
del rp, col, val, feats_c
def step():
    with torch.no_grad():
        return eng.forward(feats_g, fc, gather_logits=False)
def timeit(f, n):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
res = {"N": N, "cells": hi - lo, "tuned": bool(TUNED), "hipblaslt_preferred": os.environ.get("LT", "0") == "1"}
for fold in (False, True):
    model.fold_alpha = fold
    res[f"eager_ms_fold{int(fold)}"] = round(timeit(step, steps), 4)
from scdeepsort_amd.graphed import GraphedShardedForward
gsf = GraphedShardedForward(eng, feats_g, fc, gather_logits=False)
res["graphed_ms"] = round(timeit(lambda: gsf(), steps), 4)
res["graphed_equals_eager"] = bool(torch.equal(gsf(), step()))
print(json.dumps(res))
