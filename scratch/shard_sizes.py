"""Round 4: what ONE rank of the strong-scaling job does at N = 1/2/4/8 - rank 0's shard of the cfg3 graph
(C/N cells, gene side normalised with the GLOBAL statistics), the sharded branch of the engine without a process group (the
collectives are skipped, everything else is the production path) - per-kernel HIP-event times, geometry chosen by
auto_tile_geometry, and the forward time.  -> gpurun_out/r06_shard_sizes.json"""
import json, sys, time, torch, torch.nn.functional as F
from pathlib import Path
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, dist as D
from scdeepsort_amd.sharded import ShardedWgnn
from scdeepsort_amd import tuning
import os
OUT = '/root/repo/gpurun_out/' + os.environ.get('WGNN_SHARD_SIZES_OUT', 'r06_shard_sizes.json')
TUNED = os.environ.get('TUNED', '1') == '1' and tuning.use_tuned_gemms()      # tracked per-shape GEMM picks (what bench.py runs with)
dev = torch.device('cuda:0')
cfg = S.CONFIGS['cfg3']; G = cfg.genes
rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, seed=S.REFERENCE_SEED, device=dev)
feats_g = S.synth_features(G, cfg.dense_dim, seed=7, device=dev)
feats_c = S.synth_features(cfg.cells, cfg.dense_dim, seed=100, device=dev)
torch.manual_seed(1234)
model = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, 2, G, activation=F.relu).to(dev).eval()
gdeg, gsum = ShardedWgnn.gene_stats(col, val, G)
out = {}
for N in (1, 2, 4, 8):
    lo, hi = D.shard_range(cfg.cells, 0, N)
    b, e = int(rp[lo]), int(rp[hi])
    eng = ShardedWgnn.build(model, (rp[lo:hi + 1] - rp[lo]).clone(), col[b:e].clone(), val[b:e].clone(), G,
                            global_stats=(gdeg, gsum) if N > 1 else None)
    eng.shard_sizes = [hi - lo]
    fc = feats_c[lo:hi].contiguous()
    def step():
        with torch.no_grad():
            return eng.forward(feats_g, fc, gather_logits=False)
    for _ in range(3): step()
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()                      # the forward time: no per-launch events in the stream
    e1.record(); torch.cuda.synchronize()
    ops.PROFILE = []
    for _ in range(n): step()                      # a second pass with HIP events around every aggregation launch (per-pass table)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    per = {}
    for tag, a, b_ in prof:
        d = dict(zip(tag[::2], tag[1::2]))
        per.setdefault((d['kernel'], d['rows'], d['cols'], d['nnz']), []).append(a.elapsed_time(b_))
    g = eng.graph
    rec = {"cells_this_rank": hi - lo, "nnz": g.cg.nnz, "ms_per_forward_compute_only": round(e0.elapsed_time(e1) / n, 4),
           "passes": [{"kernel": k[0], "rows": k[1], "src": k[2], "nnz": k[3], "per_step": len(v) // n, "avg_ms": round(sum(v) / len(v), 4)} for k, v in per.items()]}
    rec["cg_cu_budget"] = g.cg.cu_budget
    hog_lib = Path('/root/repo/scratch/variants/libhog.so')
    if hog_lib.exists():            # the same forward next to 32 CUs held by a spin kernel for its whole duration (scratch/hog.hip)
        import ctypes
        hog = ctypes.CDLL(str(hog_lib)); hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
        side = torch.cuda.Stream(device=dev)
        ts = []
        for _ in range(9):
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            hog.hog_launch(32, int(rec["ms_per_forward_compute_only"] * 1e-3 * 100e6 * 3), side.cuda_stream)
            torch.cuda._sleep(20_000)
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(); step(); a1.record(); torch.cuda.synchronize()
            ts.append(a0.elapsed_time(a1))
        rec["ms_per_forward_next_to_32_held_CUs"] = round(sorted(ts)[len(ts) // 2], 4)
    for name, csr in (("cg", g.cg), ("gc", g.gc)):
        tp = csr._tile_plan
        if tp:
            p = list(tp.values())[0]
            rec[name + "_tiles"] = f"{p.n_row_tiles} x {p.n_col_splits}, partial rows {p.n_partials}"
    rec["ideal_ms"] = None
    out[f"N={N}"] = rec
    print(N, json.dumps(rec), flush=True)
    del eng
base = out["N=1"]["ms_per_forward_compute_only"]
for N in (1, 2, 4, 8):
    out[f"N={N}"]["ideal_ms"] = round(base / N, 4)
    out[f"N={N}"]["compute_scaling_efficiency"] = round(base / N / out[f"N={N}"]["ms_per_forward_compute_only"], 3)
json.dump(out, open(OUT, 'w'), indent=1)
out["_gemm_selection"] = "tuned picks (scdeepsort_amd/tuned_gemms_gfx950.csv)" if TUNED else "library heuristics"
json.dump(out, open(OUT, 'w'), indent=1)
print(json.dumps({k: (v["ms_per_forward_compute_only"], v["compute_scaling_efficiency"]) for k, v in out.items() if k.startswith("N=")}))
