"""Records of the other BASELINE configs -> gpurun_out/<WGNN_ROUND_TAG, default r06>_configs.json (copied to profiles/):
  cfg2 forward (hipGraph replay), cfg5 (764,741 cells, fp16-stored features) forward on ONE GPU, cfg4's full-batch training
  step at N = 1, and DeepSortPredictor-shaped inference at atlas scale: a predict graph of 10k support cells + 100k test
  cells over 20k genes, every test cell a seed (predict.py:61-88), the reference's predict-time sizes dense_dim 400 /
  hidden 200 (predict.py:170-172), 1 and 2 layers - against the seeds=None pass over the same graph."""
import json, os, subprocess, sys, time
from pathlib import Path
import torch, torch.nn.functional as F
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
out = {"_how": "python scratch/configs_record.py on one MI355X (gpurun); bench entries are the command's own output line"}
def run(cmd, env=None, timeout=900):
    t = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    return r.stdout, r.stderr, round(time.time() - t, 1)
for cfg, extra in (("cfg2", []), ("cfg5", ["--no-cpu-baseline"])):
    so, se, dt = run([sys.executable, "bench.py", "--config", cfg, "--steps", "20", "--warmup", "3"] + extra)
    line = [l for l in so.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]) if line else {"error": se[-800:]}
    if "roofline" in d:
        d["roofline"].pop("note", None)
    if d.get("cpu_baseline"):
        d["cpu_baseline"].pop("all", None)
    out[cfg + "_forward_1gpu"] = {"cmd": f"python bench.py --config {cfg} --steps 20 --warmup 3 " + " ".join(extra), "wall_s": dt, "line": d}
so, se, dt = run([sys.executable, "bench.py", "--config", "cfg3", "--hidden", "200", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-secondary"])
line = [l for l in so.splitlines() if l.startswith("{")]
d = json.loads(line[-1]) if line else {"error": se[-800:]}
if "roofline" in d:
    d["roofline"].pop("note", None)
out["cfg3_hidden200_forward_1gpu"] = {"cmd": "python bench.py --config cfg3 --hidden 200 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary  (the reference's default hidden_dim, train.py:137)", "wall_s": dt, "line": d}
# the headline workload under BOTH popularity laws: SURVEY 8d's (the default) and rounds 1-5's dense head
both = {}
for pop in ("testis199", "dense_head"):
    so, se, dt = run([sys.executable, "bench.py", "--popularity", pop, "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-secondary"])
    line = [l for l in so.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]) if line else {"error": se[-800:]}
    both[pop] = {"ms_per_step": d.get("ms_per_step"), "cells_per_s": d.get("value"), "roofline_frac": (d.get("roofline") or {}).get("frac"),
                 "dominant_launch_ms": (d.get("roofline") or {}).get("avg_launch_ms"),
                 "passes": [(p["rows"], p["src_rows"], p["avg_ms"]) for p in (d.get("roofline") or {}).get("passes", [])],
                 "generator": (d.get("config") or {}).get("generator"), "workload": (d.get("config") or {}).get("workload")}
out["cfg3_both_popularity_laws"] = {"cmd": "python bench.py --popularity {testis199|dense_head} --steps 20 --warmup 3 --no-cpu-baseline --no-secondary", **both}
so, se, dt = run([sys.executable, "examples/train_sharded.py", "--config", "cfg3", "--steps", "10"])
out["cfg4_full_batch_training_step_1gpu"] = {"cmd": "python examples/train_sharded.py --config cfg3 --steps 10", "wall_s": dt,
                                             "stdout": so.strip().splitlines()[-1] if so.strip() else se[-800:]}
# ---- predictor-shaped inference
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev = torch.device("cuda:0")
G, n_sup, n_test = 20_000, 10_000, 100_000
rp, col, val = S.synth_expression(n_sup + n_test, G, 0.04, device=dev)
mask = torch.zeros(n_sup + n_test, dtype=torch.bool, device=dev); mask[:n_sup] = True
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G, support_mask=mask)
feats = S.synth_features(G + n_sup + n_test, 400, device=dev)
seeds = range(G + n_sup, G + n_sup + n_test)            # what api._predict passes (contiguous block of test cells)
seeds_t = torch.arange(G + n_sup, G + n_sup + n_test, device=dev)
def timed(fn, reps=10):
    for _ in range(3): o = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): o = fn()
    e1.record(); torch.cuda.synchronize()
    return o, e0.elapsed_time(e1) / reps
pred = {}
for L in (1, 2):
    torch.manual_seed(L)
    m = sda.GNN(400, 200, 16, L, G, activation=F.relu, dropout=0.1).to(dev).eval()
    with torch.no_grad():
        m.alpha.uniform_(0.5, 1.5)
        ops.PROFILE = []
        a = m(g, feats, seeds=seeds)
        kern = sorted({dict(zip(t[::2], t[1::2]))["kernel"] for t, _, _ in ops.PROFILE}); ops.PROFILE = None
        a, t_seed = timed(lambda: m(g, feats, seeds=seeds))
        b, t_all = timed(lambda: m(g, feats))
        old, ops.SEED_FULL_PASS_MIN_FRAC = ops.SEED_FULL_PASS_MIN_FRAC, 2.0          # round-2 behaviour: every seed call on the row-wave kernel
        c, t_k1 = timed(lambda: m(g, feats, seeds=seeds_t), reps=3)
        ops.SEED_FULL_PASS_MIN_FRAC = old
        perm = seeds_t[torch.randperm(n_test, device=dev)]
        d, t_perm = timed(lambda: m(g, feats, seeds=perm))
    pred[f"{L}_layer"] = {"ms_seeds_eq_test_cells": round(t_seed, 3), "ms_seeds_none_all_cells": round(t_all, 3),
                          "ms_round2_rowwave_route": round(t_k1, 3), "ms_seeds_eq_test_cells_shuffled_tensor": round(t_perm, 3), "test_cells_per_s": round(n_test / t_seed * 1e3, 1),
                          "kernels_of_the_seed_call": kern, "max_abs_vs_all_cells_pass": float((a - b[n_sup:]).abs().max()),
                          "max_abs_vs_rowwave_route": float((a - c).abs().max())}
out["predictor_shaped_inference"] = {"graph": f"{n_sup} support + {n_test} test cells x {G} genes, nnz {g.cg.nnz}, dense_dim 400, hidden 200",
                                      "call": "GNN.forward(graph, feats, seeds = every test cell)  [DeepSortPredictor.predict, predict.py:61-88]", **pred}
Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / (os.environ.get("WGNN_ROUND_TAG", "r06") + "_configs.json")).write_text(json.dumps(out, indent=1))
print(json.dumps(out["predictor_shaped_inference"], indent=1))
print({k: (v.get("line", {}).get("ms_per_step"), v.get("stdout")) for k, v in out.items() if isinstance(v, dict) and k != "predictor_shaped_inference"})
