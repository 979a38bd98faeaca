#!/usr/bin/env python
"""bench.py - cells embedded / s for the 2-layer WGNN forward (BASELINE.json metric) on N MI355X.

A "step" = one full 2-layer forward (L1 genes<-cells, L1 cells<-genes, L2 cells<-genes, projections, head)
over one synthetic graph already resident in HBM.  The workload is BASELINE cfg3 (100k cells x 20k genes,
dense_dim 400, hidden 256, 16 classes) at every N: with N > 1 the SAME graph (generated from the reference
seed on every rank) is sharded along the cell axis - `"scaling": "strong"`, BASELINE cfg4's split - the gene
table is replicated, ONE data-path collective per forward (all-reduce of the [G,H] gene partial sums) + the
logits all-gather, and rank 0 checks the sharded logits against the unsharded evaluation of the same graph
(`config.sharded_vs_unsharded`).  `--scaling weak` gives every rank its own cfg-sized shard instead; in a
strong N > 1 run the weak line is measured too and printed as the secondary field `weak_scaling`.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
algorithmic bytes / HIP-event launch time) and `cpu_baseline` (the CPU restatement timed on this host).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable


def pass_bytes(nnz, R, S, D, G, s=4):
    """Algorithmic HBM bytes of ONE aggregation launch: CSR col+val once, rowptr,
    every source row once, every self row once, alpha, output once."""
    return 8 * nnz + 4 * (R + 1) + s * D * (S + R) + 4 * G + s * D * R


def cpu_baseline(graph, model, feats_g, feats_c, gpu_logits, cfg):
    """Times SURVEY 8d's CPU baselines on this host's cores (the reference itself cannot run: DGL 0.4.3 absent, so all
    are kind = "port"): the C/OpenMP restatement in the reference's arithmetic order, B2 = torch CSR SpMM + F.linear
    over the full graph, B1 = reference-style execution (materialised [E_b, D] messages per 500-seed batch, gnn.py:47-65 /
    train.py:71-80) on a few 1-hop batches, extrapolated.  The STRONGEST one is the quoted value.  Also returns the
    GPU-vs-CPU max abs error on the full logits."""
    import numpy as np
    import scipy.sparse as sp
    from oracle import c_oracle as CO, cpu_baselines as CB, wgnn_oracle as O
    G, C = graph.num_genes, graph.num_cells
    cg = graph.cg
    # the oracle gets the SAME normalised operand (device K4 output), rebuilt as scipy CSR on the host
    A_cg = sp.csr_matrix((cg.val.cpu().numpy(), cg.col.cpu().numpy(), cg.rowptr.cpu().numpy()), shape=(C, G))
    gc = graph.gc
    A_gc = sp.csr_matrix((gc.val.cpu().numpy(), gc.col.cpu().numpy(), gc.rowptr.cpu().numpy()), shape=(G, C))
    ocg = O.CsrGraph(G, C, A_cg, A_gc, np.diff(A_cg.indptr) + 1, np.diff(A_gc.indptr) + 1)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    feats = np.concatenate([feats_g.float().cpu().numpy(), feats_c.float().cpu().numpy()])   # fp16 storage: same rounded inputs
    gpu = gpu_logits.cpu().numpy()
    threads = torch.get_num_threads()

    def best_of(fn, max_reps, budget_s):
        reps, t_best, res, t_all = 0, 1e30, None, time.perf_counter()
        while reps < max_reps and (reps == 0 or time.perf_counter() - t_all < budget_s):
            t0 = time.perf_counter(); res = fn(); t_best = min(t_best, time.perf_counter() - t0); reps += 1
        return t_best, reps, res

    every = []
    t, reps, logits = best_of(lambda: CO.forward(sd, ocg, feats, model.n_layers), 3, 12.0)
    err = float(np.abs(logits - gpu).max())
    every.append({"name": "port_c_openmp", "value": round(C / t, 1), "unit": "cells/s", "cores": CO.num_threads(),
                  "s_per_forward": round(t, 3), "sample": f"full {cfg.name} graph, best of {reps}; C/OpenMP aggregation "
                  "(reference multiply order, aggregate-first) + torch Linear", "max_abs_err_vs_gpu": err})
    t, reps, l1 = best_of(lambda: CO.forward(sd, ocg, feats, model.n_layers, order="project_first"), 3, 8.0)
    every.append({"name": "port_c_openmp_project_first", "value": round(C / t, 1), "unit": "cells/s", "cores": CO.num_threads(),
                  "s_per_forward": round(t, 3), "sample": f"full {cfg.name} graph, best of {reps}; C/OpenMP aggregation of the "
                  "projected H-wide rows (the GPU path's order) + torch Linear", "max_abs_err_vs_gpu": float(np.abs(l1 - gpu).max())})
    t, reps, l3 = best_of(lambda: CO.forward(sd, ocg, feats, model.n_layers, order="project_first_blocked"), 5, 6.0)
    every.append({"name": "port_c_openmp_cache_blocked", "value": round(C / t, 1), "unit": "cells/s", "cores": CO.num_threads(),
                  "s_per_forward": round(t, 4), "sample": f"full {cfg.name} graph, best of {reps}; C/OpenMP aggregation tiled like the "
                  "GPU kernel (128-256 destination rows x 256 source rows per step, accumulators and the source block L2-resident, "
                  "AVX-512 body picked at load time) on the projected H-wide rows + torch Linear",
                  "max_abs_err_vs_gpu": float(np.abs(l3 - gpu).max())})
    t, reps, l2 = best_of(lambda: CB.b2_torch_csr_forward(sd, ocg, feats, model.n_layers), 3, 10.0)
    every.append({"name": "B2_torch_csr_spmm", "value": round(C / t, 1), "unit": "cells/s", "cores": threads,
                  "s_per_forward": round(t, 3), "sample": f"full {cfg.name} graph, best of {reps}; torch.sparse CSR SpMM + "
                  "F.linear, project-first", "max_abs_err_vs_gpu": float(np.abs(l2 - gpu).max())})
    H = model.layers[0].fc_neigh.weight.shape[0]
    b1 = CB.b1_reference_style(sd, ocg, feats, model.n_layers, H, batch=500, max_batches=8, budget_s=8.0)
    every.append({"name": "B1_reference_style", "value": round(C / b1["s_per_forward"], 1), "unit": "cells/s", "cores": threads,
                  "s_per_forward": round(b1["s_per_forward"], 2),
                  "sample": f"{b1['batches']} 1-hop seed batches of {b1['batch']} cells ({b1['edges']} edges, "
                            f"{b1['s_measured']:.2f} s) with materialised [E_b, D] messages + index_add_ (gnn.py:47-65, "
                            f"train.py:71-80 restated, not DGL); EXTRAPOLATED by seconds per edge-float to the "
                            f"{model.n_layers}-layer forward", "restatements_agree_max_abs": b1["check_err"]})
    # algorithmic flops of one forward (F_pass = 2 (nnz + R) D + 2 R D_in D_out, project-first widths)
    Hh, Din = model.layers[0].fc_neigh.weight.shape
    fl, d_in = 0.0, Din
    for i in range(model.n_layers):
        fl += 2.0 * (G + C) * d_in * Hh + 2.0 * (A_cg.nnz + C) * Hh + (2.0 * (A_gc.nnz + G) * Hh if i < model.n_layers - 1 else 0.0)
        d_in = Hh
    for e in every:
        e["GFLOPs"] = round(fl / e["s_per_forward"] / 1e9, 1)
    top = max(every, key=lambda e: e["value"])
    return {"value": top["value"], "unit": "cells/s", "cores": top["cores"], "kind": "port", "strongest": top["name"],
            "sample": top["sample"] + f"; host has {os.cpu_count()} logical CPUs", "gpu_vs_cpu_max_abs_err": err,
            "GFLOPs": top["GFLOPs"], "forward_GFLOP": round(fl / 1e9, 1),
            "all": every}


def describe_workload(col, cells, genes, popularity, S):
    """What was generated, measured on the operand itself: inclusion frequency (share of this rank's cells expressing the gene) at
    SURVEY 8d's five popularity ranks - taken at the same QUANTILE of the gene list (rank * G / 9339) and at the absolute rank."""
    pop = torch.sort(torch.bincount(col.long(), minlength=genes).double() / max(cells, 1), descending=True).values

    def at(r):
        lo, hi = max(0, int(r / 1.1) - 1), min(genes, int(r * 1.1) + 1)
        return round(float(pop[lo:hi].mean()), 4)
    ranks = (1, 10, 100, 1000, 3000)
    return {"popularity": popularity,
            "rng": "numpy.random.default_rng(seed), seed 10086 (+ rank in weak mode)" if popularity == "testis199" else "torch device generator",
            "survey_8d_inclusion_at_rank_1_10_100_1000_3000_of_9339": [0.98, 0.67, 0.32, 0.09, 0.035],
            "inclusion_at_same_quantile": [at(max(1, r * genes / S.TESTIS_GENES)) for r in ranks],
            "inclusion_at_absolute_rank": [at(r) for r in ranks],
            "median_gene_inclusion": round(float(pop[genes // 2]), 4),
            "top256_genes_edge_share": round(float(pop[:256].sum() / pop.sum()), 4)}


def shared_pair_share(plan):
    """Share of a tile plan's entries that ride the shared-pair stream (two entries per LDS row read)."""
    if plan is None or plan.entries is None:
        return None
    meta = plan.entries[:, 0]
    real = int(((meta & (1 << 30)) == 0).sum())
    return round(int((meta < 0).sum()) / max(real, 1), 4)


def popularity_note():
    """The non-default popularity law is part of the workload's name; SURVEY 8d's law (the default) is not spelled out."""
    p = os.environ.get("WGNN_SYNTH_POPULARITY", "testis199")
    return "" if p == "testis199" else f" ({p} popularity - NOT SURVEY 8d's generator)"


KERNEL_SOURCES = ("wgnn_tiled.hip", "gen_flat_asm.py", "wgnn_flat_asm.inc", "wgnn_common.h", "wgnn_kernels.hip")


def kernel_sources_sha():
    """Content hash of the sources of the aggregation kernels: profiles/hbm_traffic.json is stamped with it at capture time
    (scratch/profile_round.sh) and `roofline.traffic` is null when the tree's kernels are no longer the captured ones."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update((ROOT / "scdeepsort_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()


def workload_string(cfg, total_cells, mode):
    """`config.workload` of the line.  A pure function of the config and the scaling mode - NOT of the number of ranks - so that
    the N = 1 leg of a SCALE run names the same workload as the BENCH line (tests/test_host_logic.py pins it)."""
    return (f"{cfg.name}: {total_cells} cells x {cfg.genes} genes"
            f"{' (ONE job, cells sharded over the ranks)' if mode == 'strong' else ' in total (' + str(cfg.cells) + ' per GPU)'}, "
            f"density {cfg.density}{popularity_note()}, dense_dim {cfg.dense_dim}, hidden {cfg.hidden}, "
            f"{cfg.n_layers}-layer WGNN forward + {cfg.n_classes}-class head")


def device_identity(dev):
    """Hardware identity of this rank's device, so that a multi-rank record can be read for "did the N ranks sit on N
    devices": PCI domain:bus:device and UUID from the runtime's device properties (rocm-smi as a fallback for the bus id)."""
    p = torch.cuda.get_device_properties(dev)
    ident = {"name": p.name, "arch": getattr(p, "gcnArchName", None), "cus": p.multi_processor_count}
    try:
        ident["pci"] = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    except Exception:
        ident["pci"] = None
    try:
        ident["uuid"] = str(p.uuid)
    except Exception:
        ident["uuid"] = None
    if ident["pci"] is None and ident["uuid"] is None:
        try:
            import subprocess
            out = subprocess.run(["rocm-smi", "--showbus"], capture_output=True, text=True, timeout=20).stdout
            ident["rocm_smi_showbus"] = [l.strip() for l in out.splitlines() if "PCI Bus" in l][: torch.cuda.device_count()]
        except Exception:
            pass
    return ident


def one_shot_costs(cfg, model, dev, S, sda, steady_ms):
    """Secondary (never `value`): what ONE inference on a NEW graph costs - the reference's inference is one forward per built
    graph (predict.py:44-54,61-88), while `value` times the steady-state forward on a resident graph.  Every figure is wall
    time around device work with a synchronize on both sides, on operands generated outside the timed regions:
      graph_build_ms  : K4 normalisation of both directions + the gene-major transpose (CellGeneGraph.from_device_csr)
      plan_build_ms   : the tile plans of both directions at the forward's width
      first_forward_ms: the first forward on the new graph (plans already built; allocator warm)
      predict_end_to_end_ms: DeepSortPredictor-shaped - 10 000 support cells + 100 000 test cells x the config's genes
                        (test cells feed no genes, preprocess.py:184-187): build + plans + ONE forward of the test cells"""
    from scdeepsort_amd import ops
    from scdeepsort_amd.graph import CellGeneGraph
    G = cfg.genes

    def wall(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, r
    kb = ops.tiled_block_rows(-(-min(cfg.hidden, 256) // 4) * 4)
    rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, seed=S.REFERENCE_SEED + 17, device=dev)
    feats = S.synth_features(G + cfg.cells, cfg.dense_dim, seed=23, device=dev, dtype=cfg.feature_dtype)
    def plans(gr):
        return (gr.cg.tile_plan(kb), gr.gc.tile_plan(kb)) if ops.tiled_kernel_serves(gr.cg, cfg.hidden) else None
    # every build twice: the first meets the caching allocator as the benchmark left it (blocks of other sizes: new hipMallocs,
    # tens of ms), the second finds its blocks cached - the device work itself (`*_ms`); a fresh process pays the first kind
    t_graph0, g = wall(lambda: CellGeneGraph.from_device_csr(rp, col, val, G))
    t_plan0, _ = wall(lambda: plans(g))
    del g
    t_graph, g = wall(lambda: CellGeneGraph.from_device_csr(rp, col, val, G))
    t_plan, _ = wall(lambda: plans(g))
    with torch.no_grad():
        t_first, _ = wall(lambda: model(g, feats))
        t_second, _ = wall(lambda: model(g, feats))
    rec = {"graph_build_ms": round(t_graph, 2), "plan_build_ms": round(t_plan, 2),
           "graph_build_first_ms": round(t_graph0, 2), "plan_build_first_ms": round(t_plan0, 2),
           "first_forward_ms": round(t_first, 2), "second_forward_ms": round(t_second, 2),
           "forwards_that_amortise_the_plans": round(t_plan / max(steady_ms, 1e-6), 1),
           "graph": f"{cfg.cells} cells x {G} genes, a NEW graph (seed + 17); *_first_ms = the first build in this allocator state"}
    del g, rp, col, val, feats
    if not cfg.total_cells and cfg.cells >= 50_000:
        n_sup, n_test = 10_000, cfg.cells
        rp, col, val = S.synth_expression(n_sup + n_test, G, cfg.density, seed=S.REFERENCE_SEED + 29, device=dev)
        feats = S.synth_features(G + n_sup + n_test, cfg.dense_dim, seed=31, device=dev, dtype=cfg.feature_dtype)
        mask = torch.zeros(n_sup + n_test, dtype=torch.bool, device=dev); mask[:n_sup] = True
        seeds = range(G + n_sup, G + n_sup + n_test)

        def predict():
            gp = CellGeneGraph.from_device_csr(rp, col, val, G, support_mask=mask)
            with torch.no_grad():
                return model(gp, feats, seeds=seeds)
        t_e2e0, out = wall(predict)
        assert out.shape[0] == n_test and torch.isfinite(out).all()
        del out
        t_e2e, out = wall(predict)
        rec["predict_end_to_end_ms"] = round(t_e2e, 2)
        rec["predict_end_to_end_first_ms"] = round(t_e2e0, 2)
        rec["predict_graph"] = (f"{n_sup} support + {n_test} test cells x {G} genes, {cfg.n_layers} layers: graph build + tile plans + ONE "
                                "forward of the test cells (api._predict_logits' device side; CSV ingest and PCA are host work outside it)")
    return rec


def self_launch(args):
    """``python bench.py --gpus N`` (N > 1) outside a launcher: start N ranks of this script under
    torch.distributed.run, one per GPU.  A box with fewer than N GPUs fails loudly - unless WGNN_BENCH_SHARE_GPU=1
    asks for the debug mode in which all ranks share cuda:0 (then the backend defaults to gloo: RCCL refuses two
    ranks on one device)."""
    import socket
    n_dev = torch.cuda.device_count()
    share = os.environ.get("WGNN_BENCH_SHARE_GPU") == "1"
    if n_dev < args.gpus and not share:
        sys.exit(f"bench.py: --gpus {args.gpus} requested but this box exposes {n_dev} GPU(s); "
                 f"refusing to silently run fewer ranks (WGNN_BENCH_SHARE_GPU=1 runs all ranks on cuda:0 for debugging)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    os.execv(sys.executable, cmd)


def build_workload(cfg, mode, rank, world, dev, S, shard_range):
    """This rank's operand.  ``strong``: ONE job of cfg.cells cells - every rank generates the SAME graph / features from
    the reference seed and keeps its ``shard_range`` of the cell axis (BASELINE cfg4: "Same 100k x 20k graph ... cells
    sharded 8-way"); also returns the whole job (for the sharded == unsharded self-check).  ``weak``: every rank owns its own
    cfg.cells-cell shard (seed + rank)."""
    G = cfg.genes
    feats_g = S.synth_features(G, cfg.dense_dim, seed=7, device=dev, dtype=cfg.feature_dtype)
    if mode == "weak":
        rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, seed=S.REFERENCE_SEED + rank, device=dev)
        feats_c = S.synth_features(cfg.cells, cfg.dense_dim, seed=100 + rank, device=dev, dtype=cfg.feature_dtype)
        return (rp, col, val), feats_g, feats_c, cfg.cells * world, None
    feats_c = S.synth_features(cfg.cells, cfg.dense_dim, seed=100, device=dev, dtype=cfg.feature_dtype)
    if world == 1:
        rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, seed=S.REFERENCE_SEED, device=dev)
        return (rp, col, val), feats_g, feats_c, cfg.cells, None
    lo, hi = shard_range(cfg.cells, rank, world)
    if rank != 0:
        # the rows lo .. hi-1 of the SAME matrix: only the chunks of the generator that meet the shard are drawn (the host's cores
        # are shared by the node's ranks; rank 0 draws the whole job once more for the self-check below)
        mine = S.synth_expression(cfg.cells, G, cfg.density, seed=S.REFERENCE_SEED, device=dev, cell_range=(lo, hi))
        return mine, feats_g, feats_c[lo:hi].clone(), cfg.cells, None
    rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, seed=S.REFERENCE_SEED, device=dev)
    b, e = int(rp[lo]), int(rp[hi])
    mine = ((rp[lo:hi + 1] - rp[lo]).clone(), col[b:e].clone(), val[b:e].clone())
    whole = (rp, col, val, feats_c) if rank == 0 else None
    return mine, feats_g, feats_c[lo:hi].clone(), cfg.cells, whole


def timed_steps(engine, feats_g, feats_c, steps, warmup, world, dev, profile=True, step=None):
    """W untimed + EXACTLY K timed forwards between barrier + synchronize on both sides; returns (max-over-ranks seconds,
    local seconds, per-launch HIP-event records, last output).  ``step``: replaces the eager forward (hipGraph replay)."""
    from scdeepsort_amd import ops

    def eager():
        with torch.no_grad():
            return engine.forward(feats_g, feats_c, async_gather=world > 1)    # concat of step i overlaps step i+1
    step = step or eager

    out = None
    for _ in range(warmup):
        out = step()
    if world > 1:
        engine.wait_gather()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ops.PROFILE = [] if profile else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    if world > 1:
        engine.wait_gather()                        # the last step's concat is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    dt_local = dt
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, dt_local, prof or [], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.environ.get("WGNN_BENCH_CONFIG", "cfg3"))
    ap.add_argument("--scaling", choices=("strong", "weak"), default=os.environ.get("WGNN_BENCH_SCALING", "strong"),
                    help="strong (default): ONE cfg-sized job, its cells sharded over the ranks (BASELINE cfg3/cfg4/cfg5); "
                         "weak: every rank owns a cfg-sized shard")
    ap.add_argument("--graphed", choices=("auto", "on", "off"), default="auto",
                    help="replay the forward as ONE hipGraph launch per step (auto: launch-bound configs, i.e. small graphs at N = 1; "
                         "at N > 1 only with `on`: the captured sharded forward is then timed against eager issue and the faster one runs)")
    ap.add_argument("--hidden", type=int, default=None, help="override the config's hidden width (e.g. 200, the reference's default "
                                                             "hidden_dim, train.py:137) - a side measurement, not BASELINE's line")
    ap.add_argument("--popularity", choices=("testis199", "dense_head"), default=os.environ.get("WGNN_SYNTH_POPULARITY", "testis199"),
                    help="gene popularity law of the synthetic graph: testis199 = SURVEY 8d's (the demo file's inclusion curve, numpy "
                         "default_rng), dense_head = rounds 1-5's generator (rank^-0.9 taken literally, denser head) for A/B")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the weak-scaling / sustained secondary measurements")
    args = ap.parse_args()

    os.environ["WGNN_SYNTH_POPULARITY"] = args.popularity       # every synth_expression of this run (and of its child ranks)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                                        # does not return
    rank = int(os.environ.get("RANK", 0)); local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if os.environ.get("WGNN_BENCH_DUMP_AFTER"):                  # debugging aid: every rank prints its Python stack after N s
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["WGNN_BENCH_DUMP_AFTER"]), repeat=False, exit=False)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    share = os.environ.get("WGNN_BENCH_SHARE_GPU") == "1"        # debug: the N>1 path on a 1-GPU box (all ranks on cuda:0)
    backend = os.environ.get("WGNN_BENCH_BACKEND", "gloo" if share else "nccl")
    if share:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local_rank}, {torch.cuda.device_count()} visible)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            from scdeepsort_amd import dist as wdist
            wdist.reserve_comm_cus()                     # (the RCCL channel cap is opt-in: WGNN_CAP_NCCL_CHANNELS=1, see dist.py)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    try:                                             # the dense projections are plain library GEMMs: hipBLASLt's fp32 kernels
        torch.backends.cuda.preferred_blas_library("hipblaslt")      # ran ~10 % faster than the default choice on these shapes
    except Exception:                                                # (profiles/r02_issue_analysis.md)
        pass

    import scdeepsort_amd as sda
    from scdeepsort_amd import synthetic as S, tuning
    from scdeepsort_amd.sharded import ShardedWgnn

    # per-shape pick among the libraries' own GEMM kernels for the dense projections (selection only, tracked file;
    # ignored on a different library stack - see scdeepsort_amd/tuning.py)
    gemm_selection = "library heuristics"
    if os.environ.get("WGNN_BENCH_TUNED_GEMMS", "1") == "1" and tuning.use_tuned_gemms():
        gemm_selection = f"TunableOp picks from scdeepsort_amd/{tuning.TUNED_FILE.name} (selection only, no tuning at run time)"

    cfg = S.CONFIGS[args.config]
    if args.hidden:
        import dataclasses
        cfg = dataclasses.replace(cfg, hidden=int(args.hidden), name=f"{cfg.name} with hidden {int(args.hidden)}")
    G = cfg.genes
    mode = args.scaling
    t_setup = time.time()
    (rp, col, val), feats_g, feats_c, total_cells, whole = build_workload(cfg, mode, rank, world, dev, S, sda.dist.shard_range)
    torch.cuda.synchronize()
    t_generate = time.time() - t_setup
    C = feats_c.shape[0]                             # cells held by THIS rank
    generator_desc = describe_workload(col, C, G, args.popularity, S)
    torch.manual_seed(1234)
    model = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, cfg.n_layers, G, activation=F.relu)
    with torch.no_grad():
        model.alpha.uniform_(0.5, 1.5)               # reference init is ones; exercise the alpha path
    model = model.to(dev).eval()
    engine = ShardedWgnn.build(model, rp, col, val, G)          # world == 1 -> plain single-GPU graph
    del rp, col, val
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    # ---- N > 1: which device does every rank drive?  (a SCALE record must show that RCCL saw N devices)
    ident = device_identity(dev)
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        keys = [i.get("uuid") or i.get("pci") for i in idents]
        if not share and all(k is not None for k in keys) and len(set(keys)) != world:
            sys.exit(f"bench.py: {world} ranks but only {len(set(keys))} distinct device(s): {keys} "
                     "(WGNN_BENCH_SHARE_GPU=1 is the debug mode that shares one)")

    # launch-bound configs (cfg2: ~25 launches of a few us each): the same forward captured once and replayed as one hipGraph
    # launch per step (scdeepsort_amd.graphed.GraphedForward); per-launch HIP events cannot be recorded inside a replay, so
    # the roofline's per-kernel durations come from a short eager pass of the same forward outside the timed region
    # (graphed.GraphedShardedForward replays the whole sharded forward INCLUDING the RCCL collectives as one hipGraph; with a
    # host-side backend such as gloo: graph segments around eager collectives)
    # N > 1: "auto" issues the sharded forward EAGERLY - measured faster than the captured graph on the 1-GPU lease (a rank's shard
    # is GPU-bound: profiles/r04_shard_trace.json), and a capture that includes RCCL collectives across real peers cannot be
    # exercised on that lease; `--graphed on` captures it (GraphedShardedForward), times both and runs the faster one.
    # N = 1: "auto" replays the launch-bound small graphs only.  The big ones are issued eagerly so that the roofline's HIP events sit
    # INSIDE the timed region, around the very launches that are timed; `--graphed on` replays them too (cfg3: 3.453 / 3.459 ms
    # replayed vs 3.504 / 3.519 ms eager on one box, 3.599 vs 3.591 on another; cfg5 equal) and then takes the per-kernel durations
    # from a separate eager pass, reported as `config.eager_ms_per_step`.
    graphed = (world == 1 and (args.graphed == "on" or (args.graphed == "auto" and cfg.cells * cfg.genes <= 100_000_000))) \
        or (world > 1 and args.graphed == "on")
    step_fn, launch_desc, launch_calibration = None, "eager", None
    if graphed and world == 1:
        from scdeepsort_amd.graphed import GraphedForward
        gf = GraphedForward(model, engine.graph, torch.cat([feats_g, feats_c]))
        step_fn, launch_desc = (lambda: gf()), "hipGraph replay (1 launch per step)"
    elif graphed:
        from scdeepsort_amd.graphed import GraphedShardedForward
        gf = GraphedShardedForward(engine, feats_g, feats_c)
        step_fn = lambda: gf()
        launch_desc = ("hipGraph replay (1 launch per step, RCCL collectives captured)" if gf.mode == "whole" else
                       f"hipGraph replay in {gf.n_graphs} segments around {gf.n_eager_collectives} host-side collectives ({backend})")
        if True:
            # Measured on the 1-GPU lease (profiles/r04_shard_trace.json): one rank's shard at N = 4 / 8 is GPU-bound, not
            # launch-bound - its ~15 kernels add up to the forward's wall time, the host runs ahead - and a graph replay costs
            # its fixed launch overhead on top (0.596 eager vs 0.637 ms replayed at 12.5k cells).  So "auto" times a few steps
            # of both, every rank takes the job-wide faster one (max over ranks, same decision everywhere), both are reported.
            cal = {}
            for name, fn in (("graphed", step_fn), ("eager", None)):
                cal[name] = timed_steps(engine, feats_g, feats_c, max(3, min(10, args.steps)), 2, world, dev, profile=False, step=fn)[0]
            launch_calibration = {k: round(v / max(3, min(10, args.steps)) * 1e3, 4) for k, v in cal.items()}
            if cal["eager"] <= cal["graphed"]:
                step_fn, graphed = None, False
                launch_desc = f"eager (measured faster than the captured graph: {launch_calibration} ms per step)"
            else:
                launch_desc += f" (measured faster than eager: {launch_calibration} ms per step)"
    # ---- N > 1: the tile geometry of the pass that overlaps the [G, H] all-reduce - planned for 256 - WGNN_COMM_CUS CUs (default
    # 32, calibrated on a spin-kernel stand-in: profiles/r04_comm_contention.json) or for the whole chip?  Both are timed next to
    # the REAL communicator, every rank keeps the job-wide faster one (max over ranks), both are reported.
    comm_cus_calibration = None
    if world > 1 and not graphed and engine.overlap_cu_budget != 256 and os.environ.get("WGNN_BENCH_COMM_AB", "1") == "1":
        n_cal = max(3, min(10, args.steps))
        budgets = {"reserved": engine.overlap_cu_budget, "full_chip": 256}
        cal = {}
        for name, bud in budgets.items():
            engine.overlap_cu_budget = bud
            cal[name] = timed_steps(engine, feats_g, feats_c, n_cal, 2, world, dev, profile=False)[0] / n_cal * 1e3
        keep = min(cal, key=cal.get)
        engine.overlap_cu_budget = budgets[keep]
        comm_cus_calibration = {"ms_per_step": {k: round(v, 4) for k, v in cal.items()}, "kept": keep,
                                "cus_planned_for_the_overlapped_pass": budgets[keep], "WGNN_COMM_CUS": sda.dist.COMM_CUS}
    dt, dt_local, prof, out = timed_steps(engine, feats_g, feats_c, args.steps, args.warmup, world, dev, step=step_fn)
    assert torch.isfinite(out).all()
    eager_ms = None
    if graphed:
        dt_e, _, prof, _ = timed_steps(engine, feats_g, feats_c, args.steps, 1, world, dev)
        eager_ms = round(dt_e / args.steps * 1e3, 4)
    ms_per_step = dt / args.steps * 1e3
    value = total_cells / (dt / args.steps)

    # ---- N > 1, strong: the sharded job's logits against the SAME graph evaluated unsharded on rank 0's GPU (outside the
    # timed region).  Skipped for jobs too large to repeat on one GPU within the run's time budget.
    self_check = None
    if world > 1 and mode == "strong" and whole is not None and total_cells <= 200_000:
        w_rp, w_col, w_val, w_fc = whole
        from scdeepsort_amd.graph import CellGeneGraph
        g1 = CellGeneGraph.from_device_csr(w_rp, w_col, w_val, G)
        with torch.no_grad():
            ref = model.linear(model.embed(g1, (feats_g, w_fc)))
        self_check = {"max_abs_sharded_minus_unsharded": float((out.float() - ref.float()).abs().max()),
                      "rows_compared": int(ref.shape[0]), "tolerance": 1e-4}
        assert self_check["max_abs_sharded_minus_unsharded"] < 1e-4, self_check
        del g1, ref
    whole = None

    # ---- roofline of the dominant kernel (HIP events on the launch stream, averaged over the timed steps)
    per = {}
    for tag, e0, e1 in prof:
        per.setdefault(tag, []).append(e0.elapsed_time(e1))
    passes = []
    for tag, ts in per.items():
        d = dict(zip(tag[::2], tag[1::2]))
        ms = sum(ts) / len(ts)
        b = pass_bytes(d["nnz"], d["rows"], d["cols"], d["D"], G)
        passes.append({"kernel": d["kernel"], "rows": d["rows"], "src_rows": d["cols"], "nnz": d["nnz"], "D": d["D"],
                       "launches_per_step": len(ts) // args.steps, "avg_ms": round(ms, 4),
                       "alg_bytes": b, "achieved_GBs": round(b / ms / 1e6, 1)})
    passes.sort(key=lambda p: -p["avg_ms"] * p["launches_per_step"])
    dom = passes[0]
    # HBM bytes per launch of the dominant kernel from the PMC counters.  PMC collection needs rocprofv3 around the whole
    # process (separate --pmc passes, scratch/profile_round.sh), so it cannot be taken inside this run: the number is
    # read from the tracked capture of the SAME kernel on the SAME workload (profiles/hbm_traffic.json names the kernel
    # and the round it was captured in) and is null when that capture does not match the kernel that ran here.
    traffic, traffic_src = None, None
    tf = ROOT / "profiles" / "hbm_traffic.json"
    if tf.exists():
        try:
            rec = json.loads(tf.read_text())
            ent = rec.get(f"{args.config}:{dom['rows']}x{dom['src_rows']}")
            fresh = rec.get("_kernel_sources_sha") == kernel_sources_sha() and rec.get("_popularity") == args.popularity
            if (isinstance(ent, dict) and ent.get("kernel") == dom["kernel"] and int(ent.get("D", 256)) == dom["D"]
                    and ent.get("nnz", dom["nnz"]) == dom["nnz"]):
                if fresh:
                    traffic = ent["hbm_bytes_per_launch"]
                    traffic_src = ("TRACKED capture, not a counter of this run (PMC needs rocprofv3 around the whole process): "
                                   f"profiles/hbm_traffic.json ({rec.get('_captured', '?')}; commit {rec.get('_commit', '?')}, "
                                   f"kernel sources {rec.get('_kernel_sources_sha', '?')[:12]} == this tree's)")
                else:
                    traffic_src = ("null: profiles/hbm_traffic.json was captured from other kernel sources or another workload "
                                   f"(capture {str(rec.get('_kernel_sources_sha'))[:12]} / {rec.get('_popularity')}, this tree "
                                   f"{kernel_sources_sha()[:12]} / {args.popularity}) - re-run scratch/profile_round.sh")
        except Exception:
            traffic = None
    # measured device copy bandwidth (read + write of a 1 GiB fp32 buffer), outside the timed region: the achievable HBM rate
    src_buf = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev); dst_buf = torch.empty_like(src_buf)
    dst_buf.copy_(src_buf); torch.cuda.synchronize()
    ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ce0.record()
    for _ in range(5):
        dst_buf.copy_(src_buf)
    ce1.record(); torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src_buf.numel() * 4 / (ce0.elapsed_time(ce1) * 1e-3) / 1e9
    del src_buf, dst_buf
    fwd_bytes = engine.forward_alg_bytes(cfg.dense_dim, feats_g.element_size())
    agg_ms = sum(p["avg_ms"] * p["launches_per_step"] for p in passes)
    roofline = {"bound": "hbm", "kernel": f"{dom['kernel']} (rows={dom['rows']}, src={dom['src_rows']}, D={dom['D']})",
                "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dom["achieved_GBs"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "traffic_note": "FETCH_SIZE counts Infinity-Cache hits (MI355X_MICROARCH.md): the excess over the algorithmic "
                                "bytes is the per-XCD re-fetch of the source table, part of which never reaches HBM",
                "measured_copy_GBs": round(copy_gbs, 1), "frac_of_measured_copy": round(dom["achieved_GBs"] / copy_gbs, 4),
                "alg_bytes_per_launch": dom["alg_bytes"], "avg_launch_ms": dom["avg_ms"], "passes": passes,
                "agg_kernels_ms_per_step": round(agg_ms, 4), "outside_agg_kernels_ms_per_step": round(dt_local / args.steps * 1e3 - agg_ms, 4),
                "forward_alg_bytes": fwd_bytes, "forward_achieved_GBs": round(fwd_bytes / (dt_local / args.steps * 1e3) / 1e6, 1),
                # the resource that actually bounds the tile kernel: per launch every non-zero reads one D*4-byte source row
                # from LDS and every tile streams its source range into LDS once
                "on_chip": {"lds_read_bytes": dom["nnz"] * dom["D"] * 4,
                            "lds_achieved_TBs": round(dom["nnz"] * dom["D"] * 4 / dom["avg_ms"] / 1e9, 1),
                            "lds_peak_TBs": 157.0, "lds_frac": round(dom["nnz"] * dom["D"] * 4 / dom["avg_ms"] / 1e9 / 157.0, 3),
                            "fp32_fma_TFLOPs": round(2 * dom["nnz"] * dom["D"] / dom["avg_ms"] / 1e9, 1), "fp32_vector_peak_TFLOPs": 157.3},
                "note": "AI vs algorithmic bytes is 50-110 flop/B (> fp32 ridge ~20): the gather of nnz*D*4 B "
                        "from L2/MALL and fp32 FMA issue bound this kernel before HBM does (DESIGN.md section 4)"}

    # ---- N > 1: every rank's own dominant-kernel roofline (HIP events on that rank's stream) + the communicator's view
    per_gpu, comm = None, None
    if world > 1:
        mine = {"rank": rank, "device": f"cuda:{local_rank}", "gpu": torch.cuda.get_device_name(dev), "pci": ident.get("pci"),
                "uuid": ident.get("uuid"), "cells": C,
                "nnz": engine.nnz, "kernel": dom["kernel"], "avg_launch_ms": dom["avg_ms"],
                "achieved_GBs": dom["achieved_GBs"], "frac": round(dom["achieved_GBs"] / HBM_PEAK_GBS, 4),
                "forward_alg_bytes": fwd_bytes, "forward_achieved_GBs": roofline["forward_achieved_GBs"],
                "forward_frac": round(roofline["forward_achieved_GBs"] / HBM_PEAK_GBS, 4),
                "local_ms_per_step": round(dt_local / args.steps * 1e3, 4),
                "passes": [{k: p[k] for k in ("kernel", "rows", "src_rows", "D", "launches_per_step", "avg_ms", "achieved_GBs")}
                           for p in passes]}
        per_gpu = [None] * world
        dist.all_gather_object(per_gpu, mine)
        comm = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "shared_device": share,
                "distinct_devices": len({p.get("uuid") or p.get("pci") or p["device"] for p in per_gpu}),
                "cus_left_to_the_communicator": 256 - engine.overlap_cu_budget, "comm_cus_calibration": comm_cus_calibration,
                "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
                "collectives_per_step": "1 all-reduce [G,H] (genes<-cells partial sums) + 1 all-gather of the logits"}
        roofline["per_gpu"] = per_gpu
        roofline["aggregate_peak_GBs"] = HBM_PEAK_GBS * (1 if share else world)
        roofline["job_forward_achieved_GBs"] = round(sum(p["forward_alg_bytes"] for p in per_gpu) / ms_per_step / 1e6, 1)
        roofline["job_forward_frac"] = round(roofline["job_forward_achieved_GBs"] / roofline["aggregate_peak_GBs"], 4)

    # ---- secondary measurements (never `value`): a sustained run of the same step (>= ~3 s of GPU time, so that an
    # outside sampler sees the device busy and the average is not 20 steps thin), and - N > 1 - the weak-scaling line
    sustained, weak = None, None
    if not args.no_secondary:
        n_sus = max(args.steps, min(20000, int(3.0 / max(dt / args.steps, 1e-5))))     # >= 3 s of the same step back to back
        dt_s, _, _, _ = timed_steps(engine, feats_g, feats_c, n_sus, 0, world, dev, profile=False, step=step_fn)
        sustained = {"steps": n_sus, "ms_per_step": round(dt_s / n_sus * 1e3, 4), "value": round(total_cells / (dt_s / n_sus), 1),
                     "unit": "cells/s"}
        if world > 1 and mode == "strong" and not cfg.total_cells:
            del engine
            (rp, col, val), wf_g, wf_c, w_total, _ = build_workload(cfg, "weak", rank, world, dev, S, sda.dist.shard_range)
            w_engine = ShardedWgnn.build(model, rp, col, val, G)
            del rp, col, val
            dt_w, _, _, _ = timed_steps(w_engine, wf_g, wf_c, args.steps, args.warmup, world, dev, profile=False)
            weak = {"scaling": "weak", "cells_total": w_total, "cells_per_gpu": cfg.cells, "steps": args.steps,
                    "ms_per_step": round(dt_w / args.steps * 1e3, 4), "value": round(w_total / (dt_w / args.steps), 1),
                    "unit": "cells/s"}
            engine = w_engine

    # ---- secondary (never `value`): BASELINE cfg4's training step on the same resident graph at N = 1 - forward (dropout 0.1) +
    # CrossEntropyLoss(sum) + backward through K2t / the fused glue + Adam (reference train.py:80-87)
    train_step = None
    if world == 1 and not args.no_secondary and not cfg.total_cells and os.environ.get("WGNN_BENCH_TRAIN_STEP", "1") == "1":
        torch.manual_seed(4321)
        tm = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, cfg.n_layers, G, activation=F.relu, dropout=0.1).to(dev)
        topt = torch.optim.Adam(tm.parameters(), lr=1e-3, weight_decay=5e-4, fused=True)
        ty = (torch.arange(C, device=dev) * 2654435761 % cfg.n_classes).long()
        tf = (feats_g.float(), feats_c.float())

        def tstep():
            loss = sda.cross_entropy_sum(tm(engine.graph, tf), ty)
            topt.zero_grad(set_to_none=True); loss.backward(); topt.step()
            return loss
        for _ in range(3):
            tl = tstep()
        torch.cuda.synchronize()
        n_tr = 10
        t0 = time.perf_counter()
        for _ in range(n_tr):
            tl = tstep()
        torch.cuda.synchronize()
        t_train = (time.perf_counter() - t0) / n_tr * 1e3
        # per-kernel roofline entries of the step (HIP events on the launch stream, a separate profiled pass of 3 steps):
        # the aggregation passes against the HBM roofline (the same algorithmic-bytes formula on the operand each launch walks),
        # the matrix-core weight gradients against the fp32 matrix peak
        from scdeepsort_amd import ops as _ops
        _ops.PROFILE = []
        for _ in range(3):
            tstep()
        torch.cuda.synchronize()
        tprof, _ops.PROFILE = _ops.PROFILE, None
        tper = {}
        for tag, e0, e1 in tprof:
            tper.setdefault(tag, []).append(e0.elapsed_time(e1))
        tk = []
        for tag, ts in tper.items():
            d = dict(zip(tag[::2], tag[1::2]))
            ms = sum(ts) / len(ts)
            if "nnz" in d:
                b = pass_bytes(d["nnz"], d["rows"], d["cols"], d["D"], G)
                tk.append({"kernel": d["kernel"], "rows": d["rows"], "src_rows": d["cols"], "D": d["D"], "launches_per_step": len(ts) // 3,
                           "avg_ms": round(ms, 4), "bound": "hbm", "alg_bytes": b, "achieved_GBs": round(b / ms / 1e6, 1),
                           "frac": round(b / ms / 1e6 / HBM_PEAK_GBS, 4)})
            else:
                fl = 2.0 * d["rows"] * d["N"] * d["K"]
                tk.append({"kernel": d["kernel"], "M": d["rows"], "N": d["N"], "K": d["K"], "launches_per_step": len(ts) // 3,
                           "avg_ms": round(ms, 4), "bound": "mfma", "flops": fl, "achieved_TFLOPs": round(fl / ms / 1e9, 1),
                           "peak_TFLOPs": 157.3, "frac": round(fl / ms / 1e9 / 157.3, 4)})
        tk.sort(key=lambda k: -k["avg_ms"] * k["launches_per_step"])
        train_step = {"ms_per_step": round(t_train, 4), "steps": n_tr, "loss": round(float(tl), 2),
                      "what": f"full-batch {cfg.name} training step at N = 1: forward (dropout 0.1) + CrossEntropyLoss(sum) + backward "
                              "(K2t on pre-scaled rows, wgnn_agg_bwd_prepare, matrix-core weight gradients) + fused Adam",
                      "kernels": tk,
                      "kernels_ms_per_step": round(sum(k["avg_ms"] * k["launches_per_step"] for k in tk), 4)}
        del tm, topt, tf

    # ---- secondary (never `value`): the one-shot path - graph build, plan build, first forward, predictor-shaped end to end
    one_shot = None
    if world == 1 and not args.no_secondary and os.environ.get("WGNN_BENCH_ONE_SHOT", "1") == "1":
        one_shot = one_shot_costs(cfg, model, dev, S, sda, ms_per_step)
        one_shot["bench_workload_generate_s"] = round(t_generate, 2)

    # ---- CPU baseline: the restatement (C/OpenMP aggregation + torch Linear) on this host, rank 0, N = 1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(engine.graph, model, feats_g, feats_c, out, cfg)

    try:
        from scdeepsort_amd import ops as _ops
        _kb = _ops.tiled_block_rows(-(-min(cfg.hidden, 256) // 4) * 4)
        if _ops.tiled_kernel_serves(engine.graph.cg, cfg.hidden):
            generator_desc["shared_pair_share"] = {"cells<-genes": shared_pair_share(engine.graph.cg.tile_plan(_kb)),
                                                   "genes<-cells": shared_pair_share(engine.graph.gc.tile_plan(_kb))}
    except Exception as e:                                       # descriptive only
        generator_desc["shared_pair_share"] = f"unavailable: {e}"
    if rank == 0:
        line = {"metric": "cells embedded/sec (2-layer WGNN fwd)", "value": round(value, 1), "unit": "cells/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
                "higher_is_better": True, "scaling": mode, "vs_baseline": None,
                "dtype": "f32" if cfg.feature_dtype == torch.float32 else "f32 (fp16-stored input features)", "data": "synthetic",
                "config": {"workload": workload_string(cfg, total_cells, mode),
                           "generator": generator_desc, "cells_total": total_cells, "cells_this_rank": C,
                           "nnz_per_gpu": per_gpu[0]["nnz"] if per_gpu else roofline["passes"][0]["nnz"],
                           "parallelism": f"cell-shard x{world}", "setup_s": round(t_setup, 1), "device": ident, "communicator": comm,
                           "gemm_selection": gemm_selection, "step_launch": launch_desc, "eager_ms_per_step": eager_ms, "launch_calibration_ms": launch_calibration,
                           "sharded_vs_unsharded": self_check},
                "roofline": roofline, "cpu_baseline": cpu, "sustained": sustained, "weak_scaling": weak, "train_step": train_step,
                "one_shot": one_shot}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
