#!/usr/bin/env python
"""BASELINE cfg4: data-parallel training of the 2-layer WGNN over cell shards (one process per GPU, RCCL).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_sharded.py

Every rank owns a contiguous range of the cells (its rows of the cells<-genes CSR, its columns of the genes<-cells
CSR), the gene table and the parameters are replicated.  Per step: local forward, ONE all-reduce of the [G, H] gene
partial sums (and of its gradient in backward), CrossEntropyLoss(reduction='sum') on the local cells (train.py:36), SUM
all-reduce of the parameter gradients in one bucket, identical Adam step on every rank (train.py:34-35,80-87).
Synthetic data of the cfg3/cfg4 shape; WGNN_BACKEND=gloo + WGNN_SHARE_GPU=1 runs all ranks on one GPU (debug)."""
import argparse
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=0, help="global seed-batch size (0 = full batch); every rank "
                    "takes batch/world seeds of its own shard per step (train.py:71-87, data-parallel)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = 0 if os.environ.get("WGNN_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("WGNN_BACKEND", "nccl")
        kw = dict(device_id=torch.device("cuda", local)) if backend == "nccl" else {}
        if backend == "nccl":
            from scdeepsort_amd import dist as wdist
            wdist.reserve_comm_cus()
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import scdeepsort_amd as sda
    from scdeepsort_amd import synthetic as S, tuning
    tuning.use_tuned_gemms()                                          # tracked per-shape picks among the library GEMM kernels
    from scdeepsort_amd.dist import shard_range
    from scdeepsort_amd.sharded import ShardedWgnn

    cfg = S.CONFIGS[args.config]
    # BASELINE cfg4: "Same 100k x 20k graph ... cells sharded 8-way" - every rank generates the SAME graph from the reference
    # seed (1.3 s at cfg3) and keeps its contiguous range of the cell axis, so an N-rank run trains on exactly the N = 1 job
    lo, hi = shard_range(cfg.cells, rank, world)
    C, G = hi - lo, cfg.genes
    rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, seed=S.REFERENCE_SEED, device=dev)
    b, e = int(rp[lo]), int(rp[hi])
    rp, col, val = (rp[lo:hi + 1] - rp[lo]).clone(), col[b:e].clone(), val[b:e].clone()
    torch.manual_seed(1234)                                           # identical initial parameters on every rank
    model = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, cfg.n_layers, G, activation=F.relu, dropout=0.1).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4, fused=True)     # train.py:34-35 in one multi-tensor launch
    engine = ShardedWgnn.build(model, rp, col, val, G)
    feats_g = S.synth_features(G, cfg.dense_dim, seed=7, device=dev)
    feats_c = S.synth_features(cfg.cells, cfg.dense_dim, seed=100, device=dev)[lo:hi].clone()
    labels = (torch.arange(lo, hi, device=dev) * 2654435761 % cfg.n_classes).long()
    per_rank = args.batch_size // world if args.batch_size else 0
    gen = torch.Generator(device=dev).manual_seed(1 + rank)

    def one_step():                                                  # the loss stays on the device: no host read per step
        if not per_rank:
            return engine.train_step(feats_g, feats_c, labels, opt, sync_loss=False)
        sel = torch.randperm(C, device=dev, generator=gen)[:per_rank]             # this rank's seeds of the step
        return engine.train_step(feats_g, feats_c, labels[sel], opt, seeds_local=sel, sync_loss=False)
    loss = one_step()                                                # warm-up (plans, communicator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        n_seed = per_rank * world if per_rank else cfg.cells
        kind = f"mini-batch ({n_seed} seeds)" if per_rank else "full-batch"
        print(f"{args.config}: {cfg.cells} cells over {world} rank(s): {dt * 1e3:.2f} ms per {kind} training step "
              f"({n_seed / dt / 1e6:.2f} M cells/s), loss/cell {float(loss) / n_seed:.4f}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
