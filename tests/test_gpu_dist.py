"""World-size-2 GPU tests of the cell-shard path: TWO PROCESSES, a real process group, the HIP
operators in both, both ranks on cuda:0 (the round-end box has one GPU).  RCCL refuses two ranks on one device
("Duplicate GPU detected"), so the collectives of these tests go through gloo on CUDA tensors; the orchestration
(`ShardedWgnn` -> `dist.sharded_forward` / `sharded_train_step`), the kernels and the rank-invariant decisions are the
production ones.  `test_bench_two_ranks_on_one_gpu` drives `bench.py --gpus 2` end to end the same way.
The reference has no counterpart (single process, train.py:23): the oracle is the UNSHARDED forward / autograd."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

pytestmark = pytest.mark.gpu
BACKEND = os.environ.get("WGNN_TEST_DIST_BACKEND", "gloo")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, hidden, dropout, backend=BACKEND, force=False, n_layers=2):
    import torch.nn.functional as F
    import scdeepsort_amd as sda
    from oracle import wgnn_oracle as O
    from scdeepsort_amd import dist as D, synthetic as S
    from scdeepsort_amd.sharded import ShardedWgnn
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        D.reserve_comm_cus()                                        # default: no process-wide cap (opt-in since round 5)
        D.reserve_comm_cus(cap_channels=True)                       # the opt-in sequence: channel cap, then the communicator
        assert os.environ["NCCL_MAX_NCHANNELS"] == str(D.COMM_CUS)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    D.FORCE_COLLECTIVES = force                                 # a one-rank group still issues every collective
    try:
        G, C, Din, ncls = 300, 1001, 40, 5                      # ragged shards (501 + 500)
        rp, col, val = S.synth_expression(C, G, 0.08, seed=3, device=dev)
        torch.manual_seed(0)                                     # identical parameters on both ranks
        m = sda.GNN(Din, hidden, ncls, n_layers, G, activation=F.relu, dropout=dropout).to(dev)
        with torch.no_grad():
            m.alpha.uniform_(0.5, 1.5)
        # (the torch-generator features this test's tolerances were calibrated on: a gradient check against an fp64 oracle is
        #  sensitive to pre-activations that round to different sides of the ReLU, i.e. to the particular inputs)
        feats = S.synth_features(G + C, Din, device=dev, rng="torch")
        lo, hi = D.shard_range(C, rank, world)
        b, e = int(rp[lo]), int(rp[hi])
        eng = ShardedWgnn.build(m, (rp[lo:hi + 1] - rp[lo]).clone(), col[b:e].clone(), val[b:e].clone(), G, seed=11)
        assert eng.world == 2 and eng.shard_sizes == [D.shard_range(C, r, world)[1] - D.shard_range(C, r, world)[0]
                                                      for r in range(world)]
        # ---- inference: concat of both ranks' logits == the unsharded oracle on the whole graph
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        expr = S.to_scipy(rp, col, val, G)
        want = O.csr_forward(sd, O.build_csr_graph(expr), feats.cpu().numpy(), n_layers)
        m.eval()
        with torch.no_grad():
            got = eng.forward(feats[:G], feats[G + lo:G + hi])
            got2 = eng.forward(feats[:G], feats[G + lo:G + hi], async_gather=True)
            eng.wait_gather()
        assert got.shape == (C, ncls)
        err = float(np.abs(got.cpu().numpy() - want).max())
        assert err < 1e-4, err
        assert torch.equal(got, got2)
        # ---- the same forward replayed as a hipGraph (round 4): RCCL collectives captured with the kernels ("whole"), or -
        # with a host-side backend - graph segments around eager collectives; replay == eager bit for bit, also on new inputs
        from scdeepsort_amd.graphed import GraphedShardedForward
        with torch.no_grad():
            gsf = GraphedShardedForward(eng, feats[:G], feats[G + lo:G + hi])
            assert gsf.mode == ("whole" if backend == "nccl" else "segments"), gsf.mode
            # one [G, H] all-reduce per layer below the last + the logits concat
            assert gsf.n_graphs == (1 if backend == "nccl" else n_layers + 1)
            assert gsf.n_eager_collectives == (0 if backend == "nccl" else n_layers)
            for _ in range(2):
                assert torch.equal(gsf(), got)
            f2 = feats * 0.5 + 0.01
            want2 = eng.forward(f2[:G], f2[G + lo:G + hi])
            assert torch.equal(gsf(f2[:G], f2[G + lo:G + hi]), want2) and not torch.equal(want2, got)
        graphed_mode = gsf.mode
        del gsf
        if n_layers != 2:                                       # (the training checks below are written for two layers)
            Path(out_dir, f"ok{rank}").write_text(json.dumps({"err": err, "backend": dist.get_backend(), "graphed": graphed_mode}))
            return
        # ---- training step (cfg4): loss and ALL-REDUCED gradients == single-process autograd of the oracle
        labels = (torch.arange(C, device=dev) * 7 % ncls).long()
        opt = torch.optim.SGD(m.parameters(), lr=0.0)           # lr 0: inspect the gradients after the step
        masks = None
        if dropout:                                             # fixed masks, the gene part identical on both ranks
            gm = torch.Generator().manual_seed(5)
            full = [(torch.rand(G + C, w, generator=gm) >= dropout).float() / (1 - dropout) for w in (Din, hidden)]
            Hp = eng._weights()[0][0].shape[0]
            masks = [(fm[:G].to(dev), fm[G + lo:G + hi].to(dev)) for fm in full]
            if Hp != hidden:                                    # carried zero-padded: the pad columns are 0 anyway
                masks[1] = tuple(F.pad(x, (0, Hp - hidden), value=1.0) for x in masks[1])
        D.TRACE = []
        total = eng.train_step(feats[:G], feats[G + lo:G + hi], labels[lo:hi], opt, dropout_masks=masks)
        trace, D.TRACE = D.TRACE, None
        # round 6: the differentiable [G, H] all-reduce is split around the rank's own cells<-genes pass, forward and backward
        # (issue, overlapped pass, wait); the parameter gradients live in ONE flat bucket that is all-reduced in place
        assert trace == ["ar_fwd_issue", "ar_fwd_wait", "ar_bwd_issue", "ar_bwd_wait"], trace
        bucket = opt._wgnn_grad_bucket
        assert all(p_.grad.data_ptr() == v.data_ptr() for p_, v in zip(bucket.params, bucket.views))
        rg = O.build_reference_graph(expr)
        p64 = {k: v.detach().clone().double().requires_grad_(True) for k, v in sd.items()}
        logits = O.nodeflow_forward(p64, rg, feats.cpu().double(), np.arange(G, G + C), 2,
                                    dropout_masks=[fm.double() for fm in full] if dropout else None)
        loss = F.cross_entropy(logits, labels.cpu(), reduction="sum")
        loss.backward()
        assert abs(total - float(loss)) < 2e-4 * max(1.0, abs(float(loss))), (total, float(loss))
        for k, p_ in m.named_parameters():
            np.testing.assert_allclose(p_.grad.cpu().numpy(), p64[k].grad.numpy(), atol=3e-4, rtol=2e-3, err_msg=k)
        # ---- a mini-batch step: every rank contributes the seeds of ITS shard (local indices); the summed loss and the
        # all-reduced gradients equal the single-process loss over the union of the seeds (train.py:71-87 data-parallel)
        if not dropout:
            sel_local = torch.arange(0, hi - lo, 7, device=dev)
            for p_ in m.parameters():
                p_.grad = None
            total_b = eng.train_step(feats[:G], feats[G + lo:G + hi], labels[lo:hi][sel_local], opt, seeds_local=sel_local)
            sel_all = np.concatenate([np.arange(0, D.shard_range(C, r, world)[1] - D.shard_range(C, r, world)[0], 7)
                                      + D.shard_range(C, r, world)[0] for r in range(world)])
            pb = {k: v.detach().clone().double().requires_grad_(True) for k, v in sd.items()}
            lg = O.nodeflow_forward(pb, rg, feats.cpu().double(), G + sel_all, 2)
            lb = F.cross_entropy(lg, labels.cpu()[sel_all], reduction="sum")
            lb.backward()
            assert abs(total_b - float(lb)) < 2e-4 * max(1.0, abs(float(lb))), (total_b, float(lb))
            for k, p_ in m.named_parameters():
                np.testing.assert_allclose(p_.grad.cpu().numpy(), pb[k].grad.numpy(), atol=3e-4, rtol=2e-3, err_msg=k)
        # ---- the engine's own dropout streams: same gene mask on both ranks, different cell masks
        if dropout:
            m.train()
            mk = eng.dropout_masks(feats[:G], feats[G + lo:G + hi])
            both = [torch.zeros_like(mk[0][0]) for _ in range(world)]
            dist.all_gather(both, mk[0][0])
            assert all(torch.equal(both[0], b_) for b_ in both)
            cm = [torch.zeros(500, Din, device=dev) for _ in range(world)]
            dist.all_gather(cm, mk[0][1][:500].contiguous())
            assert world == 1 or not torch.equal(cm[0], cm[1])
        Path(out_dir, f"ok{rank}").write_text(json.dumps({"err": err, "backend": dist.get_backend(), "graphed": graphed_mode}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("hidden,dropout", [(32, 0.0), (32, 0.25), (50, 0.0)])
def test_world2_hip_sharded_engine_matches_unsharded_oracle(tmp_path, hidden, dropout):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), hidden, dropout), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))


def test_world2_three_layer_segmented_replay_reduces_every_layer(tmp_path):
    """the replayed collectives of a segmented capture must each target THEIR layer's partial sums.  With three
    layers there are two [G, H] all-reduces; a thunk that closed over the loop variable would reduce the last layer's tensor
    twice at replay and leave the first layer's partial sums unreduced (invisible with two layers: one all-reduce)."""
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), 32, 0.0, BACKEND, False, 3), nprocs=2, join=True)
    recs = [json.loads(Path(tmp_path, f"ok{r}").read_text()) for r in range(2)]
    assert all(r["graphed"] == "segments" and r["err"] < 1e-4 for r in recs)


def test_world1_nccl_group_drives_the_sharded_branch(tmp_path):
    """RCCL on the 1-GPU lease: RCCL refuses two ranks on one device, but a ONE-rank `nccl` communicator
    is legal.  With `dist.FORCE_COLLECTIVES` the sharded branch issues every collective it would issue at N > 1 -
    `init_process_group("nccl", device_id=)`, the two [G] all-reduces of the gene-side normalisation, the size exchange,
    the async [G, H] all-reduce + stream-ordered `work.wait()` under the cells<-genes pass, `all_gather_into_tensor`
    (sync and left in flight), the differentiable all-reduce in both directions and the gradient bucket - so RCCL's
    library load and stream semantics run on hardware; results are checked against the unsharded oracle."""
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), 32, 0.0, "nccl", True), nprocs=1, join=True)
    rec = json.loads((tmp_path / "ok0").read_text())
    assert rec["backend"] == "nccl" and rec["err"] < 1e-4
    assert rec["graphed"] == "whole"          # kernels AND the RCCL all-reduce / all-gather in ONE captured graph, replay == eager


def test_world1_nccl_group_with_dropout(tmp_path):
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), 32, 0.25, "nccl", True), nprocs=1, join=True)
    assert json.loads((tmp_path / "ok0").read_text())["backend"] == "nccl"


def _pad_worker(rank, world, port, out_dir):
    """the zero-padding decision of a narrow hidden width must be rank-invariant.  Rank 0's shard is above the
    tile-kernel threshold, rank 1's below: both must carry 256 columns or the [G, Hp] all-reduce mismatches.  (Since
    round 2 narrow widths run natively and nothing is padded by default; the legacy route is switched on here.)"""
    import torch.nn.functional as F
    import scdeepsort_amd as sda
    from scdeepsort_amd import dist as D, ops, synthetic as S
    from scdeepsort_amd.sharded import ShardedWgnn
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(BACKEND, rank=rank, world_size=world)
    try:
        G, Din, H = 400, 24, 200
        C = 900 if rank == 0 else 300
        rp, col, val = S.synth_expression(C, G, 0.1, seed=20 + rank, device=dev)
        nnz = [torch.zeros(1, dtype=torch.long, device=dev) for _ in range(world)]
        dist.all_gather(nnz, torch.tensor([col.shape[0]], device=dev))
        n0, n1 = int(nnz[0]), int(nnz[1])
        ops.PAD_NARROW_TO_256 = True                             # the round-1 route, where the width depends on the kernel choice
        ops.TILED_MIN_WORK = (n0 + n1) // 2 * H                  # rank 0 above, rank 1 below
        assert n0 * H >= ops.TILED_MIN_WORK > n1 * H
        torch.manual_seed(0)
        m = sda.GNN(Din, H, 4, 2, G, activation=F.relu).to(dev).eval()
        eng = ShardedWgnn.build(m, rp, col, val, G)
        assert eng._weights()[0][0].shape[0] == 256              # both ranks
        fg = S.synth_features(G, Din, seed=1, device=dev); fc = S.synth_features(C, Din, seed=2 + rank, device=dev)
        with torch.no_grad():
            out = eng.forward(fg, fc)                            # would hang / raise on mismatched widths
        assert out.shape == (1200, 4) and torch.isfinite(out).all()
        Path(out_dir, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_world2_padding_decision_is_rank_invariant(tmp_path):
    mp.spawn(_pad_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` starts two ranks ITSELF (never silently one): on this 1-GPU box in the shared-device
    debug mode.  The line must say n_gpus = 2 and carry one roofline record per rank."""
    env = dict(os.environ, WGNN_BENCH_SHARE_GPU="1", WGNN_BENCH_CONFIG="cfg2")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--graphed", "on"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["communicator"]["ranks"] == 2
    # `--graphed on`: the sharded forward was captured (graph segments around the gloo collectives here) and timed against eager issue;
    # the line says which one the timed steps used and carries both calibration times
    assert set(line["config"]["launch_calibration_ms"]) == {"graphed", "eager"}
    assert line["config"]["step_launch"].startswith(("hipGraph replay", "eager (measured faster"))
    assert [p["rank"] for p in line["roofline"]["per_gpu"]] == [0, 1]
    # default = strong scaling: the SAME 10k-cell cfg2 graph split over the ranks, checked against its unsharded evaluation
    assert line["scaling"] == "strong" and line["config"]["cells_total"] == 10_000 and line["cpu_baseline"] is None
    assert sorted(p["cells"] for p in line["roofline"]["per_gpu"]) == [5000, 5000]
    chk = line["config"]["sharded_vs_unsharded"]
    assert chk["rows_compared"] == 10_000 and chk["max_abs_sharded_minus_unsharded"] < 1e-4
    # ... and the weak-scaling line (every rank its own cfg2-sized shard) as the secondary field
    assert line["weak_scaling"]["cells_total"] == 20_000 and line["weak_scaling"]["value"] > 0
    assert line["sustained"]["steps"] >= 3


def test_bench_strong_scaling_config_shards_the_job(tmp_path):
    """cfg5's mode (`total_cells`: ONE job of fixed size split over the ranks, fp16-stored features) at a small size:
    ragged shards (3001 cells over 2 ranks), `scaling: strong`, the per-rank records add up to the job."""
    env = dict(os.environ, WGNN_BENCH_SHARE_GPU="1", WGNN_BENCH_CONFIG="tiny_atlas")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["cells_total"] == 3001
    assert sorted(p["cells"] for p in line["roofline"]["per_gpu"]) == [1500, 1501]
    assert "fp16" in line["dtype"]


def test_bench_refuses_more_ranks_than_gpus():
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "WGNN_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
