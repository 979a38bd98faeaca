"""World-size-2 tests of the cell-shard orchestration (scdeepsort_amd/dist.py) on CPU with gloo.
The local arithmetic is injected (torch CPU math here, the HIP operators in production), so what is
tested is the N>1 logic itself: shard ranges, the gene-side global normalisation, the single [G,H]
all-reduce per forward, the logits all-gather and the SUM gradient all-reduce."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, cells=90):
    import scipy.sparse as sp
    from conftest import small_case
    from oracle import wgnn_oracle as O
    from scdeepsort_amd import dist as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D.FORCE_COLLECTIVES = world == 1          # a one-rank group still issues every collective (the GPU suite does this with nccl)
    try:
        c = small_case(cells=cells, genes=40, dim=12, hidden=8, n_classes=3, seed=21, test_cells=0)
        G, C = c["G"], c["C"]
        sd = O.init_params(12, 8, 3, 2, G, seed=3, dtype=torch.float64)
        expr = sp.csr_matrix(c["expr"]).astype(np.float64)
        lo, hi = D.shard_range(C, rank, world)
        loc = expr[lo:hi]
        # cells<-genes: per-cell normalisation is local by construction
        nnz_c = np.diff(loc.indptr); rs = np.asarray(loc.sum(1)).ravel()
        A_cg = loc.copy(); A_cg.data = np.repeat(nnz_c, nnz_c) * loc.data / np.repeat(rs, nnz_c)
        inv_c = torch.from_numpy(1.0 / (nnz_c + 1.0))
        # genes<-cells: locally normalised, then rescaled with the all-reduced statistics
        XT = sp.csr_matrix(loc.T); XT.sort_indices()
        deg_loc = torch.from_numpy(np.diff(XT.indptr).astype(np.float32))
        sum_loc = torch.from_numpy(np.asarray(XT.sum(1)).ravel())
        row_of = torch.from_numpy(np.repeat(np.arange(G), np.diff(XT.indptr)))
        w_loc = torch.from_numpy(np.where(np.repeat(sum_loc.numpy(), np.diff(XT.indptr)) > 0,
                                          np.repeat(deg_loc.numpy(), np.diff(XT.indptr)) * XT.data /
                                          np.repeat(np.maximum(sum_loc.numpy(), 1e-30), np.diff(XT.indptr)), 0.0))
        w_glob, inv_g = D.normalise_gene_side(deg_loc, sum_loc, w_loc, row_of)
        ref = O.build_csr_graph(c["expr"], dtype=np.float64)
        want = ref.A_gc[:, lo:hi].tocsr(); want.sort_indices()
        np.testing.assert_allclose(w_glob.numpy(), want.data, rtol=2e-6)
        np.testing.assert_allclose(inv_g.numpy(), 1.0 / ref.deg_g, rtol=1e-7)
        A_gc = sp.csr_matrix((w_glob.numpy(), XT.indices, XT.indptr), shape=XT.shape)

        def dense(m):
            return torch.from_numpy(m.toarray())
        Acg, Agc = dense(A_cg), dense(A_gc)
        alpha = sd["alpha"].reshape(-1)
        ops = D.LocalOps(
            cells_layer=lambda p_g, p_c, b, relu, rows=None, sc=False: torch.relu(((Acg @ (alpha[:G, None] * p_g)) + alpha[G + 1] * p_c) * inv_c[:, None] + b),
            genes_partial=lambda p_c: Agc @ p_c,
            genes_finish=lambda part, p_g, b, relu: torch.relu((alpha[:G, None] * part + alpha[G] * p_g) * inv_g[:, None].double() + b))
        weights = [(sd[f"layers.{i}.fc_neigh.weight"], sd[f"layers.{i}.fc_neigh.bias"]) for i in range(2)] + \
                  [(sd["linear.weight"], sd["linear.bias"])]
        feats = torch.from_numpy(c["feats"]).double()
        sizes = [D.shard_range(C, r, world)[1] - D.shard_range(C, r, world)[0] for r in range(world)]
        with pytest.raises(ValueError):                  # no per-forward size exchange: the sizes are a build-time fact
            D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops, 2, gather_logits=True)
        logits = D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops, 2, True, sizes)
        full = O.csr_forward(sd, ref, c["feats"].astype(np.float64), 2, dtype=np.float64)
        np.testing.assert_allclose(logits.numpy(), full, atol=1e-6)          # every rank holds ALL cells' logits, in order
        # shard sizes known at graph build: the concat needs no size exchange (equal shards -> one all_gather_into_tensor)
        l2 = D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops, 2, True, sizes)
        assert torch.equal(l2, logits)
        res = D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops, 2, True, sizes, async_gather=True)
        if res[1] is not None:
            res[1].wait()
        assert torch.equal(res[0], logits)
        # ---- round 4: the capture hook sees exactly the data-path collectives (all-reduce of the gene partial sums, logits concat),
        # runs them synchronously, and the result does not change (graphed.GraphedShardedForward cuts its hipGraph there)
        seen = []
        D.COLLECTIVE_HOOK = lambda fn: (seen.append(1), fn())[1]
        try:
            lh = D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops, 2, True, sizes)
        finally:
            D.COLLECTIVE_HOOK = None
        assert len(seen) == 2 and torch.equal(lh, logits)
        # ---- round 4: the last layer in the reference's literal order (aggregate, then Linear + ReLU: `cells_mean_linear`) with
        # the gene rows handed over alpha-folded by `genes_finish(scale_out=True)` - inference; and the same order in training
        flags = []
        def cml(h_g, h_c, W, b, relu, rows=None, sc=False, prescaled=False):
            flags.append(prescaled)
            src = h_g if prescaled else alpha[:G, None] * h_g
            z = ((Acg @ src) + alpha[G + 1] * h_c) * inv_c[:, None]
            y = z @ W.t() + b
            return torch.relu(y) if relu else y
        def gfin(part, p_g, b, relu, scale_out=False):
            y = torch.relu((alpha[:G, None] * part + alpha[G] * p_g) * inv_g[:, None].double() + b)
            return alpha[:G, None] * y if scale_out else y
        ops2 = D.LocalOps(ops.cells_layer, ops.genes_partial, gfin, cml, lambda width, n_seed: True)
        with torch.no_grad():
            l_fold = D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops2, 2, True, sizes)
        assert flags == [True]
        np.testing.assert_allclose(l_fold.numpy(), full, atol=1e-6)
        flags.clear()
        # ---- round 4: `overlapped` wraps exactly the cells<-genes pass that runs next to the in-flight all-reduce (no-grad path):
        # entered once per non-last layer, never around the last layer's pass, never when the collective is synchronous
        import contextlib
        trace = []
        @contextlib.contextmanager
        def overlapped():
            trace.append("enter")
            try:
                yield
            finally:
                trace.append("exit")
        def cl(*a, **k):
            trace.append("cells_layer"); return ops.cells_layer(*a, **k)
        def cml2(*a, **k):
            trace.append("cells_mean_linear"); return cml(*a, **k)
        ops3 = D.LocalOps(cl, ops.genes_partial, gfin, cml2, lambda width, n_seed: True, overlapped)
        with torch.no_grad():
            l_ov = D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops3, 2, True, sizes)
        assert trace == ["enter", "cells_layer", "exit", "cells_mean_linear"] and torch.equal(l_ov, l_fold)
        trace.clear(); flags.clear()
        D.COLLECTIVE_HOOK = lambda fn: fn()                  # a segmented capture issues the collective synchronously: no overlap
        try:
            with torch.no_grad():
                D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops3, 2, True, sizes)
        finally:
            D.COLLECTIVE_HOOK = None
        assert trace == ["cells_layer", "cells_mean_linear"]
        flags.clear()
        # ---- round 4: `pre_aggregate` (the engine's wait on the previous forward's in-flight logits concat) fires ONCE, after the
        # first layer's projections have been issued and before the first aggregation launch
        trace.clear()
        def lin(x, W, b=None):
            trace.append("linear"); return torch.nn.functional.linear(x, W, b)
        def gp(p_c):
            trace.append("genes_partial"); return ops.genes_partial(p_c)
        ops4 = D.LocalOps(cl, gp, gfin, cml2, lambda width, n_seed: True)
        with torch.no_grad():
            l_pre = D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops4, 2, True, sizes, linear=lin,
                                      pre_aggregate=lambda: trace.append("pre_aggregate"))
        assert trace == ["linear", "linear", "pre_aggregate", "genes_partial", "cells_layer", "cells_mean_linear"]
        assert torch.equal(l_pre, l_fold)
        flags.clear()
        l_grad = D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops2, 2, True, sizes)      # grad mode: no fold
        assert flags == [False]
        np.testing.assert_allclose(l_grad.numpy(), full, atol=1e-6)
        # fp16-stored features (cfg5): widened on the way into the projection
        f16 = feats.half()
        l3 = D.sharded_forward(weights, None, f16[:G], f16[G + lo:G + hi], ops, 2, True, sizes)
        l3ref = D.sharded_forward(weights, None, f16[:G].double(), f16[G + lo:G + hi].double(), ops, 2, True, sizes)
        assert torch.equal(l3, l3ref)

        # ---- train-mode dropout on every layer's input rows (gnn.py:60-64): fixed masks, the gene mask replicated
        rgd = O.build_reference_graph(c["expr"])
        gm = torch.Generator().manual_seed(77)
        masks = [(torch.rand(G + C, w, generator=gm) >= 0.3).double() / 0.7 for w in (12, 8)]
        ld = D.sharded_forward(weights, None, feats[:G], feats[G + lo:G + hi], ops, 2, True, sizes,
                               dropout_masks=[(m[:G], m[G + lo:G + hi]) for m in masks])
        want_d = O.nodeflow_forward({k: v.double() for k, v in sd.items()}, rgd, feats, np.arange(G, G + C), 2,
                                    dropout_masks=masks)
        np.testing.assert_allclose(ld.numpy(), want_d.numpy(), atol=1e-5)
        assert not np.allclose(ld.numpy(), logits.numpy(), atol=1e-3)          # the masks did something
        # the shared stream gives every rank the same gene mask, the local stream differs per rank
        g_sh = torch.Generator().manual_seed(5); g_lo = torch.Generator().manual_seed(6 + rank)
        mg = D.dropout_mask((G, 4), 0.5, g_sh, "cpu"); mc = D.dropout_mask((3, 4), 0.5, g_lo, "cpu")
        both = [torch.zeros_like(mg) for _ in range(world)]
        dist.all_gather(both, mg)
        assert all(torch.equal(b, mg) for b in both)
        assert set(mg.unique().tolist()) <= {0.0, 2.0}

        # ---- data-parallel training step (cfg4): grads after SUM all-reduce == single-process autograd grads
        labels = torch.from_numpy(np.random.default_rng(5).integers(0, 3, C))
        params = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items()}
        al = params["alpha"].reshape(-1)
        tops = D.LocalOps(
            cells_layer=lambda p_g, p_c, b, relu, rows=None, sc=False: torch.relu(((Acg @ (al[:G, None] * p_g)) + al[G + 1] * p_c) * inv_c[:, None] + b),
            genes_partial=lambda p_c: Agc @ p_c,
            genes_finish=lambda part, p_g, b, relu: torch.relu((al[:G, None] * part + al[G] * p_g) * inv_g[:, None].double() + b))
        wfn = lambda: [(params[f"layers.{i}.fc_neigh.weight"], params[f"layers.{i}.fc_neigh.bias"]) for i in range(2)] + \
                      [(params["linear.weight"], params["linear.bias"])]
        opt = torch.optim.SGD(list(params.values()), lr=0.0)           # lr 0: inspect the all-reduced grads
        total = D.sharded_train_step(list(params.values()), wfn, feats[:G], feats[G + lo:G + hi], labels[lo:hi], tops, 2, opt)
        rg = O.build_reference_graph(c["expr"])
        loss_ref, grads_ref, _ = O.loss_and_grads(sd, rg, feats, np.arange(G, G + C), labels, 2)
        assert abs(total - float(loss_ref)) < 1e-5 * max(1.0, abs(float(loss_ref)))
        for k, p_ in params.items():
            np.testing.assert_allclose(p_.grad.numpy(), grads_ref[k].numpy(), atol=1e-6, rtol=1e-4, err_msg=k)   # fp32-normalised graph weights in the oracle

        # round 6: the gradients live in ONE flat bucket (views), and the differentiable [G, H] all-reduce is SPLIT around the
        # rank's own cells<-genes pass in both directions: forward issue -> cells pass -> wait; backward issue (dH1_g) ->
        # backward of that cells pass -> wait.  A logging autograd Function stands in for the pass.
        bucket = opt._wgnn_grad_bucket
        assert all(p_.grad.data_ptr() == v.data_ptr() and p_.grad.shape == v.shape for p_, v in zip(bucket.params, bucket.views))
        assert bucket.flat.numel() == sum(p_.numel() for p_ in params.values())
        events = []

        class LogPass(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                events.append("cells_fwd")
                return x.view_as(x)

            @staticmethod
            def backward(ctx, g):
                events.append("cells_bwd")
                return g
        tops_log = D.LocalOps(lambda p_g, p_c, b, relu, rows=None, sc=False: LogPass.apply(tops.cells_layer(p_g, p_c, b, relu)),
                              tops.genes_partial, tops.genes_finish)
        D.TRACE = events
        try:
            total_l = D.sharded_train_step(list(params.values()), wfn, feats[:G], feats[G + lo:G + hi], labels[lo:hi], tops_log, 2, opt)
        finally:
            D.TRACE = None
        # (the last layer's cells pass logs too: forward after the wait, backward before the backward issue)
        assert events == ["ar_fwd_issue", "cells_fwd", "ar_fwd_wait", "cells_fwd",
                          "cells_bwd", "ar_bwd_issue", "cells_bwd", "ar_bwd_wait"], events
        assert abs(total_l - total) < 1e-9 * max(1.0, abs(total))
        assert opt._wgnn_grad_bucket is bucket                                  # created once, reused
        for k, p_ in params.items():
            np.testing.assert_allclose(p_.grad.numpy(), grads_ref[k].numpy(), atol=1e-6, rtol=1e-4, err_msg=k + " (split all-reduce)")
        # the synchronous differentiable all-reduce stays available and agrees
        xs = torch.full((3,), float(rank + 1), dtype=torch.float64, requires_grad=True)
        ys, fin = D.all_reduce_sum_begin(xs * 2.0)
        (fin(ys) * torch.arange(3.0, dtype=torch.float64)).sum().backward()
        want_sum = 2.0 * sum(r + 1 for r in range(world))
        assert torch.equal(fin(ys).detach(), torch.full((3,), want_sum, dtype=torch.float64)) or world == 1
        np.testing.assert_allclose(xs.grad.numpy(), 2.0 * world * np.arange(3.0))

        # round 4: the same step with the LAST layer in the reference's literal order (`cells_mean_linear`, also in training):
        # same loss, same all-reduced gradients
        def tcml(h_g, h_c, W, b, relu, rows=None, sc=False, prescaled=False):
            assert not prescaled                                        # folding is an inference-path fusion
            y = (((Acg @ (al[:G, None] * h_g)) + al[G + 1] * h_c) * inv_c[:, None]) @ W.t() + b
            return torch.relu(y) if relu else y
        tops2 = D.LocalOps(tops.cells_layer, tops.genes_partial, tops.genes_finish, tcml, lambda width, n_seed: True)
        total2 = D.sharded_train_step(list(params.values()), wfn, feats[:G], feats[G + lo:G + hi], labels[lo:hi], tops2, 2, opt,
                                      sync_loss=False)
        assert torch.is_tensor(total2) and abs(float(total2) - float(loss_ref)) < 1e-5 * max(1.0, abs(float(loss_ref)))
        for k, p_ in params.items():
            np.testing.assert_allclose(p_.grad.numpy(), grads_ref[k].numpy(), atol=1e-6, rtol=1e-4, err_msg=k + " (aggregate-first)")

        # SUM all-reduce of gradients == single-process gradient of the summed loss (train.py:36)
        p = torch.nn.Parameter(torch.ones(5, dtype=torch.float64))
        x = torch.arange(10, dtype=torch.float64).reshape(2, 5)
        (p * x[rank]).sum().backward()
        D.all_reduce_grads([p])
        np.testing.assert_allclose(p.grad.numpy(), x[:world].sum(0).numpy())
        Path(out_dir, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cells", [90, 91])           # equal shards (one all_gather_into_tensor) / ragged shards
def test_world2_sharded_forward_matches_unsharded(tmp_path, cells):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), cells), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_one_rank_group_with_forced_collectives(tmp_path):
    """`dist.FORCE_COLLECTIVES`: a ONE-rank communicator runs the whole sharded orchestration incl. every collective (how the
    GPU suite exercises RCCL on a 1-GPU box); same checks as the world-2 run."""
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), 90), nprocs=1, join=True)
    assert (tmp_path / "ok0").exists()


def test_shard_ranges_partition_cells():
    from scdeepsort_amd.dist import shard_range
    for C in (1, 7, 100_000, 764_741):
        for W in (1, 2, 4, 8):
            spans = [shard_range(C, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == C
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
