"""Cell-axis sharding of the hot path across the GPUs of one node.

The reference is single-process (no collective anywhere in its tree); this is new design.
One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm,
``gloo`` in CPU tests).  Rank p owns a contiguous range of cells:

* cells<-genes passes are row-independent: local CSR rows, replicated gene table -> no communication;
* the genes<-cells pass has ONE real exchange: every rank reduces its own cells into a partial
  ``[G, H]`` sum (NO_ALPHA / NO_MEAN / no self), the partials are all-reduced (20.5 MB at G=20k,
  H=256 - a single bucket), then alpha, the gene self-loop, 1/deg, bias and ReLU are applied
  redundantly on every rank;
* gene-side normalisation needs global per-gene degree and weight sum: two ``[G]`` all-reduces at
  graph-build time;
* inference outputs are concatenated with one all-gather of ``[C/N, n_classes]`` logits;
* training: CrossEntropyLoss(reduction='sum') (train.py:36) makes summed per-rank gradients equal
  the single-GPU gradient, so parameter grads are all-reduced with SUM (no rescale).

The local arithmetic is injected (``local_ops``) so the orchestration is testable on CPU with
gloo; the product binds it to the HIP operators.
"""
from __future__ import annotations

import contextlib
import os

from dataclasses import dataclass
from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(num_cells: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced (sizes differ by <= 1) cell range of ``rank``."""
    base, rem = divmod(num_cells, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# A one-rank communicator normally short-circuits every collective (there is nothing to exchange).  With this switch on,
# an INITIALISED process group of any size - also a single rank - runs them all: that is how the RCCL branch (library
# load, ``device_id=``, stream-ordered ``work.wait()``, ``all_gather_into_tensor``) is exercised on a 1-GPU box, where
# RCCL refuses two ranks on one device (tests/test_gpu_dist.py::test_world1_nccl_*).
FORCE_COLLECTIVES = False


def comm_active() -> bool:
    """True when collectives must actually be issued."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or FORCE_COLLECTIVES


# hipGraph capture of the sharded forward (graphed.GraphedShardedForward).  With RCCL the data-path collectives are captured
# with the kernels (one graph launch per forward).  A backend whose collectives run on the host (gloo: the shared-GPU debug
# mode and the CPU tests) cannot be captured; there the capture is cut at every collective: ``COLLECTIVE_HOOK(fn)`` - set by
# the capturing object for the duration of the capture - ends the current graph segment, runs ``fn`` eagerly, remembers it
# for replay and opens the next segment.
COLLECTIVE_HOOK: Optional[Callable] = None


# CUs a rank leaves to the communicator while a data-path collective is in flight next to a tile pass.  RCCL runs one
# 256-thread workgroup per channel; a tile workgroup needs a whole CU, and the shard geometries fill the chip in ONE round
# (N = 8: 61 x 4 = 244 tiles, N = 4: 121 x 2, N = 2: 256 x 1), so every CU a channel holds sends a tile to a second round:
# measured with a spin kernel standing in for the communicator (profiles/r04_issue_analysis.md §10), the cells<-genes pass of a
# rank's shard takes +55 .. +65 % next to 16 or 32 held CUs, while a geometry for 224 CUs costs -2 .. +5 % when nothing else
# runs and is immune up to 32 held CUs.  Applied to the pass that overlaps the [G, H] all-reduce (sharded.ShardedWgnn.build).
COMM_CUS = int(os.environ.get("WGNN_COMM_CUS", "32"))


def reserve_comm_cus(cap_channels: Optional[bool] = None) -> None:
    """Call BEFORE ``init_process_group("nccl")``.  OPT-IN (round 5): with ``cap_channels=True`` or
    ``WGNN_CAP_NCCL_CHANNELS=1`` the communicator is capped at ``COMM_CUS`` channels (RCCL reads NCCL_MAX_NCHANNELS at
    communicator creation; one channel = one workgroup = one CU taken from the tile kernel), so that the CUs the tile
    geometry leaves free are the CUs the collective uses.  The cap is process-wide - it also limits the gradient SUM
    all-reduce of training and any other communicator of the process - and its value comes from a spin-kernel stand-in on a
    1-GPU lease, never from RCCL between two devices: by default nothing is exported and only the tile geometry of the
    overlapped pass (``COMM_CUS``) applies.  A value the user exported always wins."""
    if cap_channels is None:
        cap_channels = os.environ.get("WGNN_CAP_NCCL_CHANNELS", "0") == "1"
    if cap_channels and COMM_CUS > 0:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(COMM_CUS))


def _issue(fn: Callable):
    """Run one data-path collective ``fn()`` (returns a Work handle or None) - through the capture hook when one is set."""
    if COLLECTIVE_HOOK is not None:
        COLLECTIVE_HOOK(fn)
        return None
    return fn()


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if comm_active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def rescale_gene_side(local_deg: torch.Tensor, local_sum: torch.Tensor, global_deg: torch.Tensor,
                      global_sum: torch.Tensor, local_val_locally_normalised: torch.Tensor,
                      row_of_nnz: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Turn per-shard normalised gene<-cell weights into globally normalised ones.

    Locally ``w = deg_loc * x / sum_loc`` (K4 on the shard); globally the reference needs
    ``deg_glob * x / sum_glob`` (normalize_weight over ALL in-edges of the gene,
    preprocess_internal.py:17-23), i.e. a per-gene factor ``(deg_glob/deg_loc) * (sum_loc/sum_glob)``.
    Returns (values, inv_deg_global).
    """
    fac = torch.where(local_deg > 0, (global_deg / local_deg.clamp(min=1).float()) *
                      (local_sum.double() / global_sum.clamp(min=1e-30)).float(), torch.zeros_like(global_deg))
    return local_val_locally_normalised * fac[row_of_nnz.long()], 1.0 / (global_deg + 1.0)


def normalise_gene_side(local_deg: torch.Tensor, local_sum: torch.Tensor, local_val_locally_normalised: torch.Tensor,
                        row_of_nnz: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``rescale_gene_side`` with the global statistics obtained by two ``[G]`` all-reduces."""
    g_deg = all_reduce_sum_(local_deg.clone().float())
    g_sum = all_reduce_sum_(local_sum.clone().double())
    return rescale_gene_side(local_deg, local_sum, g_deg, g_sum, local_val_locally_normalised, row_of_nnz)


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x.  Every rank's loss depends on the summed value, so the gradient w.r.t. one rank's
    contribution is the SUM over ranks of the upstream gradients: backward is an all-reduce too."""

    @staticmethod
    def forward(ctx, x):
        y = x.clone()
        all_reduce_sum_(y)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        all_reduce_sum_(g)
        return g


def all_reduce_sum(x: torch.Tensor) -> torch.Tensor:
    """Differentiable SUM all-reduce (out of place)."""
    return _AllReduceSum.apply(x) if comm_active() else x


# Test / profiling aid: when a list, the split all-reduce appends "ar_fwd_issue" / "ar_fwd_wait" / "ar_bwd_issue" / "ar_bwd_wait"
# as they happen (tests/test_dist_gloo.py proves the order around the overlapped pass; the GPU timeline test reads it too).
TRACE: Optional[list] = None


def _trace(what: str) -> None:
    if TRACE is not None:
        TRACE.append(what)


class _AllReduceBegin(torch.autograd.Function):
    """First half of the SPLIT differentiable SUM all-reduce (round 6): forward issues the collective asynchronously on a copy
    and returns the in-flight tensor (not to be read before ``_AllReduceEnd``); backward completes the all-reduce of the
    upstream gradient that ``_AllReduceEnd.backward`` put in flight."""

    @staticmethod
    def forward(ctx, x, box):
        ctx.box = box
        y = x.clone()
        box["fwd"] = dist.all_reduce(y, op=dist.ReduceOp.SUM, async_op=True)
        _trace("ar_fwd_issue")
        return y

    @staticmethod
    def backward(ctx, g):
        work = ctx.box.pop("bwd", None)
        if work is not None:
            work.wait()                                 # stream-level dependency on GPU backends, no host sync
        _trace("ar_bwd_wait")
        return g, None


class _AllReduceEnd(torch.autograd.Function):
    """Second half: forward waits for the collective; backward issues the all-reduce of the gradient asynchronously.  The two
    halves bracket the work that overlaps the collective: whatever is recorded BETWEEN them in the forward (the rank's own
    cells<-genes pass) has its backward node scheduled between ``End.backward`` and ``Begin.backward`` (autograd runs ready
    nodes latest-created first), i.e. under the all-reduce of dH1_g."""

    @staticmethod
    def forward(ctx, y, box):
        ctx.box = box
        work = box.pop("fwd", None)
        if work is not None:
            work.wait()
        _trace("ar_fwd_wait")
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.box["bwd"] = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)
        _trace("ar_bwd_issue")
        return g, None


def all_reduce_sum_begin(x: torch.Tensor):
    """Split form of :func:`all_reduce_sum`: returns ``(pending, finish)``; the collective is in flight until ``finish(pending)``
    returns the summed tensor.  Both directions overlap whatever the caller does in between (forward: that work itself;
    backward: its backward)."""
    if not comm_active():
        return x, (lambda y: y)
    box = {}
    return _AllReduceBegin.apply(x, box), (lambda y: _AllReduceEnd.apply(y, box))


class GradBucket:
    """ONE pre-allocated flat buffer that every parameter's ``.grad`` is a view of (X1, SURVEY 8e: ~0.8 MB at cfg3 incl.
    alpha[G+2]): autograd accumulates into the views in place, the SUM all-reduce of the step is one call on ``flat`` - no
    ``cat`` before it, no per-tensor ``copy_`` after it - and the optimizer reads the views."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no parameter requires a gradient")
        dt, dev = self.params[0].dtype, self.params[0].device
        if any(p.dtype != dt or p.device != dev for p in self.params):
            raise ValueError("GradBucket: parameters of one dtype on one device")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def matches(self, params) -> bool:
        ps = [p for p in params if p.requires_grad]
        return len(ps) == len(self.params) and all(a is b for a, b in zip(ps, self.params))

    def zero(self) -> None:
        """``optimizer.zero_grad()`` for bucketed parameters: one memset; (re-)attaches the views as ``.grad``."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def all_reduce(self) -> None:
        for p, v in zip(self.params, self.views):        # a hook / optimizer replaced a .grad: fold it back in (not expected)
            if p.grad is not v and p.grad is not None:
                v.copy_(p.grad)
                p.grad = v
        if comm_active():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)


@dataclass
class LocalOps:
    """Local arithmetic of one shard (bound to the HIP operators in production)."""
    cells_layer: Callable      # (p_g, p_c_local, bias, relu[, rows, self_compact]) -> h_c_local' (of `rows` only when given;
                               #  self_compact: p_c_local already holds one row per entry of rows)
    genes_partial: Callable    # (p_c_local)                               -> partial [G, H] (plain weighted sum)
    genes_finish: Callable     # (partial_sum_global, p_g, bias, relu)     -> h_g'
    cells_mean_linear: Optional[Callable] = None   # (h_g, h_c_local, W, bias, relu[, rows, self_compact, prescaled]) -> the LAST
                               #  layer in the reference's literal order, act(mean-aggregation W^T + b): at equal widths it skips
                               #  the replicated [G, H] x [H, H] product of project-first (used when nothing is differentiated)
    fold_alpha_ok: Optional[Callable] = None       # (width, n_seed_rows | None) -> bool: genes_finish(..., scale_out=True) may write the
                               #  gene rows alpha-folded and cells_mean_linear(..., prescaled=True) reads them as its source
                               #  table without a scale launch (HIP binding only)
    overlapped: Optional[Callable] = None          # () -> context manager around the cells<-genes pass that runs next to the in-flight
                               #  all-reduce (HIP binding at N > 1: that pass plans its tiles for the CUs the communicator leaves)


def dropout_mask(shape, p: float, generator: Optional[torch.Generator], device, dtype=torch.float32) -> torch.Tensor:
    """Inverted-dropout multiplier (0 or 1/(1-p)) drawn from ``generator`` - ``nn.Dropout`` (gnn.py:33-36,62-63) with an
    explicit random stream, so that the REPLICATED gene rows get the same mask on every rank (one generator seeded
    identically everywhere) while every rank draws its own cells' mask from a rank-local one."""
    # two launches (Bernoulli draw in place, scale in place) instead of rand / compare / cast / divide: the masks of a cfg3 step
    # are 4 x [1e5 .. 2e4, 256 .. 400] floats, ~0.5 ms of framework elementwise work in the four-launch form
    m = torch.empty(shape, dtype=dtype, device=device).bernoulli_(1.0 - p, generator=generator)
    return m.mul_(1.0 / (1.0 - p))


def sharded_forward(weights, alpha_unused, feats_g: torch.Tensor, feats_c_local: torch.Tensor, ops: LocalOps,
                    n_layers: int, gather_logits: bool = True, shard_sizes: Optional[Sequence[int]] = None,
                    async_gather: bool = False, dropout_masks: Optional[Sequence[Tuple[torch.Tensor, torch.Tensor]]] = None,
                    relu: bool = True, linear: Callable = torch.nn.functional.linear,
                    seeds_local: Optional[torch.Tensor] = None, pre_aggregate: Optional[Callable] = None):
    """Layer-wise forward over a cell shard.  ``weights`` = list of (W_i, b_i) + (W_out, b_out) last.
    Features may be stored in fp16 (BASELINE cfg5): they are widened on the way into the fp32 projection, i.e. the
    arithmetic is "fp16-rounded inputs, fp32 accumulate".  ``shard_sizes`` (cells per rank, known at graph build)
    lets the logits concat run without a size exchange / host sync; with ``async_gather`` (equal shards only) the
    concat is left running on the communicator's stream and ``(logits_all, work)`` is returned - the caller waits on
    ``work`` before reading, so the output collection of one batch overlaps the next batch's compute.
    ``dropout_masks[i] = (mask_genes [G,D_i], mask_cells_local [C_p,D_i])``: train-mode dropout on the input rows of
    layer i (gnn.py:60-64) - the gene mask must be identical on every rank (see :func:`dropout_mask`).
    ``pre_aggregate``: called once, right before the FIRST aggregation launch of this forward (after the first layer's
    projections): the engine completes the previous forward's in-flight logits concat there, so that the concat runs under
    the projections (library GEMMs share the chip with the communicator's workgroups) and never next to a tile pass."""
    def first_aggregation():
        nonlocal pre_aggregate
        if pre_aggregate is not None:
            pre_aggregate()
            pre_aggregate = None
    h_g, h_c = feats_g, feats_c_local
    compact = False                                     # h_c holds the seeds' rows only (see GNN.embed for the rule)
    folded = False                                      # h_g holds alpha-folded gene rows (written so by genes_finish)
    for i in range(n_layers):
        W, b = weights[i]
        last = i == n_layers - 1
        rows = seeds_local if (seeds_local is not None and i >= n_layers - 2) else None
        if dropout_masks is not None:
            m_g, m_c = dropout_masks[i]
            if compact:
                m_c = m_c[seeds_local.long()]
            h_g, h_c = h_g.to(m_g.dtype) * m_g, h_c.to(m_c.dtype) * m_c
        if last and ops.cells_mean_linear is not None and W.shape[0] == W.shape[1] and h_g.dtype == W.dtype:
            # the last layer in the reference's literal order (gnn.py:65-66), in training as well (round 4; every step of it is
            # differentiable): no replicated [G, H] x [H, H] projection of the gene rows, forward or backward
            kw = {"prescaled": True} if folded else {}
            first_aggregation()
            h_c = (ops.cells_mean_linear(h_g, h_c, W, b, relu, **kw) if rows is None
                   else ops.cells_mean_linear(h_g, h_c, W, b, relu, rows, compact, **kw))
            break
        if getattr(linear, "widens_fp16", False):       # ops.linear: fp16-stored rows are widened inside the GEMM's loader
            p_g, p_c = linear(h_g, W), linear(h_c, W)
        else:
            p_g, p_c = linear(h_g.to(W.dtype), W), linear(h_c.to(W.dtype), W)
        first_aggregation()
        if last:                                        # a seed mini-batch only needs its own rows of the last layer
            h_c = ops.cells_layer(p_g, p_c, b, relu) if rows is None else ops.cells_layer(p_g, p_c, b, relu, rows, compact)
            break
        part = ops.genes_partial(p_c)
        if torch.is_grad_enabled() and part.requires_grad:
            # training: the same overlap, differentiable (round 6).  Forward: the [G, H] all-reduce is issued BEFORE this rank's
            # cells<-genes pass and completed after it; backward: the all-reduce of dH1_g is issued when the gradient of the
            # summed gene rows is known and completed after the backward of that cells<-genes pass (K2t), see _AllReduceEnd.
            part, finish = all_reduce_sum_begin(part)
            with (ops.overlapped() if (comm_active() and ops.overlapped is not None) else contextlib.nullcontext()):
                new_c = ops.cells_layer(p_g, p_c, b, relu) if rows is None else ops.cells_layer(p_g, p_c, b, relu, rows, False)
            part = finish(part)
        else:
            # the ONE data-path collective (X2) runs on the communicator's stream while this rank's
            # cells<-genes pass (row-independent, no communication) computes
            overlap = COLLECTIVE_HOOK is None           # (a segmented capture runs the collective synchronously, now and at replay)
            # (the thunk is kept for replay by a segmented capture: bind THIS layer's tensor, not the loop variable)
            work = _issue(lambda part=part, overlap=overlap: dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=overlap)) \
                if comm_active() else None
            with (ops.overlapped() if (overlap and ops.overlapped is not None) else contextlib.nullcontext()):
                new_c = ops.cells_layer(p_g, p_c, b, relu) if rows is None else ops.cells_layer(p_g, p_c, b, relu, rows, False)
            if work is not None:
                work.wait()                             # stream-level dependency on GPU backends, no host sync
        # the gene rows below a cells-only, aggregate-first last layer are only read as that layer's alpha-folded source table
        Wn = weights[i + 1][0]
        folded = bool(i == n_layers - 2 and ops.fold_alpha_ok is not None and ops.cells_mean_linear is not None
                      and not torch.is_grad_enabled() and dropout_masks is None and Wn.shape[0] == Wn.shape[1] == W.shape[0]
                      # the consumer's `h_g.dtype == W.dtype` test above must hold for the folded rows, or the last layer would
                      # take cells_layer and apply alpha a second time (the HIP binding's fold_alpha_ok also demands f32)
                      and W.dtype == Wn.dtype == p_g.dtype
                      and ops.fold_alpha_ok(W.shape[0], None if seeds_local is None else int(seeds_local.shape[0])))
        h_g = ops.genes_finish(part, p_g, b, relu, scale_out=True) if folded else ops.genes_finish(part, p_g, b, relu)
        h_c = new_c
        compact = rows is not None
    Wo, bo = weights[n_layers]
    logits = torch.nn.functional.linear(h_c, Wo, bo)
    rank, ws = world()
    if gather_logits and comm_active():
        if shard_sizes is None or len(shard_sizes) != ws:
            # the sizes are known when the graph is sharded (ShardedWgnn.build exchanges them once): no per-forward size
            # exchange, no device->host read on the data path
            raise ValueError("gather_logits on a sharded job needs shard_sizes (cells per rank, one entry per rank)")
        if len(set(shard_sizes)) == 1:
            out = torch.empty((ws * logits.shape[0], logits.shape[1]), dtype=logits.dtype, device=logits.device)
            mine, in_flight = logits.contiguous(), async_gather and COLLECTIVE_HOOK is None
            work = _issue(lambda out=out, mine=mine, in_flight=in_flight:
                          dist.all_gather_into_tensor(out, mine, async_op=in_flight))              # X3: inference concat
            return (out, work) if async_gather else out
        mx = max(shard_sizes)                           # ragged shards: pad to the largest, gather, cut
        pad = torch.zeros(mx, logits.shape[1], dtype=logits.dtype, device=logits.device)
        pad[: logits.shape[0]] = logits
        outs = [torch.empty_like(pad) for _ in range(ws)]
        _issue(lambda outs=outs, pad=pad: dist.all_gather(outs, pad))
        cat = torch.cat([o[:n] for o, n in zip(outs, shard_sizes)])
        return (cat, None) if async_gather else cat
    return (logits, None) if async_gather else logits


def all_reduce_grads(params, bucket: Optional[GradBucket] = None) -> None:
    """X1: SUM all-reduce of parameter gradients in one flat bucket (~0.8 MB at cfg3).  With a :class:`GradBucket` the
    gradients already live in the flat buffer (one collective call, nothing else); without one they are packed and unpacked."""
    if bucket is not None:
        bucket.all_reduce()
        return
    if not comm_active():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def grad_bucket_of(optimizer, params) -> GradBucket:
    """The step's gradient bucket, created once per (optimizer, parameter list) and kept on the optimizer object."""
    b = getattr(optimizer, "_wgnn_grad_bucket", None)
    if b is None or not b.matches(params):
        b = GradBucket(params)
        try:
            optimizer._wgnn_grad_bucket = b
        except AttributeError:
            pass
    return b


def sharded_train_step(params, weights_fn, feats_g, feats_c_local, labels_local, ops: LocalOps, n_layers: int,
                       optimizer, seeds_local: Optional[torch.Tensor] = None, dropout_masks=None, relu: bool = True,
                       linear: Callable = torch.nn.functional.linear, loss_sum: Optional[Callable] = None,
                       sync_loss: bool = True):
    """One full-batch data-parallel training step over cell shards (BASELINE cfg4).

    loss = CrossEntropyLoss(reduction='sum') over this rank's cells (train.py:36); because the loss is a SUM, adding
    the per-rank parameter gradients (all_reduce_grads) reproduces the single-process gradient exactly, and every rank
    then applies the identical optimizer step.  Returns the global loss: a Python float (``sync_loss``: one device->host read
    per step, which also lets the host wait for the whole step before it enqueues the next one - ~0.9 ms of exposed launch
    latency per cfg3 step) or, with ``sync_loss=False``, a 0-d device tensor the caller reads when it wants to."""
    logits = sharded_forward(weights_fn(), None, feats_g, feats_c_local, ops, n_layers, gather_logits=False,
                             dropout_masks=dropout_masks, relu=relu, linear=linear, seeds_local=seeds_local)
    loss = loss_sum(logits, labels_local) if loss_sum is not None else \
        torch.nn.functional.cross_entropy(logits, labels_local, reduction="sum")     # train.py:36
    bucket = grad_bucket_of(optimizer, params)          # .grad of every parameter = a view of ONE flat buffer
    bucket.zero()                                       # (optimizer.zero_grad() would drop the views)
    loss.backward()
    all_reduce_grads(params, bucket)
    optimizer.step()
    total = loss.detach().clone()
    all_reduce_sum_(total)
    return float(total) if sync_loss else total
