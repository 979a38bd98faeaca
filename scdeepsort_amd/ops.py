"""Host-side operators of the hot path: thin wrappers over the C ABI (``include/wgnn.h``)
plus the ``torch.autograd.Function`` that stands where the reference calls

    nf.block_compute(i, self.message_func, fn.mean('m', 'neigh'), layer)   (models/gnn.py:65)

torch is used for device memory, streams and autograd bookkeeping only; all
aggregation arithmetic runs in the HIP kernels.  There is no CPU fallback: a
non-CUDA tensor or a missing ``libwgnn_hip.so`` raises ``WgnnError``.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import DST_IS_GENE, NO_ALPHA, SRC_IS_GENE, WgnnError
from .graph import AggCsr, Plan, _ptr, _stream


def _require_cuda(*ts: Optional[torch.Tensor]) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise WgnnError("wgnn operators run on the GPU only (tensor on %s); there is no CPU fallback" % t.device)
        dev = t.device
    return dev


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    if t.dim() != 2:
        raise ValueError("expected a 2-D feature matrix")
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.float16:
        return _lib.F16
    raise WgnnError(f"unsupported feature dtype {t.dtype}")


# Optional launch-level timing hook (bench.py): when set to a list, every K1 call appends
# (tag, start_event, end_event) recorded on the stream the kernels are enqueued on.
PROFILE = None
DEBUG_FLAGS = 0      # ablation switches of the tiled kernel (timing experiments only)
# K1 dispatch: passes with nnz*D above this go to the LDS-streamed kernel (None = always row-wave)
SAVE_NEIGH_SUM = True               # training: the forward of gene rows saves its raw neighbour sums (no K3 pass in backward)
SEED_BLOCK_MAX_CAP = 12_000_000     # B x longest row above which a seed batch's backward walks the full transposed graph instead
                                    # (sorting the padded block costs ~0.05 ms per million slots; the full K2t pass 1.2 ms at cfg3)
PAD_NARROW_TO_256 = False           # round-1 behaviour (hidden < 256 carried as 256 zero-padded columns); kept for A/B timing
# nnz * max(D, 128) above which the LDS-streamed kernels take a pass.  The tile kernel's time hardly depends on D (it is bound
# by per-edge instruction issue), the row-wave kernel's gathers scale with it: at BASELINE cfg2's 2.0 M edges a pass costs
# 60 / 59 us tiled against 62 / 71 us row-wave at D = 128 and 62 / 61 against 100 / 129 us at D = 200 (round 4,
# profiles/r04_issue_analysis.md §4) - rounds 1-3 used 5e8 (~2 M edges at D = 256), which left cfg2 and every hidden-200 graph of
# that size on the row-wave kernel.
TILED_MIN_WORK = int(__import__("os").environ.get("WGNN_TILED_MIN_WORK", 250_000_000))
SEED_FULL_PASS_MIN_FRAC = 0.2       # a seed set of at least this share of the rows of a tile-kernel operand runs the FULL LDS-streamed
                                    # pass and gathers its rows (row-wave K1 costs ~5x per edge: 6.2 vs 1.19 ms for all of cfg3)


def tiled_kernel_serves(csr: AggCsr, D: int) -> bool:
    """True when a FULL pass over ``csr`` at width ``D`` is dispatched to the LDS-streamed kernel (K1t)."""
    return (TILED_MIN_WORK is not None and D <= 256 and D % 4 == 0 and csr.nnz * max(D, 128) >= TILED_MIN_WORK
            and csr.ell_cnt is None)


def will_run_tiled(csr: AggCsr, D: int, n_seed_rows: Optional[int] = None) -> bool:
    """True when ``agg_fwd`` on f32 rows of width ``D`` (all rows, or a seed set of ``n_seed_rows`` rows) takes the
    LDS-streamed route - the one that honours ``src_scaled``."""
    return tiled_kernel_serves(csr, D) and (n_seed_rows is None or n_seed_rows >= SEED_FULL_PASS_MIN_FRAC * csr.n_rows)


class _Timed:
    """``with _Timed(dev, tag):`` records one (tag, start, end) HIP-event triple into ``PROFILE`` on the launch stream
    (a no-op when no profile is being collected)."""

    def __init__(self, dev, tag):
        self.dev, self.tag, self.ev = dev, tag, None

    def __enter__(self):
        if PROFILE is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.ev is not None and PROFILE is not None:
            self.ev[1].record(torch.cuda.current_stream(self.dev))
            PROFILE.append((self.tag, self.ev[0], self.ev[1]))
        return False


def _partials(plan: Plan, D: int, device) -> Optional[torch.Tensor]:
    return torch.empty(plan.n_partials * D, dtype=torch.float32, device=device) if plan.n_partials else None


def agg_fwd(csr: AggCsr, alpha: Optional[torch.Tensor], mode: int, self_idx: int,
            h_src: torch.Tensor, h_self: Optional[torch.Tensor], *, bias: Optional[torch.Tensor] = None,
            relu: bool = False, row_ids: Optional[torch.Tensor] = None, self_compact: bool = False,
            no_mean: bool = False, out_dtype: Optional[torch.dtype] = None,
            out: Optional[torch.Tensor] = None, neigh_sum: Optional[torch.Tensor] = None,
            src_scaled: Optional[torch.Tensor] = None, out_scale_alpha: bool = False) -> torch.Tensor:
    """K1 ``wgnn_agg_fwd``: weighted mean of in-neighbours incl. the implicit self-loop.  ``neigh_sum`` (f32 [n_out, D],
    contiguous) optionally receives the raw neighbour sum of every output row (saved by training for dalpha).
    ``src_scaled`` (SRC_IS_GENE only): ``alpha[s] * h_src[s]`` already formed (``linear_fwd(..., row_scale=alpha)`` or a
    gene pass run with ``out_scale_alpha``) - the LDS-streamed kernel then reads it in place of its own scale pass; the
    row-wave kernel ignores it (callers check ``will_run_tiled`` before handing over an ONLY-scaled table).
    ``out_scale_alpha`` (DST_IS_GENE): the finished gene rows are written multiplied by alpha[row]
    (WGNN_FLAG_OUT_SCALE_ALPHA) - the next layer's alpha-folded source table, no separate scale launch."""
    dev = _require_cuda(h_src, h_self, alpha, bias, csr.col)
    h_src = _rowmajor(h_src)
    D = h_src.shape[1]
    if D % 4:
        raise WgnnError(f"feature width {D} must be a multiple of 4")
    if out_scale_alpha and mode != DST_IS_GENE:
        raise WgnnError("out_scale_alpha is defined for gene rows (DST_IS_GENE) only")
    tiled_ok = (out is None and h_src.dtype == torch.float32 and (out_dtype in (None, torch.float32))
                and tiled_kernel_serves(csr, D))
    if tiled_ok and row_ids is None:
        return agg_fwd_tiled(csr, csr.tile_plan(tiled_block_rows(D)), alpha, mode, self_idx, h_src, h_self, bias=bias, relu=relu,
                             no_mean=no_mean, neigh_sum=neigh_sum, src_scaled=src_scaled, out_scale_alpha=out_scale_alpha)
    if (tiled_ok and neigh_sum is None and row_ids.shape[0] >= SEED_FULL_PASS_MIN_FRAC * csr.n_rows
            and (h_self is None or h_self.dtype == torch.float32)):
        # a LARGE seed set (predict.py:61-88: every test cell is a seed; fit's accuracy() over the training cells): one full
        # LDS-streamed pass over all rows, then the seeds' rows in seed order.  A compact self table (one row per seed slot)
        # is spread to row positions first; rows outside the seed set produce values nobody reads.
        idl = row_ids.to(device=dev, dtype=torch.long)
        if h_self is not None and self_compact:
            full_self = torch.empty((csr.n_rows, D), dtype=torch.float32, device=dev)
            full_self.index_copy_(0, idl, _rowmajor(h_self))
            h_self = full_self
        full = agg_fwd_tiled(csr, csr.tile_plan(tiled_block_rows(D)), alpha, mode, self_idx, h_src, h_self, bias=bias, relu=relu,
                             no_mean=no_mean, src_scaled=src_scaled, out_scale_alpha=out_scale_alpha)
        return full.index_select(0, idl)
    if src_scaled is not None and src_scaled.data_ptr() == h_src.data_ptr():
        raise WgnnError("an alpha-folded table without its unscaled original reached the row-wave kernel, which folds alpha "
                        "per edge itself (callers check ops.will_run_tiled first)")
    if h_self is not None:
        h_self = _rowmajor(h_self)
        if h_self.dtype != h_src.dtype or h_self.shape[1] != D:
            raise WgnnError("h_self must match h_src in dtype and width")
    if row_ids is not None:
        ids, plan = csr.subplan(row_ids)
        n_out = ids.shape[0]
    else:
        ids, plan, n_out = None, csr.plan, csr.n_rows
    out_dtype = out_dtype or h_src.dtype
    if out is None:
        out = torch.empty((n_out, D), dtype=out_dtype, device=dev)
    if n_out == 0:
        return out
    flags = (_lib.FLAG_RELU if relu else 0) | (_lib.FLAG_NO_MEAN if no_mean else 0) | \
            (_lib.FLAG_NO_SELF if h_self is None else 0) | (_lib.FLAG_SELF_COMPACT if self_compact else 0) | \
            (_lib.FLAG_OUT_SCALE_ALPHA if out_scale_alpha else 0)
    if alpha is not None:
        alpha = alpha.reshape(-1)
        if alpha.dtype != torch.float32 or not alpha.is_contiguous():
            alpha = alpha.float().contiguous()
    if bias is not None:
        bias = bias.float().contiguous()
    part = _partials(plan, D, dev)
    ev = None
    if PROFILE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(torch.cuda.current_stream(dev))
    rc = _lib.call(dev, "wgnn_agg_fwd",
        _ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), _ptr(alpha), mode, self_idx,
        _ptr(h_src), h_src.stride(0), _ptr(h_self), h_self.stride(0) if h_self is not None else 0,
        _ptr(ids), _ptr(csr.inv_deg), _ptr(bias), _ptr(out), out.stride(0), _ptr(neigh_sum), n_out, D,
        _dtype_code(h_src), _dtype_code(out), flags,
        _ptr(plan.items), plan.n_items, _ptr(plan.long_rows) if plan.n_long else None, plan.n_long,
        _ptr(part), plan.n_partials, _stream(dev))
    _lib.check(rc, "wgnn_agg_fwd")
    if ev is not None:
        ev[1].record(torch.cuda.current_stream(dev))
        PROFILE.append((("rows", csr.n_rows, "cols", csr.n_cols, "nnz", csr.nnz, "D", D, "mode", mode,
                         "kernel", "agg_main"), ev[0], ev[1]))
    return out


def agg_bwd_src(csr: AggCsr, alpha: Optional[torch.Tensor], mode: int, g: torch.Tensor,
                h_src: Optional[torch.Tensor], dalpha: Optional[torch.Tensor] = None,
                dh_src: Optional[torch.Tensor] = None, accumulate: bool = False,
                dst_scale: Optional[torch.Tensor] = None, prescaled: bool = False) -> torch.Tensor:
    """K2 ``wgnn_agg_bwd_src``: gradient w.r.t. the gathered rows (transposed SpMM);
    for SRC_IS_GENE also writes dalpha[0:n_src] = <h_src[s], T[s]>.  ``dst_scale`` replaces the per-destination
    factor 1/(deg+1) (``csr.inv_deg``), e.g. ones for the backward of a plain weighted sum.  ``prescaled``: ``g`` already
    carries the per-destination factors (``agg_bwd_prepare``) - LDS-streamed route only."""
    dev = _require_cuda(g, h_src, alpha)
    inv_deg = csr.inv_deg if dst_scale is None else dst_scale.float().contiguous()
    t = csr.transposed()
    g = _rowmajor(g.float())
    D = g.shape[1]
    if g.shape[0] != csr.n_rows:
        raise WgnnError("g must have one row per destination row of the CSR")
    if dh_src is None:
        dh_src = torch.empty((t.n_rows, D), dtype=torch.float32, device=dev)
        accumulate = False
    if h_src is not None:
        h_src = _rowmajor(h_src.float())
    if alpha is not None:
        alpha = alpha.reshape(-1).float().contiguous()
    if tiled_kernel_serves(csr, D):
        # K2t: LDS-streamed kernel over the transposed structure; per-destination factors folded into g once
        tp = t.tile_plan(tiled_block_rows(D))
        g = g.contiguous()
        if prescaled:
            scale, scratch = None, None
        else:
            scale = (inv_deg if mode != DST_IS_GENE else inv_deg * alpha[: csr.n_rows]).contiguous()
            scratch = torch.empty_like(g)
        part = torch.empty(tp.n_partials * D, dtype=torch.float32, device=dev) if tp.n_partials else None
        n_long = tp.long_rows.shape[0]
        with _Timed(dev, ("rows", t.n_rows, "cols", t.n_cols, "nnz", t.nnz, "D", D, "mode", mode, "kernel",
                          "agg_tiled_tall<EPI_BWD_SRC>" if tp.geom.tall else "agg_tiled_flat4<EPI_BWD_SRC>")):
            rc = _lib.call(dev, "wgnn_agg_bwd_src_tiled",
                _ptr(alpha), mode, _ptr(scale), _ptr(g), g.shape[0], _ptr(scratch),
                _ptr(h_src), h_src.stride(0) if h_src is not None else 0, _ptr(dh_src), dh_src.stride(0), _ptr(dalpha),
                int(accumulate), t.n_rows, D, _ptr(tp.entries), _ptr(tp.seg_ptr), tp.nblk_max, tp.block_rows_arg,
                _ptr(tp.items), _ptr(tp.hdr), tp.n_tiles, _ptr(tp.long_rows) if n_long else None, n_long,
                _ptr(part), tp.n_partials, _stream(dev))
        _lib.check(rc, "wgnn_agg_bwd_src_tiled")
        return dh_src
    if prescaled:
        raise WgnnError("prescaled gradient rows are an input of the LDS-streamed K2t only")
    part = _partials(t.plan, D, dev)
    rc = _lib.call(dev, "wgnn_agg_bwd_src",
        _ptr(t.rowptr), _ptr(t.col), _ptr(t.val), _ptr(alpha), mode, _ptr(inv_deg),
        _ptr(g), g.stride(0), _ptr(h_src), h_src.stride(0) if h_src is not None else 0,
        _ptr(dh_src), dh_src.stride(0), _ptr(dalpha), int(accumulate), t.n_rows, D,
        _ptr(t.plan.items), t.plan.n_items, _ptr(t.plan.long_rows) if t.plan.n_long else None, t.plan.n_long,
        _ptr(part), t.plan.n_partials, _stream(dev))
    _lib.check(rc, "wgnn_agg_bwd_src")
    return dh_src


def agg_bwd_src_block(csr: AggCsr, row_ids: torch.Tensor, alpha: Optional[torch.Tensor], mode: int, g: torch.Tensor,
                      inv_rows: torch.Tensor, h_src: Optional[torch.Tensor], dalpha: Optional[torch.Tensor]) -> torch.Tensor:
    """K2 for ONE seed batch: ``g`` holds one gradient row per seed SLOT ([B, D]); the source-major structure of the
    batch's in-edges comes from ``AggCsr.seed_block_transposed`` (device-built, static shapes).  Writes every source
    row of ``dh_src`` ([n_cols, D]; zero where the batch gathers nothing) and, for SRC_IS_GENE, dalpha[0:n_cols]."""
    dev = _require_cuda(g, h_src, alpha)
    ids32 = row_ids.to(device=dev, dtype=torch.int32).contiguous()
    g = _rowmajor(g.float())
    D = g.shape[1]
    if g.shape[0] != ids32.shape[0]:
        raise WgnnError("g must have one row per seed")
    dh_src = torch.empty((csr.n_cols, D), dtype=torch.float32, device=dev)
    if ids32.shape[0] == 0:
        return dh_src.zero_()
    t_rowptr, t_slot, t_val, items = csr.seed_block_transposed(ids32)
    if h_src is not None:
        h_src = _rowmajor(h_src.float())
    if alpha is not None:
        alpha = alpha.reshape(-1).float().contiguous()
    rc = _lib.call(dev, "wgnn_agg_bwd_src",
        _ptr(t_rowptr), _ptr(t_slot), _ptr(t_val), _ptr(alpha), mode, _ptr(inv_rows.float().contiguous()),
        _ptr(g), g.stride(0), _ptr(h_src), h_src.stride(0) if h_src is not None else 0,
        _ptr(dh_src), dh_src.stride(0), _ptr(dalpha), 0, csr.n_cols, D,
        _ptr(items), items.shape[0], None, 0, None, 0, _stream(dev))
    _lib.check(rc, "wgnn_agg_bwd_src")
    return dh_src


def agg_bwd_alpha(csr: AggCsr, g: torch.Tensor, h_src: torch.Tensor, h_self: Optional[torch.Tensor],
                  row_ids: Optional[torch.Tensor] = None, self_compact: bool = False):
    """K3 ``wgnn_agg_bwd_alpha``: per-row alpha gradients for DST_IS_GENE rows and the self-loop scalar."""
    dev = _require_cuda(g, h_src, h_self)
    g = _rowmajor(g.float()); h_src = _rowmajor(h_src.float())
    D = g.shape[1]
    if row_ids is not None:
        ids, plan = csr.subplan(row_ids)
        n_out = ids.shape[0]
    else:
        ids, plan, n_out = None, csr.plan, csr.n_rows
    if h_self is not None:
        h_self = _rowmajor(h_self.float())
    d_row = torch.empty(n_out, dtype=torch.float32, device=dev)
    d_self = torch.empty(n_out, dtype=torch.float32, device=dev) if h_self is not None else None
    if row_ids is None and tiled_kernel_serves(csr, D):
        tp = csr.tile_plan(tiled_block_rows(D))                                   # K3t
        h_src = h_src.contiguous()
        part = torch.empty(tp.n_partials * D, dtype=torch.float32, device=dev) if tp.n_partials else None
        n_long = tp.long_rows.shape[0]
        rc = _lib.call(dev, "wgnn_agg_bwd_alpha_tiled",
            _ptr(csr.inv_deg), _ptr(g), g.stride(0), _ptr(h_src), _ptr(h_self),
            h_self.stride(0) if h_self is not None else 0, _ptr(d_row), _ptr(d_self), n_out, D,
            _ptr(tp.entries), _ptr(tp.seg_ptr), tp.nblk_max, tp.block_rows_arg, _ptr(tp.items), _ptr(tp.hdr), tp.n_tiles,
            _ptr(tp.long_rows) if n_long else None, n_long, _ptr(part), tp.n_partials, _stream(dev))
        _lib.check(rc, "wgnn_agg_bwd_alpha_tiled")
        return d_row, d_self
    part = _partials(plan, D, dev)
    rc = _lib.call(dev, "wgnn_agg_bwd_alpha",
        _ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), _ptr(csr.inv_deg), _ptr(ids),
        _ptr(g), g.stride(0), _ptr(h_src), h_src.stride(0), _ptr(h_self), h_self.stride(0) if h_self is not None else 0,
        _ptr(d_row), _ptr(d_self), n_out, D, _lib.FLAG_SELF_COMPACT if self_compact else 0,
        _ptr(plan.items), plan.n_items, _ptr(plan.long_rows) if plan.n_long else None, plan.n_long,
        _ptr(part), plan.n_partials, _stream(dev))
    _lib.check(rc, "wgnn_agg_bwd_alpha")
    return d_row, d_self


NARROW_LDS_ROWS = True       # round 4: LDS rows of the flat tile kernel are 256 / 512 / 1024 bytes by width (False: 78-row blocks for
                             # every D, the round-2/3 geometry - the kernel itself always packs; kept for A/B timing)


def flat_lds_row_bytes(D: int) -> int:
    """LDS row stride of ``agg_tiled_flat4`` (csrc/wgnn_tiled.hip::flat_lds_row_bytes): the power of two covering a row."""
    return 256 if D <= 64 else (512 if D <= 128 else 1024)


FUSED_BWD_GLUE = True        # round 4: one wgnn_agg_bwd_prepare launch instead of the ~8 framework elementwise / reduce launches between
                             # the upstream gradient and K2t (A/B switch)


def agg_bwd_prepare(gout: torch.Tensor, out: Optional[torch.Tensor], inv_deg: Optional[torch.Tensor],
                    alpha: Optional[torch.Tensor], mode: int, self_idx: int, *, want_scaled: bool = True,
                    h_self: Optional[torch.Tensor] = None, want_dh_self: bool = False,
                    neigh_sum: Optional[torch.Tensor] = None, want_dself: bool = False, want_dbias: bool = False) -> dict:
    """``wgnn_agg_bwd_prepare``: from the upstream gradient ``gout`` [R, D] of one aggregation pass (and the saved forward
    output ``out`` for the ReLU mask) in ONE read: ``g_scaled`` (K2t's pre-scaled source table), ``dh_self``, ``dalpha_row``
    (needs ``neigh_sum``), ``dself_row`` (needs ``h_self``) and ``dbias`` - whichever are asked for."""
    import ctypes as C
    dev = _require_cuda(gout, out, inv_deg, alpha, h_self, neigh_sum)
    gout = _rowmajor(gout.float())
    R, D = gout.shape
    if out is not None:
        out = _rowmajor(out.float())
    if h_self is not None:
        h_self = _rowmajor(h_self.float())
    if neigh_sum is not None:
        neigh_sum = neigh_sum.float().contiguous()
    if alpha is not None:
        alpha = alpha.reshape(-1).float().contiguous()
    if inv_deg is not None:
        inv_deg = inv_deg.float().contiguous()
    res = {"g_scaled": torch.empty((R, D), dtype=torch.float32, device=dev) if want_scaled else None,
           "dh_self": torch.empty((R, D), dtype=torch.float32, device=dev) if want_dh_self else None,
           "dalpha_row": torch.empty(R, dtype=torch.float32, device=dev) if neigh_sum is not None else None,
           "dself_row": torch.empty(R, dtype=torch.float32, device=dev) if (want_dself and h_self is not None) else None,
           "dbias": torch.empty(D, dtype=torch.float32, device=dev) if want_dbias else None}
    ws, nf = None, C.c_int64(0)
    if want_dbias:
        _lib.check(_lib.lib().wgnn_agg_bwd_prepare_workspace(R, D, C.addressof(nf)), "wgnn_agg_bwd_prepare_workspace")
        ws = torch.empty(max(1, nf.value), dtype=torch.float32, device=dev)
    need_self = res["dh_self"] is not None or res["dself_row"] is not None
    rc = _lib.call(dev, "wgnn_agg_bwd_prepare", _ptr(gout), gout.stride(0), _ptr(out), out.stride(0) if out is not None else 0,
                   _ptr(inv_deg), _ptr(alpha), mode, self_idx, _ptr(res["g_scaled"]),
                   _ptr(h_self) if need_self else None, h_self.stride(0) if (need_self and h_self is not None) else 0,
                   _ptr(res["dh_self"]), D, _ptr(neigh_sum), _ptr(res["dalpha_row"]), _ptr(res["dself_row"]), _ptr(res["dbias"]),
                   R, D, _ptr(ws), nf.value, _stream(dev))
    _lib.check(rc, "wgnn_agg_bwd_prepare")
    return res


class _CrossEntropySum(torch.autograd.Function):
    """``CrossEntropyLoss(reduction='sum')`` (train.py:36) through ``wgnn_ce_sum_fwd_bwd``: loss and softmax - onehot from one
    read of the logits (the framework's log_softmax / nll_loss pair costs ~0.2 ms per step at 1e5 rows x 16 classes)."""

    @staticmethod
    def forward(ctx, logits, labels):
        import ctypes as C
        dev = _require_cuda(logits, labels)
        x = logits.float()
        if x.stride(1) != 1:
            x = x.contiguous()
        y = labels.to(torch.int64).contiguous()
        n, c = x.shape
        nf = C.c_int64(0)
        _lib.check(_lib.lib().wgnn_ce_sum_workspace(n, C.addressof(nf)), "wgnn_ce_sum_workspace")
        ws = torch.empty(max(1, nf.value), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        d = torch.empty((n, c), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        rc = _lib.call(dev, "wgnn_ce_sum_fwd_bwd", _ptr(x), x.stride(0), _ptr(y), n, c, _ptr(loss), _ptr(d), c, _ptr(ws), nf.value,
                       _stream(dev))
        _lib.check(rc, "wgnn_ce_sum_fwd_bwd")
        ctx.save_for_backward(d)
        ctx.in_dtype = logits.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return (d * g).to(ctx.in_dtype), None


def cross_entropy_sum(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """``F.cross_entropy(logits, labels, reduction='sum')`` on the GPU path (2-D logits, class-index labels)."""
    if logits.dim() != 2 or labels.dim() != 1 or labels.shape[0] != logits.shape[0]:
        raise ValueError("cross_entropy_sum: logits [n, classes], labels [n]")
    if logits.shape[0] == 0:
        return logits.sum() * 0.0
    return _CrossEntropySum.apply(logits, labels)


def tiled_block_rows(D: int) -> int:
    """Source rows per LDS block of the tile kernels: two buffers + the 4 KiB of per-wave weight strips fill the 160 KiB of
    a CU (measured best at D = 256: 78 x 1 KiB x 2).  Narrower rows pack closer (row stride 512 B at D <= 128, 256 B at
    D <= 64): 156 / 255 rows per block - half the per-block barriers and pipeline warm-ups per edge, which measured
    NEUTRAL (cfg2 60.9 / 58.8 -> 60.0 / 59.5 us per pass, cfg3 at D = 128 1.050 / 0.956 -> 1.039 / 0.972 ms: the kernel is
    bound by per-edge issue latency, not by its barriers).  An entry names its source row inside a block with 8 bits, so a
    block holds at most 255 rows."""
    if not NARROW_LDS_ROWS:
        return 78
    return min(255, (160 * 1024 - 4096) // 2 // flat_lds_row_bytes(D))


def agg_fwd_tiled(csr: AggCsr, tplan, alpha: Optional[torch.Tensor], mode: int, self_idx: int,
                  h_src: torch.Tensor, h_self: Optional[torch.Tensor], *, bias: Optional[torch.Tensor] = None,
                  relu: bool = False, no_mean: bool = False, neigh_sum: Optional[torch.Tensor] = None,
                  src_scaled: Optional[torch.Tensor] = None, out_scale_alpha: bool = False) -> torch.Tensor:
    """K1t ``wgnn_agg_fwd_tiled``: same result as :func:`agg_fwd`, source table streamed through LDS.  ``src_scaled``: the
    alpha-folded source table (SRC_IS_GENE) when the caller already has it (WGNN_FLAG_SRC_PRESCALED)."""
    dev = _require_cuda(h_src, h_self, alpha, bias, csr.col)
    if h_src.dtype != torch.float32:
        raise WgnnError("tiled kernel is f32 only")
    h_src = h_src.contiguous()
    D = h_src.shape[1]
    if h_self is not None:
        h_self = _rowmajor(h_self)
    out = torch.empty((csr.n_rows, D), dtype=torch.float32, device=dev)
    flags = (_lib.FLAG_RELU if relu else 0) | (_lib.FLAG_NO_MEAN if no_mean else 0) | \
            (_lib.FLAG_NO_SELF if h_self is None else 0) | (_lib.FLAG_OUT_SCALE_ALPHA if out_scale_alpha else 0) | DEBUG_FLAGS
    if alpha is not None:
        alpha = alpha.reshape(-1)
        if alpha.dtype != torch.float32 or not alpha.is_contiguous():
            alpha = alpha.float().contiguous()
    if bias is not None:
        bias = bias.float().contiguous()
    part = torch.empty(tplan.n_partials * D, dtype=torch.float32, device=dev) if tplan.n_partials else None
    ev = None
    if PROFILE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(torch.cuda.current_stream(dev))
    n_long = tplan.long_rows.shape[0]
    scratch = None
    if mode == SRC_IS_GENE:
        if src_scaled is not None and src_scaled.shape == h_src.shape and src_scaled.dtype == torch.float32:
            h_src, flags = src_scaled.contiguous(), flags | _lib.FLAG_SRC_PRESCALED
        else:
            scratch = torch.empty_like(h_src)
    rc = _lib.call(dev, "wgnn_agg_fwd_tiled",
        _ptr(csr.rowptr), _ptr(alpha), mode, self_idx,
        _ptr(h_src), h_src.shape[0], _ptr(scratch), _ptr(h_self), h_self.stride(0) if h_self is not None else 0,
        None, _ptr(csr.inv_deg), _ptr(bias), _ptr(out), out.stride(0), _ptr(neigh_sum), csr.n_rows, D, flags,
        _ptr(tplan.entries), _ptr(tplan.seg_ptr), tplan.nblk_max, tplan.block_rows_arg, _ptr(tplan.items), _ptr(tplan.hdr), tplan.n_tiles,
        _ptr(tplan.long_rows) if n_long else None, n_long, _ptr(part), tplan.n_partials, _stream(dev))
    _lib.check(rc, "wgnn_agg_fwd_tiled")
    if ev is not None:
        ev[1].record(torch.cuda.current_stream(dev))
        PROFILE.append((("rows", csr.n_rows, "cols", csr.n_cols, "nnz", csr.nnz, "D", D, "mode", mode,
                         "kernel", "agg_tiled_tall" if tplan.geom.tall else "agg_tiled_flat4"), ev[0], ev[1]))
    return out


class WeightedMeanAggregate(torch.autograd.Function):
    """Differentiable ``block_compute(message_func, fn.mean)`` (+ fused bias / ReLU).

    forward  = K1;  backward = K2 (dh_src, dalpha[genes] for gene->cell edges),
    K3 (dalpha for cell->gene edges + self-loop scalar) and a row scale for dh_self.
    Gradients follow train.py:84's autograd through gnn.py:47-56,65:
      dh[u]     = sum_{e=(u->v)} alpha[k(e)] w_e g[v]/deg(v)
      dalpha[k] = sum_{e:k(e)=k} w_e <g[v], h[u]>/deg(v)           (w has no grad)
    """

    @staticmethod
    def forward(ctx, h_src, h_self, alpha, bias, csr: AggCsr, mode: int, self_idx: int, relu: bool,
                row_ids, self_compact: bool):
        # gene rows (DST_IS_GENE) in training: keep the raw neighbour sum S[r]; dalpha[r] = inv_deg[r]*<g[r], S[r]> is then
        # a row dot product in backward instead of a second pass over all edges (K3)
        nsum = None
        if (mode == DST_IS_GENE and row_ids is None and ctx.needs_input_grad[2] and h_src.dtype == torch.float32
                and SAVE_NEIGH_SUM):
            nsum = torch.empty((csr.n_rows, h_src.shape[1]), dtype=torch.float32, device=h_src.device)
        out = agg_fwd(csr, alpha, mode, self_idx, h_src, h_self, bias=bias, relu=relu, row_ids=row_ids,
                      self_compact=self_compact, neigh_sum=nsum)
        ctx.csr, ctx.mode, ctx.self_idx, ctx.relu, ctx.self_compact = csr, mode, self_idx, relu, self_compact
        ctx.has_bias = bias is not None
        ctx.save_for_backward(h_src, h_self, alpha, out if relu else None, row_ids, nsum)
        return out

    @staticmethod
    def backward(ctx, gout):
        h_src, h_self, alpha, out, row_ids, nsum = ctx.saved_tensors
        csr: AggCsr = ctx.csr
        mode, self_idx = ctx.mode, ctx.self_idx
        Dg = gout.shape[1]
        if (FUSED_BWD_GLUE and row_ids is None and gout.is_cuda and Dg % 4 == 0 and tiled_kernel_serves(csr, Dg)
                and (mode != DST_IS_GENE or nsum is not None or not ctx.needs_input_grad[2])
                and (h_self is None or h_self.dtype == torch.float32) and h_src.dtype == torch.float32):
            # full pass on the LDS-streamed route: ONE fused launch turns the upstream gradient into K2t's pre-scaled source
            # table, the self-row gradient, the alpha row dots and the bias gradient
            a = alpha.reshape(-1)
            want_dalpha = ctx.needs_input_grad[2] and mode != NO_ALPHA
            want_src_dalpha = want_dalpha and mode == SRC_IS_GENE
            need_k2 = ctx.needs_input_grad[0] or want_src_dalpha
            res = agg_bwd_prepare(gout, out if ctx.relu else None, csr.inv_deg, a if mode != NO_ALPHA else None, mode, self_idx,
                                  want_scaled=need_k2, h_self=h_self, want_dh_self=h_self is not None and ctx.needs_input_grad[1],
                                  neigh_sum=nsum if (want_dalpha and mode == DST_IS_GENE) else None,
                                  want_dself=want_dalpha and h_self is not None, want_dbias=ctx.has_bias)
            dalpha = torch.zeros_like(a) if ctx.needs_input_grad[2] else None
            dh_src = None
            if need_k2:
                dh_src = agg_bwd_src(csr, a if mode != NO_ALPHA else None, mode, res["g_scaled"], h_src if want_src_dalpha else None,
                                     dalpha if want_src_dalpha else None, prescaled=True)
                dh_src = dh_src.to(h_src.dtype) if ctx.needs_input_grad[0] else None
            if dalpha is not None:
                if res["dalpha_row"] is not None:
                    dalpha[: csr.n_rows] += res["dalpha_row"]
                if res["dself_row"] is not None:
                    dalpha[self_idx] += res["dself_row"].sum()
                dalpha = dalpha.reshape(alpha.shape)
            return dh_src, res["dh_self"], dalpha, res["dbias"], None, None, None, None, None, None
        g = gout.float()
        if ctx.relu:
            g = g * (out > 0)
        g = g.contiguous()
        dbias = g.sum(0) if ctx.has_bias else None
        D = g.shape[1]
        dev = g.device
        a = alpha.reshape(-1)
        rows = row_ids.long() if row_ids is not None else None
        inv_rows = csr.inv_deg if rows is None else csr.inv_deg[rows]
        dalpha = torch.zeros_like(a) if ctx.needs_input_grad[2] else None
        need_src = ctx.needs_input_grad[0]
        want_src_dalpha = dalpha is not None and mode == SRC_IS_GENE
        dh_src = None
        if need_src or want_src_dalpha:
            if rows is None:
                dh_src = agg_bwd_src(csr, a if mode != NO_ALPHA else None, mode, g, h_src if want_src_dalpha else None, dalpha)
            elif mode != DST_IS_GENE and rows.shape[0] * max(1, csr.max_row_nnz) <= SEED_BLOCK_MAX_CAP:
                # seed mini-batch (train.py:71-87): K2 over the source-major view of just the batch's in-edges, built on
                # the device with static shapes - no [n_rows, D] zero-padded gradient, no pass over the whole graph, no
                # host synchronisation.  Repeated seeds are separate slots, so their gradients add up.
                dh_src = agg_bwd_src_block(csr, row_ids, a if mode != NO_ALPHA else None, mode, g, inv_rows,
                                           h_src if want_src_dalpha else None, dalpha)
            else:                                       # gene rows as a subset (not produced by GNN) or a "batch" of most of the
                                                        # graph (block capacity B x longest row too large): generic route
                g_full = torch.zeros((csr.n_rows, D), dtype=torch.float32, device=dev).index_add_(0, rows, g)
                dh_src = agg_bwd_src(csr, a if mode != NO_ALPHA else None, mode, g_full, h_src if want_src_dalpha else None, dalpha)
            dh_src = dh_src.to(h_src.dtype)
        dh_self = None
        hs_rows = None
        if h_self is not None:
            hs_rows = h_self if (rows is None or ctx.self_compact) else h_self[rows]
            coef = (a[self_idx] if mode != NO_ALPHA else 1.0) * inv_rows
            if ctx.needs_input_grad[1]:
                d = (g * coef.unsqueeze(1)).to(h_self.dtype)
                if rows is None or ctx.self_compact:
                    dh_self = d
                else:
                    dh_self = torch.zeros_like(h_self).index_add_(0, rows, d)      # a repeated seed contributes twice
        if dalpha is not None and mode != NO_ALPHA:
            if mode == DST_IS_GENE and nsum is not None:
                dalpha[: csr.n_rows] += (g * nsum).sum(1) * inv_rows
                if hs_rows is not None:
                    dalpha[self_idx] += ((g * hs_rows.float()).sum(1) * inv_rows).sum()
            elif mode == DST_IS_GENE:
                d_row, d_self = agg_bwd_alpha(csr, g, h_src, hs_rows, row_ids, self_compact=True if rows is not None else False)
                if rows is None:
                    dalpha[: csr.n_rows] += d_row
                else:
                    dalpha.index_add_(0, rows, d_row)
                if d_self is not None:
                    dalpha[self_idx] += d_self.sum()
            elif hs_rows is not None:
                dalpha[self_idx] += ((g * hs_rows.float()).sum(1) * inv_rows).sum()
        if dalpha is not None:
            dalpha = dalpha.reshape(alpha.shape)
        return dh_src, dh_self, dalpha, dbias, None, None, None, None, None, None


def weighted_mean_aggregate(csr: AggCsr, alpha: torch.Tensor, mode: int, self_idx: int, h_src: torch.Tensor,
                            h_self: Optional[torch.Tensor], bias: Optional[torch.Tensor] = None, relu: bool = False,
                            row_ids: Optional[torch.Tensor] = None, self_compact: bool = False,
                            src_scaled: Optional[torch.Tensor] = None, out_scale_alpha: bool = False) -> torch.Tensor:
    """Differentiable K1 (+ fused bias / ReLU).  ``src_scaled``: the caller's alpha-folded source table; ``out_scale_alpha``:
    gene rows written alpha-folded for the next layer (see ``agg_fwd``) - both only when nothing is recorded for backward
    (they are functions of alpha that autograd does not see)."""
    if out_scale_alpha and torch.is_grad_enabled():
        raise WgnnError("out_scale_alpha is an inference-path fusion: not differentiable")
    if (src_scaled is not None or out_scale_alpha) and not torch.is_grad_enabled():
        return agg_fwd(csr, alpha, mode, self_idx, h_src, h_self, bias=bias, relu=relu, row_ids=row_ids,
                       self_compact=self_compact, src_scaled=src_scaled, out_scale_alpha=out_scale_alpha)
    return WeightedMeanAggregate.apply(h_src, h_self, alpha, bias, csr, mode, self_idx, relu, row_ids, self_compact)


class _WeightedSum(torch.autograd.Function):
    """out = A @ h_src (plain weighted sum: NO_ALPHA, no mean, no self-loop) - the per-shard partial of the
    genes<-cells pass.  backward: dh_src = A^T g (K2 over the transposed structure, unit column scale)."""

    @staticmethod
    def forward(ctx, h_src, csr: AggCsr):
        ctx.csr = csr
        return agg_fwd(csr, None, NO_ALPHA, 0, h_src, None, no_mean=True)

    @staticmethod
    def backward(ctx, g):
        csr: AggCsr = ctx.csr
        ones = getattr(csr, "_ones", None)
        if ones is None or ones.shape[0] != csr.n_rows:
            ones = torch.ones(csr.n_rows, dtype=torch.float32, device=g.device)
            csr._ones = ones
        return agg_bwd_src(csr, None, NO_ALPHA, g, None, dst_scale=ones), None      # plain A^T g: unit factor


def weighted_sum(csr: AggCsr, h_src: torch.Tensor) -> torch.Tensor:
    return _WeightedSum.apply(h_src, csr)


# ------------------------------------------------------------------------------------------------
# dense half of a layer through the C ABI (fp32 matrix cores) - see csrc/wgnn_linear.hip
# ------------------------------------------------------------------------------------------------
# GNN's projections on the no-grad path: which of them run through wgnn_linear_fwd_ex instead of the library GEMM
#   "auto"  - fp16-stored inputs (never materialised in fp32) and shapes where the kernel measured faster than hipBLASLt
#             (>= 50k rows, K >= 384: 215 vs 231 us on 100k x 400 x 256, profiles/r03_kernel_stats_*.txt);
#   "always" / "never" - A/B switches.
# WGNN_LINEAR_DUAL: also produce the gene table P_g together with alpha * P_g (one kernel, no scale_rows launch).  Off by
# default: on the 20k-row gene projections the 128 x 128-tile kernel runs 69 us against the library's 31 us + 8 us of
# scale_rows (314 tiles = one thin round over 256 CUs), so the fusion costs more than it saves at cfg3.
WGNN_LINEAR = __import__("os").environ.get("WGNN_LINEAR", "auto")
WGNN_LINEAR_DUAL = __import__("os").environ.get("WGNN_LINEAR_DUAL", "0") == "1"
WGNN_LINEAR_MIN_ROWS, WGNN_LINEAR_MIN_K = 50_000, 384


def use_wgnn_linear(x: torch.Tensor, weight: torch.Tensor, dual: bool = False) -> bool:
    """Routing rule of the model's no-grad projections (see WGNN_LINEAR)."""
    if WGNN_LINEAR == "never" or not x.is_cuda or x.dim() != 2 or x.shape[1] % 4 or torch.is_grad_enabled() and (
            x.requires_grad or weight.requires_grad):
        return False
    if weight.dtype != torch.float32:          # wgnn_linear_fwd computes and returns fp32: a .half() / .bfloat16() model keeps
        return False                           # F.linear's "output in the parameter dtype"
    if dual:
        return WGNN_LINEAR_DUAL
    if WGNN_LINEAR == "always":
        return True
    if x.dtype == torch.float16:
        return True                            # fp16-stored rows: widened in the kernel's loader, never materialised in fp32
    from . import tuning
    if tuning.active():
        # with the tracked per-shape picks loaded the LIBRARY wins the one fp32 shape this kernel used to take: 155 us (tuned
        # rocBLAS pick) vs 207 us here vs 232 us (the libraries' own heuristic) on 100k x 400 x 256 (round 4, scratch/tune_gemms.py)
        return False
    return x.shape[0] >= WGNN_LINEAR_MIN_ROWS and x.shape[1] >= WGNN_LINEAR_MIN_K


def linear_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
               row_scale: Optional[torch.Tensor] = None, tile_rows: Optional[int] = None):
    """``act(x @ weight.T + bias)`` with ``wgnn_linear_fwd_ex`` (v_mfma_f32_32x32x2_f32, exact fp32).  ``x`` may be stored
    in fp16 (widened in registers, no fp32 copy).  With ``row_scale`` ([M]) returns ``(out, row_scale[:, None] * out)``,
    both written by the one kernel.  ``tile_rows`` (64 | 128) overrides the kernel's own choice of tile height (timing
    experiments).  Inference helper: no autograd (training keeps torch's Linear, whose backward is a
    library GEMM as well)."""
    dev = _require_cuda(x, weight, bias, row_scale)
    if x.dtype not in (torch.float32, torch.float16):
        x = x.float()
    x = _rowmajor(x) if x.dtype == torch.float32 else (x if x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 8 == 0
                                                       else x.contiguous())
    weight = _rowmajor(weight.float())
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise WgnnError("x and weight disagree on K")
    if K % 4:
        raise WgnnError(f"K = {K} must be a multiple of 4")
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    out2 = None
    if row_scale is not None:
        row_scale = row_scale.reshape(-1).float().contiguous()
        if row_scale.shape[0] < M:
            raise WgnnError("row_scale needs one entry per row of x")
        out2 = torch.empty((M, N), dtype=torch.float32, device=dev)
    if bias is not None:
        bias = bias.float().contiguous()
    rc = _lib.call(dev, "wgnn_linear_fwd_ex", _ptr(x), _dtype_code(x), x.stride(0), _ptr(weight), weight.stride(0), _ptr(bias),
                   _ptr(out), out.stride(0), _ptr(row_scale), _ptr(out2), N if out2 is not None else 0, M, N, K,
                   (_lib.FLAG_RELU if relu else 0) | {None: 0, 64: 1 << 16, 128: 1 << 17}[tile_rows], _stream(dev))
    _lib.check(rc, "wgnn_linear_fwd_ex")
    return out if out2 is None else (out, out2)


WGRAD_MIN_ROWS = 16384       # from this many rows on, the weight gradient of a Linear runs through wgnn_linear_wgrad


def linear_wgrad(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """``dW = g.T @ x`` ([N, K]) with ``wgnn_linear_wgrad``: split along the row (node) axis, fp32 matrix cores, partial
    products folded in fixed order."""
    import ctypes as C
    dev = _require_cuda(g, x)
    g = _rowmajor(g.float()); x = _rowmajor(x.float())
    M, N = g.shape
    K = x.shape[1]
    if x.shape[0] != M or N % 4 or K % 4:
        raise WgnnError("linear_wgrad: g [M, N] and x [M, K] with N, K multiples of 4")
    ns, nb = C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().wgnn_linear_wgrad_workspace(M, N, K, C.addressof(ns), C.addressof(nb)), "wgnn_linear_wgrad_workspace")
    ws = torch.empty(max(1, nb.value // 4), dtype=torch.float32, device=dev)
    dW = torch.empty((N, K), dtype=torch.float32, device=dev)
    with _Timed(dev, ("rows", M, "N", N, "K", K, "kernel", "wgrad_mfma_f32 + wgrad_reduce")):
        rc = _lib.call(dev, "wgnn_linear_wgrad", _ptr(g), g.stride(0), _ptr(x), x.stride(0), _ptr(dW), K, M, N, K, 0, _ptr(ws),
                       ns.value, _stream(dev))
    _lib.check(rc, "wgnn_linear_wgrad")
    return dW


class _LinearBigM(torch.autograd.Function):
    """``F.linear`` whose weight gradient - a [N, K] product reduced over up to 1e5-1e6 node rows - runs through
    ``wgnn_linear_wgrad`` instead of the library GEMM the framework picks for that shape (0.9 ms = 22 TF at cfg3)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        dx = g @ weight if ctx.needs_input_grad[0] else None
        dw = linear_wgrad(g, x).to(weight.dtype) if ctx.needs_input_grad[1] else None
        db = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class _LinearReluBigM(torch.autograd.Function):
    """``relu(x W^T + b)`` of the aggregate-first order in TRAINING (gnn.py:65-66 under train.py:84): bias and ReLU ride in the
    library GEMM's epilogue as on the no-grad path; backward masks the upstream gradient and reduces the bias gradient in ONE
    ``wgnn_agg_bwd_prepare`` launch (instead of a threshold pass + a column-sum pass over [rows, H]), the weight gradient runs
    on ``wgnn_linear_wgrad``."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        try:
            out = torch._addmm_activation(bias, x, weight.t())
        except (RuntimeError, TypeError):                          # private torch entry point: fall back to the plain composition
            out = torch.relu(torch.nn.functional.linear(x, weight, bias))
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight, out = ctx.saved_tensors
        res = agg_bwd_prepare(gout, out, None, None, NO_ALPHA, 0, want_scaled=True, want_dbias=ctx.needs_input_grad[2])
        g = res["g_scaled"]                                           # gout * (out > 0)
        dx = g @ weight if ctx.needs_input_grad[0] else None
        dw = linear_wgrad(g, x).to(weight.dtype) if ctx.needs_input_grad[1] else None
        return dx, dw, res["dbias"]


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.linear`` for the model's projections; in training on many rows the weight gradient uses the matrix-core kernel;
    with nothing to differentiate, fp16-stored inputs and the shapes where it wins run through ``wgnn_linear_fwd_ex``."""
    if use_wgnn_linear(x, weight):
        return linear_fwd(x, weight, bias)
    if x.dtype != weight.dtype:
        x = x.to(weight.dtype)
    if (torch.is_grad_enabled() and weight.requires_grad and x.is_cuda and x.dim() == 2 and x.shape[0] >= WGRAD_MIN_ROWS
            and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.shape[0] % 4 == 0
            and weight.shape[1] % 4 == 0):
        return _LinearBigM.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


linear.widens_fp16 = True


def linear_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu: bool) -> torch.Tensor:
    """``act(x W^T + b)`` of the aggregate-first order (gnn.py:65-66: neigh -> fc_neigh -> activation).  With nothing to
    differentiate the bias and the ReLU ride in the library GEMM's epilogue (one launch, no [n, H] elementwise pass);
    otherwise the plain composition."""
    if (relu and bias is not None and not torch.is_grad_enabled() and x.is_cuda and x.dim() == 2
            and x.dtype == weight.dtype == bias.dtype and not use_wgnn_linear(x, weight)
            and hasattr(torch, "_addmm_activation")):
        try:                                       # a private torch entry point: any change of its contract falls back to
            return torch._addmm_activation(bias, x, weight.t())      # the plain composition below
        except (RuntimeError, TypeError):
            pass
    if (FUSED_BWD_GLUE and relu and bias is not None and torch.is_grad_enabled() and weight.requires_grad and x.is_cuda
            and x.dim() == 2 and x.shape[0] >= WGRAD_MIN_ROWS and x.dtype == weight.dtype == bias.dtype == torch.float32
            and weight.shape[0] % 4 == 0 and weight.shape[1] % 4 == 0 and weight.shape[0] <= 1024
            and hasattr(torch, "_addmm_activation")):
        return _LinearReluBigM.apply(x, weight, bias)
    out = linear(x, weight, bias)
    return torch.relu(out) if relu else out
