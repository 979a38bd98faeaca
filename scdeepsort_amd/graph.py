"""HBM-resident operand of the hot path: the bipartite cell-gene graph.

Reference counterpart: the DGLGraph assembled in
``utils/preprocess_internal.py:107-110,160-173,210-215`` (training graph) and
``utils/preprocess.py:102-134,184-187,212-221`` (predict graph): genes are nodes
``[0,G)``, cells follow; every expression value > threshold is a gene->cell edge
and (for support/training cells) a cell->gene edge; in-edge weights are
normalised per destination (``normalize_weight``, preprocess_internal.py:15-23)
and THEN a unit self-loop is added to every node.

Layout here (one copy per GPU, all device tensors):

* ``cg``  destination-major CSR of cells<-genes  (rows = cells,  col = gene id)
* ``gc``  destination-major CSR of genes<-cells  (rows = genes,  col = cell id)
* per direction: ``inv_deg = 1/(in_degree+1)`` (self-loop counted, implicit),
  an execution *plan* (row chunks, see ``wgnn_plan_build_host``) and - built
  lazily for training - the transposed structure with the same normalised
  values re-ordered (``normalize_weight`` is per destination, so the two
  directions carry different values and the backward of one direction is NOT
  the forward CSR of the other).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

DEFAULT_CHUNK = 2048
TRANSPOSE_CHUNK = 2048       # device-built source-major blocks: longest run of entries one wave reduces serially


def auto_chunk(nnz: int) -> int:
    """Row-chunk length of the row-wave kernels: short enough that the ~8k resident wavefronts of the chip all get
    work and the longest item does not dominate (cfg2: 256 is 2x faster than 2048), long enough that the partial-sum
    traffic stays small on big graphs.  Power of two in [256, 2048]."""
    c = 256
    while c < 2048 and c * 8192 < nnz:
        c *= 2
    return c


_EMPTY: dict = {}


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address for the C ABI.  An EMPTY tensor (e.g. col / val of a graph without a single edge) has a null
    data pointer, which the ABI would reject as a missing argument: it gets the address of a small per-device
    placeholder instead (never dereferenced - the row pointers say there is nothing to read)."""
    if t is None:
        return None
    if t.numel() == 0 and t.is_cuda:
        if t.device not in _EMPTY:
            _EMPTY[t.device] = torch.zeros(64, dtype=torch.int32, device=t.device)
        return _EMPTY[t.device].data_ptr()
    return t.data_ptr()


def _stream(device: torch.device) -> Optional[int]:
    return torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None


@dataclass
class Plan:
    """Row-chunk work list for one CSR (device copies + counts)."""
    items: torch.Tensor          # int32 [n_items, 4]  {row_slot, begin, end, partial_slot|-1}
    long_rows: torch.Tensor      # int32 [n_long, 4]   {row_slot, first_partial, n_partials, 0}
    n_partials: int
    chunk: int

    @property
    def n_items(self) -> int:
        return self.items.shape[0]

    @property
    def n_long(self) -> int:
        return self.long_rows.shape[0]


def build_plan(rowptr_host: np.ndarray, chunk: Optional[int] = None, row_ids_host: Optional[np.ndarray] = None,
               device: torch.device | str = "cpu") -> Plan:
    """Host-side plan construction through the C ABI (``wgnn_plan_build_host``); ``chunk=None`` -> ``auto_chunk``."""
    lib = _lib.lib()
    rowptr_host = np.ascontiguousarray(rowptr_host, dtype=np.int32)
    if chunk is None:
        if row_ids_host is not None:
            ids = np.asarray(row_ids_host, dtype=np.int64)
            chunk = auto_chunk(int((rowptr_host[ids + 1] - rowptr_host[ids]).sum()) if len(ids) else 0)
        else:
            chunk = auto_chunk(int(rowptr_host[-1]) if len(rowptr_host) else 0)
    n_rows = len(row_ids_host) if row_ids_host is not None else len(rowptr_host) - 1
    rid = None
    if row_ids_host is not None:
        rid = np.ascontiguousarray(row_ids_host, dtype=np.int32)
    ni, nl, npart = C.c_int64(), C.c_int64(), C.c_int64()
    rid_p = rid.ctypes.data if rid is not None else None
    _lib.check(lib.wgnn_plan_build_host(rowptr_host.ctypes.data, rid_p, n_rows, chunk, None, None,
                                        C.addressof(ni), C.addressof(nl), C.addressof(npart)), "wgnn_plan_build_host")
    items = np.empty((ni.value, 4), dtype=np.int32)
    longs = np.empty((nl.value, 4), dtype=np.int32)
    _lib.check(lib.wgnn_plan_build_host(rowptr_host.ctypes.data, rid_p, n_rows, chunk,
                                        items.ctypes.data, longs.ctypes.data,
                                        C.addressof(ni), C.addressof(nl), C.addressof(npart)), "wgnn_plan_build_host")
    return Plan(torch.from_numpy(items).to(device), torch.from_numpy(longs).to(device), int(npart.value), chunk)


def device_plan(rowptr32: torch.Tensor, n_rows: int, max_row_nnz: int, chunk: int, nnz_bound: Optional[int] = None) -> Plan:
    """Row-chunk plan built ON THE DEVICE with static shapes (no device->host copy).  ``max_row_nnz`` is a host-known BOUND
    on the row length, S = min(8, ceil(max_row_nnz / chunk)) the number of parts a long row is cut into (their partial
    sums are folded by ``agg_finalize``).

    Without ``nnz_bound`` EVERY row gets S items cut at multiples of the chunk length (later items of short rows are
    empty) and S partial rows.  With ``nnz_bound`` (a host-known bound on the number of entries) only rows LONGER than the
    chunk are cut: there are fewer than nnz_bound / chunk of them, so a static number H of hub slots suffices - every row
    keeps one item, hubs get S - 1 more (unused hub slots are items with row -1, which the kernels skip), and the partial
    buffer holds H * S rows instead of n_rows * S (a [G * 8, D] buffer written and re-read per backward of a
    sampled block where only the few hub genes need splitting)."""
    dev = rowptr32.device
    chunk = max(1, int(chunk), -(-int(max_row_nnz) // 8))     # at most 8 items per row, whatever the bound
    S = max(1, -(-int(max_row_nnz) // chunk))
    beg, ln = rowptr32[:-1], rowptr32[1:] - rowptr32[:-1]
    if nnz_bound is not None and S > 1:
        H = min(int(n_rows), int(nnz_bound) // chunk)         # rows longer than `chunk`: each holds > chunk of the <= nnz_bound entries
        rows = torch.arange(n_rows, device=dev, dtype=torch.int32)
        if H == 0:                                            # no row can exceed the chunk: one item per row
            items = torch.stack([rows, beg, beg + ln, torch.full_like(rows, -1)], 1).contiguous()
            return Plan(items, torch.empty((0, 4), dtype=torch.int32, device=dev), 0, chunk)
        is_long = ln > chunk
        rank = (torch.cumsum(is_long.to(torch.int32), 0) - 1).to(torch.int32)
        cl = (ln + (S - 1)) // S                              # part length of a long row (S near-equal parts)
        base = torch.stack([rows, beg, beg + torch.where(is_long, torch.minimum(cl, ln), ln),
                            torch.where(is_long, rank * S, torch.full_like(rows, -1))], 1)
        hub = torch.nonzero_static(is_long, size=H, fill_value=-1).squeeze(1)          # ascending row ids, -1 = unused slot
        hv = hub >= 0
        hr = hub.clamp(min=0)
        h_beg, h_ln, h_cl = beg[hr], ln[hr], cl[hr]
        j = torch.arange(1, S, device=dev, dtype=torch.int32).unsqueeze(0)             # parts 1 .. S-1
        lo = h_beg.unsqueeze(1) + torch.minimum(j * h_cl.unsqueeze(1), h_ln.unsqueeze(1))
        hi = h_beg.unsqueeze(1) + torch.minimum((j + 1) * h_cl.unsqueeze(1), h_ln.unsqueeze(1))
        hslot = torch.arange(H, device=dev, dtype=torch.int32).unsqueeze(1)
        row_e = torch.where(hv, hub.to(torch.int32), torch.full_like(hub, -1).to(torch.int32)).unsqueeze(1).expand(H, S - 1)
        extra = torch.stack([row_e, lo.to(torch.int32), hi.to(torch.int32), (hslot * S + j).expand(H, S - 1)], 2).reshape(H * (S - 1), 4)
        # unused hub slots -> {-1, 0, 0, -1}; built from device-side fills (no host->device copy: this runs inside captured steps)
        live = hv.repeat_interleave(S - 1)
        neg = torch.full_like(live, -1, dtype=torch.int32)
        zero = torch.zeros_like(neg)
        extra = torch.where(live.unsqueeze(1), extra, torch.stack([neg, zero, zero, neg], 1))
        items = torch.cat([base.to(torch.int32), extra.to(torch.int32)]).contiguous()
        long_rows = torch.stack([row_e[:, 0], hslot[:, 0] * S, torch.full((H,), S, dtype=torch.int32, device=dev),
                                 torch.zeros(H, dtype=torch.int32, device=dev)], 1).contiguous()
        return Plan(items, long_rows, H * S, chunk)
    s = torch.arange(S, device=dev, dtype=torch.int32).unsqueeze(0)
    lo = beg.unsqueeze(1) + torch.minimum(s * chunk, ln.unsqueeze(1))
    hi = beg.unsqueeze(1) + torch.minimum((s + 1) * chunk, ln.unsqueeze(1))
    slot = torch.arange(n_rows, device=dev, dtype=torch.int32).unsqueeze(1).expand(n_rows, S)
    if S > 1:
        pslot = slot * S + s
        long_rows = torch.stack([slot[:, 0], slot[:, 0] * S, torch.full_like(slot[:, 0], S), torch.zeros_like(slot[:, 0])], 1)
    else:
        pslot = torch.full_like(slot, -1)
        long_rows = torch.empty((0, 4), dtype=torch.int32, device=dev)
    items = torch.stack([slot, lo, hi, pslot], 2).reshape(n_rows * S, 4).contiguous()
    return Plan(items, long_rows.contiguous(), n_rows * S if S > 1 else 0, chunk)


@dataclass
class AggCsr:
    """One aggregation direction: destination-major CSR + normalised values + plan."""
    rowptr: torch.Tensor                 # int32 [R+1]
    col: torch.Tensor                    # int32 [nnz]
    val: torch.Tensor                    # float32 [nnz]   normalised weights
    inv_deg: torch.Tensor                # float32 [R]     1/(deg+1)
    n_rows: int
    n_cols: int
    plan: Plan
    rowptr_host: Optional[np.ndarray]    # None for blocks built on the device (sampler): then ``row_nnz_bound`` is set
    _t: Optional["AggCsr"] = field(default=None, repr=False)
    _tile_plan: Optional[object] = field(default=None, repr=False)
    # ELL-padded block drawn by the device sampler (sampler.sample_block_static): row i owns col/val[i*ell_k, i*ell_k +
    # ell_cnt[i]); rowptr holds the slot starts, the plan's items carry the real ranges.  Row-wave kernels only.
    ell_k: int = 0
    ell_cnt: Optional[torch.Tensor] = field(default=None, repr=False)
    # CUs the tile kernel may count on for a ONE-ROUND launch of this operand (see tile_plan): a rank of a sharded job whose
    # pass runs next to an in-flight collective leaves the communicator's workgroups their CUs (sharded.ShardedWgnn.build)
    cu_budget: int = 256

    @property
    def nnz(self) -> int:
        return self.col.shape[0]

    @property
    def device(self) -> torch.device:
        return self.col.device

    def transposed(self) -> "AggCsr":
        """Source-major copy: row s lists the destinations it feeds, values re-ordered (for K2)."""
        if self._t is None and self.ell_cnt is not None:
            # drawn block in ELL form: sort the valid entries by source behind a sentinel key (static shapes, no sync)
            dev, n, kk = self.device, self.n_rows, self.ell_k
            j = torch.arange(n * kk, device=dev)
            owner = torch.div(j, kk, rounding_mode="floor")
            valid = (j - owner * kk) < self.ell_cnt.long()[owner]
            key = torch.where(valid, self.col, torch.full_like(self.col, self.n_cols))       # int32 keys
            skey, perm = torch.sort(key, stable=True)
            t_rowptr32 = torch.searchsorted(skey, torch.arange(self.n_cols + 1, device=dev, dtype=torch.int32)).to(torch.int32)
            t_val = torch.where(valid[perm], self.val[perm], torch.zeros((), device=dev)).contiguous()
            # a source is drawn by at most every row once; a hub source (up to n entries) is cut into <= 8 items of >= 2048
            # entries whose partial sums agg_finalize folds - one wave per source would serialise the hub genes
            self._t = AggCsr(t_rowptr32, owner[perm].to(torch.int32).contiguous(), t_val, torch.empty(0, device=dev),
                             self.n_cols, n, device_plan(t_rowptr32, self.n_cols, max(1, n), TRANSPOSE_CHUNK, nnz_bound=n * kk), None)
            self._t._max_row_nnz = max(1, n)
        if self._t is None:
            dev = self.device
            if (CSR_TRANSPOSE_KERNEL and dev.type == "cuda" and 0 < self.n_cols <= 32768 and self.n_rows > 0 and self.ell_cnt is None
                    and self.rowptr.dtype == torch.int32 and self.col.dtype == torch.int32 and self.val.dtype == torch.float32):
                t_rowptr32, t_col, t_val = _transpose_on_device(self.rowptr, self.col, self.val, self.n_rows, self.n_cols, None)
            else:
                t_rowptr32, t_col, t_val = _transpose_by_sort(self.rowptr, self.col, self.val, self.n_rows, self.n_cols, None)
            if self.rowptr_host is None:             # device-built block: a source feeds at most every row once
                plan, host = device_plan(t_rowptr32, self.n_cols, max(1, self.n_rows), TRANSPOSE_CHUNK,
                                         nnz_bound=self.nnz), None                  # only hub sources are cut (<= 8 items each)
            else:
                host = t_rowptr32.cpu().numpy()
                plan = build_plan(host, self.plan.chunk, device=dev)
            self._t = AggCsr(t_rowptr32, t_col, t_val, torch.empty(0, device=dev), self.n_cols, self.n_rows, plan, host)
            self._t._max_row_nnz = self.n_rows if host is None else None
        return self._t

    def tile_plan(self, block_rows: int = 64, loaders: Optional[int] = None):
        """Lazily built plan for the LDS-streamed kernel (heuristic tile geometry, see build_tile_plan), cached per
        LDS block height and number of dedicated loader waves (``loaders=None``: the module default).  The backward entries
        K2t / K3t (``ops.agg_bwd_src`` / ``agg_bwd_alpha``) use the SAME default: they read the loader waves and shared pairs
        off the plan like the forward, and were measured that way (training step 10.7 -> 10.1 -> 9.9 ms in round 3).
        The tall geometry (opt-in, ``TILE_TALL``) exists for the flat kernel family only - every width the tile route serves
        (``ops.tiled_kernel_serves``: D <= 256); the library's A/B flag that forces the generic kernel answers a tall plan with
        WGNN_ERR_PLAN."""
        if self._tile_plan is None:
            self._tile_plan = {}
        tall = TILE_SHARED_PAIRS and (TILE_TALL == "on" or (TILE_TALL == "auto" and self.n_rows >= TALL_MIN_ROWS))
        geom = GEOM_TALL if tall else GEOM_FLAT
        key = (block_rows, TILE_LOADER_WAVES if loaders is None else loaders, TILE_SHARED_PAIRS, self.cu_budget, geom)
        if key not in self._tile_plan:
            tp = build_tile_plan(self, None, None, n_cus=self.cu_budget, block_rows=block_rows, n_loaders=key[1], geom=geom)
            if self.cu_budget < 256 and tp.n_tiles > self.cu_budget:
                # A tile workgroup takes a whole CU (16 waves x 128 VGPRs, 160 KB LDS).  In a ONE-round launch every CU held
                # by another kernel pushes a tile into a second round (+55 .. +65 % per pass next to 16 - 32 held CUs,
                # profiles/r04_issue_analysis.md section 10); a launch of several rounds re-balances by itself, and shrinking
                # its rounds would only cost the uncontended case - it keeps the full chip.
                tp = build_tile_plan(self, None, None, block_rows=block_rows, n_loaders=key[1], geom=geom)
            self._tile_plan[key] = tp
        return self._tile_plan[key]

    @property
    def max_row_nnz(self) -> int:
        """Longest row (host int; one device->host read, cached): sizes the static-shape buffers of seed blocks."""
        if getattr(self, "_max_row_nnz", None) is None:
            if self.rowptr_host is None:
                raise _lib.WgnnError("device-built block without a row-length bound")
            self._max_row_nnz = int(np.diff(self.rowptr_host).max()) if self.n_rows else 0
        return self._max_row_nnz

    def subplan(self, row_ids: torch.Tensor) -> Tuple[torch.Tensor, Plan]:
        """Plan restricted to a seed subset (one NodeFlow batch, train.py:71-81), built ON THE DEVICE with static
        shapes: no device->host copy, no host loop, no upload - a mini-batch step never synchronises and can be captured
        in a hipGraph.  Every seed row gets the same number S = ceil(longest row / chunk) of items, cut at the SAME chunk
        boundaries as the full-graph plan (later items of short rows are empty); S > 1 rows fold their partial sums in
        ``agg_finalize`` in chunk order, so a seed's result is bit-identical to its row of the full-graph pass whatever
        batch it arrives in.  Exception: an operand whose longest row needs more than 64 chunks (> 64 x plan.chunk
        non-zeros) cuts every seed row into 64 EQUAL parts instead - still deterministic and identical from batch to batch,
        but summed in a different order than the full-graph pass (equal to ~1 ulp of the partial sums, not bitwise;
        ``tests/test_gpu_parity.py::test_subplan_with_pathologically_long_rows``).  Not cached: the id tensor's contents
        change from batch to batch (a captured step re-runs this)."""
        dev = self.device
        ids32 = row_ids.to(device=dev, dtype=torch.int32).contiguous()
        B = ids32.shape[0]
        chunk = max(1, self.plan.chunk)
        S = max(1, -(-self.max_row_nnz // chunk))
        idl = ids32.long()
        beg = self.rowptr[idl]
        ln = self.rowptr[idl + 1] - beg
        if S > 64:                                                       # pathological row lengths: S equal chunks per row
            S = 64
            cl = (ln + (S - 1)) // S
        else:
            cl = torch.full_like(ln, chunk)
        s = torch.arange(S, device=dev, dtype=torch.int32).unsqueeze(0)  # [1, S]
        lo = beg.unsqueeze(1) + torch.minimum(s * cl.unsqueeze(1), ln.unsqueeze(1))
        hi = beg.unsqueeze(1) + torch.minimum((s + 1) * cl.unsqueeze(1), ln.unsqueeze(1))
        slot = torch.arange(B, device=dev, dtype=torch.int32).unsqueeze(1).expand(B, S)
        if S > 1:
            pslot = slot * S + s
            long_rows = torch.stack([slot[:, 0], slot[:, 0] * S, torch.full_like(slot[:, 0], S), torch.zeros_like(slot[:, 0])], 1)
        else:
            pslot = torch.full_like(slot, -1)
            long_rows = torch.empty((0, 4), dtype=torch.int32, device=dev)
        items = torch.stack([slot, lo.to(torch.int32), hi.to(torch.int32), pslot], 2).reshape(B * S, 4).contiguous()
        return ids32, Plan(items, long_rows.contiguous(), B * S if S > 1 else 0, 0)

    def seed_block_transposed(self, ids32: torch.Tensor):
        """Source-major view of the in-edges of the seed rows ``ids32`` (one NodeFlow block, train.py:71-81), for the
        backward of a mini-batch: (t_rowptr int32 [n_cols+1], t_slot int32 [cap], t_val f32 [cap], items int32 [n_cols,4]).
        Row s lists the seed SLOTS that gather source s, ascending (deterministic summation order, no atomics).
        Static shapes (cap = batch x longest row, padding sorted behind a sentinel key): no host synchronisation."""
        dev = self.device
        B = ids32.shape[0]
        if B == 0:                                                        # no seed: every source row is empty
            z = torch.zeros(self.n_cols + 1, dtype=torch.int32, device=dev)
            rows = torch.arange(self.n_cols, device=dev, dtype=torch.int32)
            return (z, torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev),
                    torch.stack([rows, z[:-1], z[1:], torch.full_like(rows, -1)], 1).contiguous())
        cap = max(1, B * max(1, self.max_row_nnz))
        idl = ids32.long()
        beg = self.rowptr[idl].long()
        ln = self.rowptr[idl + 1].long() - beg
        off = torch.cumsum(ln, 0)                                         # inclusive
        j = torch.arange(cap, device=dev)
        owner = torch.searchsorted(off, j, right=True).clamp_(max=max(B - 1, 0))
        valid = j < off[-1] if B else torch.zeros_like(j, dtype=torch.bool)
        pos = j - (off[owner] - ln[owner])
        eidx = torch.where(valid, beg[owner] + pos, torch.zeros_like(j))
        key = torch.where(valid, self.col[eidx], torch.full_like(self.col[eidx], self.n_cols))   # int32 keys: half the radix passes
        skey, perm = torch.sort(key, stable=True)                         # ties keep ascending slot order
        t_slot = owner[perm].to(torch.int32)
        t_val = torch.where(valid[perm], self.val[eidx[perm]], torch.zeros((), device=dev))
        t_rowptr = torch.searchsorted(skey, torch.arange(self.n_cols + 1, device=dev, dtype=torch.int32)).to(torch.int32)
        rows = torch.arange(self.n_cols, device=dev, dtype=torch.int32)
        items = torch.stack([rows, t_rowptr[:-1], t_rowptr[1:], torch.full_like(rows, -1)], 1).contiguous()
        return t_rowptr, t_slot.contiguous(), t_val.contiguous(), items


def sorted_columns(rowptr: torch.Tensor, col: torch.Tensor, raw: torch.Tensor, n_cols: int):
    """(col, raw) of the CSR with strictly ascending columns inside every row.  Already-sorted input (every producer in this
    package) costs three element-wise passes and one flag read; otherwise the non-zeros are sorted within their rows."""
    nnz = col.shape[0]
    if nnz < 2:
        return col, raw
    dev = col.device
    inside = torch.ones(nnz, dtype=torch.bool, device=dev)          # False at the first non-zero of a row
    starts = rowptr[:-1].long()
    inside[starts[starts < nnz]] = False
    if not bool(((col[1:] <= col[:-1]) & inside[1:]).any()):
        return col, raw
    counts = (rowptr[1:] - rowptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(rowptr.shape[0] - 1, device=dev), counts)
    key, perm = torch.sort(rows * int(n_cols) + col.long())
    if bool(((key[1:] == key[:-1])).any()):
        raise ValueError("expression CSR lists a (cell, gene) pair more than once")
    return col[perm].contiguous(), raw[perm].contiguous()


# The gene-major copy of the operand is built by the library's stable transpose (csrc/wgnn_transpose.hip: per-chunk LDS histograms +
# an in-order walk, no sort) when the operand lives on a GPU and the gene ids fit its counters; False = the framework path
# (bincount, stable radix sort, two gathers) - also what CPU operands and > 32768 genes take.
CSR_TRANSPOSE_KERNEL = __import__("os").environ.get("WGNN_CSR_TRANSPOSE_KERNEL", "1") == "1"


def _transpose_by_sort(rowptr, col, raw, n_rows: int, n_cols: int, row_mask: Optional[torch.Tensor]):
    """(t_rowptr int32 [n_cols + 1], t_col int32, t_raw f32): column-major copy, rows ascending inside a column, rows with
    ``row_mask == False`` dropped.  Framework primitives."""
    dev = col.device
    rows = torch.repeat_interleave(torch.arange(n_rows, device=dev, dtype=torch.int32), (rowptr[1:] - rowptr[:-1]).long())
    s_col, s_raw = col, raw
    if row_mask is not None:
        keep = row_mask.to(dev)[rows.long()]
        rows, s_col, s_raw = rows[keep], col[keep], raw[keep]
    counts = torch.bincount(s_col.long(), minlength=n_cols)
    t_rowptr = torch.zeros(n_cols + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=t_rowptr[1:])
    # radix sort by gene id: 16-bit keys when they fit (two 8-bit passes; 32-bit keys take four, 64-bit ones eight)
    order = torch.sort(s_col.to(torch.int16) if n_cols < 2 ** 15 else s_col, stable=True).indices
    return t_rowptr.to(torch.int32), rows[order].contiguous(), s_raw[order].contiguous()


def _transpose_on_device(rowptr, col, raw, n_rows: int, n_cols: int, row_mask: Optional[torch.Tensor]):
    """The same through ``wgnn_csr_transpose_count`` / ``_fill`` (rowptr int32, col int32 - a row lists a column once)."""
    import ctypes as C
    dev = col.device
    nch, nb = C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().wgnn_csr_transpose_workspace(n_rows, n_cols, C.addressof(nch), C.addressof(nb)), "wgnn_csr_transpose_workspace")
    counts = torch.empty(nb.value // 4, dtype=torch.int32, device=dev)
    t_count = torch.empty(n_cols, dtype=torch.int32, device=dev)
    keep = None if row_mask is None else row_mask.to(device=dev, dtype=torch.uint8).contiguous()
    if keep is not None and keep.shape[0] != n_rows:
        raise ValueError(f"support_mask has {keep.shape[0]} entries for {n_rows} cells")
    _lib.check(_lib.call(dev, "wgnn_csr_transpose_count", _ptr(rowptr), _ptr(col), _ptr(keep), n_rows, n_cols, nch.value,
                         _ptr(counts), _ptr(t_count), _stream(dev)), "wgnn_csr_transpose_count")
    t_rowptr = torch.zeros(n_cols + 1, dtype=torch.int32, device=dev)
    torch.cumsum(t_count, 0, out=t_rowptr[1:])
    nnz_t = col.shape[0] if keep is None else int(t_rowptr[-1])
    t_col = torch.empty(nnz_t, dtype=torch.int32, device=dev)
    t_raw = torch.empty(nnz_t, dtype=torch.float32, device=dev)
    _lib.check(_lib.call(dev, "wgnn_csr_transpose_fill", _ptr(rowptr), _ptr(col), _ptr(raw), _ptr(keep), n_rows, n_cols, nch.value,
                         _ptr(counts), _ptr(t_rowptr), _ptr(t_col), _ptr(t_raw), _stream(dev)), "wgnn_csr_transpose_fill")
    return t_rowptr, t_col, t_raw


def _normalize_on_device(rowptr: torch.Tensor, raw: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """K4 (``wgnn_normalize_rows``): w <- deg*w/sum(w) per destination, inv_deg = 1/(deg+1)."""
    if raw.device.type != "cuda":
        raise _lib.WgnnError("graph normalisation runs on the GPU (wgnn_normalize_rows); no CPU fallback")
    n_rows = rowptr.shape[0] - 1
    out = torch.empty_like(raw)
    inv_deg = torch.empty(n_rows, dtype=torch.float32, device=raw.device)
    _lib.check(_lib.call(raw.device, "wgnn_normalize_rows", _ptr(rowptr), _ptr(raw), _ptr(out), _ptr(inv_deg), n_rows,
                         _stream(raw.device)), "wgnn_normalize_rows")
    return out, inv_deg


@dataclass
class CellGeneGraph:
    num_genes: int
    num_cells: int
    cg: AggCsr          # cells <- genes
    gc: AggCsr          # genes <- cells   (support cells only; test cells feed nothing back, preprocess.py:184-187)
    cell_offset: int = 0     # first global cell id held by this shard (multi-GPU)
    num_cells_global: int = 0

    @property
    def device(self) -> torch.device:
        return self.cg.device

    @property
    def num_nodes(self) -> int:
        return self.num_genes + self.num_cells

    @staticmethod
    def from_expression(expr, support_mask: Optional[np.ndarray] = None, device: torch.device | str = "cuda",
                        chunk: Optional[int] = None) -> "CellGeneGraph":
        """Build from a scipy (cells x genes) matrix of raw expression values (entries > threshold only).

        Mirrors the reference build order: both edge directions from the same raw value
        (preprocess_internal.py:170-173), per-destination normalisation (:211), implicit self-loops (:213).
        """
        import scipy.sparse as sp
        device = torch.device(device)
        x = sp.csr_matrix(expr).astype(np.float32)
        x.sort_indices()
        if x.indptr[-1] >= 2 ** 31:
            raise ValueError("nnz >= 2^31: shard the cell axis (sharded.ShardedWgnn)")
        mask = None if support_mask is None else torch.from_numpy(np.asarray(support_mask, dtype=bool)).to(device)
        # one upload of the CSR; the gene-major copy (transpose), both normalisations and the plans are built on the device
        return CellGeneGraph.from_device_csr(torch.from_numpy(x.indptr.astype(np.int64)).to(device),
                                             torch.from_numpy(x.indices.astype(np.int32)).to(device),
                                             torch.from_numpy(x.data).to(device), x.shape[1], chunk, mask)

    @staticmethod
    def from_device_csr(rowptr: torch.Tensor, col: torch.Tensor, raw: torch.Tensor, num_genes: int,
                        chunk: Optional[int] = None, support_mask: Optional[torch.Tensor] = None,
                        check_sorted: bool = True) -> "CellGeneGraph":
        """Build from a device CSR of the (cells x genes) raw expression.  ``support_mask`` (bool [cells], default all
        True): only support cells feed the genes; test cells of a predict graph get gene->cell edges only
        (preprocess.py:126-134,184-187).

        Precondition of the tile plans (``build_tile_plan`` -> ``wgnn_tile_plan_count`` / ``_fill`` walk a row's non-zeros
        in column order): gene ids ASCENDING and unique inside every cell.  Checked here on the device
        (``sorted_columns``): an unsorted CSR is sorted within its rows, a repeated (cell, gene) pair is an error - the
        reference adds one edge pair per non-zero of the expression matrix (preprocess_internal.py:158-173).
        ``check_sorted=False`` skips the check for producers that guarantee the order."""
        dev = col.device
        C_ = rowptr.shape[0] - 1
        if col.shape[0] >= 2 ** 31:                        # the operand's offsets are int32 (same guard as from_expression)
            raise ValueError("nnz >= 2^31: shard the cell axis (sharded.ShardedWgnn)")
        if raw.shape[0] != col.shape[0]:
            raise ValueError(f"col has {col.shape[0]} entries, raw {raw.shape[0]}")
        rowptr = rowptr.to(torch.int32).contiguous()
        col = col.to(torch.int32).contiguous()
        raw = raw.to(torch.float32).contiguous()
        if check_sorted:
            col, raw = sorted_columns(rowptr, col, raw, num_genes)
        val, inv_deg = _normalize_on_device(rowptr, raw)
        host = rowptr.cpu().numpy()
        cg = AggCsr(rowptr, col, val, inv_deg, C_, num_genes, build_plan(host, chunk, device=dev), host)
        # transpose the RAW values (support cells only), then normalise per gene
        if CSR_TRANSPOSE_KERNEL and dev.type == "cuda" and 0 < num_genes <= 32768 and C_ > 0:
            t_rowptr, t_col, t_raw = _transpose_on_device(rowptr, col, raw, C_, num_genes, support_mask)
        else:
            t_rowptr, t_col, t_raw = _transpose_by_sort(rowptr, col, raw, C_, num_genes, support_mask)
        t_val, t_inv = _normalize_on_device(t_rowptr, t_raw)
        thost = t_rowptr.cpu().numpy()
        gc = AggCsr(t_rowptr, t_col, t_val, t_inv, num_genes, C_, build_plan(thost, chunk, device=dev), thost)
        return CellGeneGraph(num_genes, C_, cg, gc, 0, C_)

    @staticmethod
    def cell_features(rowptr: torch.Tensor, col: torch.Tensor, raw: torch.Tensor, gene_feat: torch.Tensor,
                      chunk: Optional[int] = None) -> torch.Tensor:
        """``cell_feat = rownorm(X) . gene_feat`` with ``rownorm(X)[c,g] = x/(sum_g x + 1e-6)``
        (preprocess_internal.py:197-199, preprocess.py:205-207) as the same weighted SpMM the hot path uses (K1,
        NO_ALPHA, no mean, no self-loop) instead of the reference's dense (cells x genes) matrix product."""
        from .ops import agg_fwd
        from ._lib import NO_ALPHA
        dev = col.device
        C_ = rowptr.shape[0] - 1
        rowptr32 = rowptr.to(torch.int32).contiguous()
        nnz = (rowptr32[1:] - rowptr32[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(C_, device=dev), nnz)
        rs = torch.zeros(C_, dtype=torch.float64, device=dev).index_add_(0, rows, raw.double())
        val = (raw.double() / (rs[rows] + 1e-6)).float().contiguous()
        host = rowptr32.cpu().numpy()
        csr = AggCsr(rowptr32, col.to(torch.int32).contiguous(), val, torch.ones(C_, device=dev), C_, gene_feat.shape[0],
                     build_plan(host, chunk, device=dev), host)
        gf = gene_feat.float().contiguous()
        D = gf.shape[1]
        pad = (-D) % 4
        if pad:
            gf = torch.nn.functional.pad(gf, (0, pad))
        out = agg_fwd(csr, None, NO_ALPHA, 0, gf, None, no_mean=True)
        return out[:, :D] if pad else out

    def bytes_resident(self) -> int:
        tot = 0
        for d in (self.cg, self.gc):
            for t in (d.rowptr, d.col, d.val, d.inv_deg, d.plan.items, d.plan.long_rows):
                tot += t.numel() * t.element_size()
        return tot


# ------------------------------------------------------------------------------------------------
# tile plans for the LDS-streamed kernel (wgnn_agg_fwd_tiled)
# ------------------------------------------------------------------------------------------------
TILE_ROWS = 256          # 16 waves x 16 rows (kTW x kRPW in csrc/wgnn_tiled.hip)
TILE_WAVES = 16


@dataclass(frozen=True)
class TileGeom:
    """Tile geometry of a plan: waves per tile workgroup x destination rows per wave (csrc/wgnn_tiled.hip)."""
    waves: int
    rpw: int
    tall: bool = False

    @property
    def rows(self) -> int:
        return self.waves * self.rpw


GEOM_FLAT = TileGeom(TILE_WAVES, TILE_ROWS // TILE_WAVES)      # agg_tiled_flat4 / agg_tiled: 16 waves x 16 rows, 128 VGPRs per wave
GEOM_TALL = TileGeom(8, 49, True)                              # agg_tiled_tall (round 5): 8 waves x 49 rows, 256 VGPRs per wave
# Which operands get the tall tile: "off" (default) never; "auto" many-row operands without a column split - the cell side of a
# graph, the transposed gene side in training; "on" every operand.  One 392-row tile per CU covers cfg3's 100 000 rows in ONE
# round (the table is streamed once per CU instead of twice, ~150 entries per (wave, block) instead of ~47).  MEASURED SLOWER
# than the 16-wave kernel at cfg3 (profiles/r05_issue_analysis.md: cells<-genes 1.17 vs 1.06 ms): its entry pipeline is on par
# (0.985 vs 1.01 ms without the table stream), but with 8 waves x 256 VGPRs there is no wave to spare for a dedicated loader and
# the DMA issued by the computing waves costs +0.18 ms.  Kept opt-in (parity-tested, register contract audited) as the base for a
# tall tile WITH a loader.
TILE_TALL = __import__("os").environ.get("WGNN_TILE_TALL", "off")
TALL_MIN_ROWS = 40_000
# Dedicated loader waves of the flat tile kernel (round 3): the first L waves of a tile own no destination rows and issue the
# whole global->LDS stream of the steady-state blocks; the other 16 - L waves only compute.  Used when a tile's rows fit the
# remaining 16 x (16 - L) accumulator slots (cfg3's cell side: 195 rows per tile = 15 waves x 13 rows at L = 1; the gene side's
# 243-row tiles are brought under 240 rows, see build_tile_plan).  Which wave streams does not change a row's summation order (with slot-sorted entries the results
# are the same bit for bit; shared pairs order a row's entries by its wave-mates, so plans then agree to rounding).  First form:
# cfg3 cells<-genes 1.28 -> 1.20 ms on one box, 1.29 -> 1.15 on another; with the lean piece loop L = 1 is the default.
# 0 = off.
TILE_LOADER_WAVES = 1
# Shared pairs: entries of a (wave, block) segment that read the same source row are consumed two per LDS read
# (_pair_segment_entries; 55 % of the entries at cells<-genes, 43 % at genes<-cells of the headline graph).
TILE_SHARED_PAIRS = True
TILE_PAIR_FLAG = -(1 << 31)      # int32 sign bit of an entry's meta word: member of a shared pair
TILE_PAD_FLAG = 1 << 30          # a zero-weight filler that keeps the pairs at even offsets
# Few-row operands (one round of column-split tiles): entries per (computing wave, LDS block) from which L loader waves pay
# (profiles/r03_issue_analysis.md: cfg3 node counts, density 0.5 .. 8 %, lean loader loop): one loader wave on the gene side wins
# from ~30 (0.88 vs 0.97 ms at 37; 0.87 vs 0.79 at 24 - its stream has a floor of ~0.85 ms).  Many-row operands (no column
# split) gain at every density measured (5 .. 87 entries per wave and block) and are not guarded.
LOADER_MIN_ENTRIES = {1: 30.0, 2: 16.0, 3: 16.0}
VIRTUAL_ROW_SHARE = 0.5   # few-row operands: a row heavier than this share of an average wave's load is dealt as virtual rows


def _pair_segment_entries(seg: torch.Tensor, meta: torch.Tensor, val_bits: torch.Tensor, n_seg: int, slot_mask: int = 0xF):
    """Entries of every (tile, block, wave) segment, ordered for the flat kernel's shared-pair stream.

    Two entries of a segment that read the SAME source row (one gene drawn by two of the wave's 16 cells; one cell
    expressing two of the wave's genes) form a shared pair: the kernel stages that LDS row once for both.  A segment is
    laid out as [unshared entries][pad, iff the entry count is odd][shared pairs], so that a pair always starts at an even offset
    (chunks of 64 never cut one) and the pairs are the LAST pair steps of a right-aligned chunk.  Meta word:
        unshared / second of a pair : slot << 8 | src_local                    (second: | TILE_PAIR_FLAG)
        first of a pair             : TILE_PAIR_FLAG | slot_of_second << 16 | slot << 8 | src_local
        pad                         : TILE_PAD_FLAG | the entry before it, weight 0
    Returns (entries int32 [n, 2], seg_ptr int64 [n_seg + 1])."""
    dev = seg.device
    n = seg.shape[0]
    # Round 5: ONE sort instead of two, on 32-bit keys where they fit (a 64-bit stable sort of 8e7 keys is most of a plan's build
    # time).  The second sort ("unshared first, pairs stay adjacent") is replaced by arithmetic: an entry's position inside its
    # segment is the number of unshared (or shared) entries in front of it in (segment, source row) order - two running sums.
    small = n_seg * 256 < 2 ** 31 - 1
    kdt = torch.int32 if small else torch.int64
    src_local = (meta & 0xFF).to(kdt)
    key1 = seg.to(kdt) * 256 + src_local
    del src_local
    perm = torch.sort(key1, stable=True).indices        # group by (segment, source row), CSR order inside
    gkey = key1[perm]
    del key1
    meta_s, val_s = meta.to(torch.int32)[perm], val_bits[perm]
    del perm
    seg_s = torch.div(gkey, 256, rounding_mode="floor")
    idt = torch.int32 if n < 2 ** 31 - 1 else torch.int64
    ar = torch.arange(n, device=dev, dtype=idt)
    new_group = torch.ones(n, dtype=torch.bool, device=dev)
    new_group[1:] = gkey[1:] != gkey[:-1]
    del gkey
    gid = torch.cumsum(new_group, 0, dtype=idt) - 1
    gstart = ar[new_group]
    gsize = torch.diff(gstart, append=torch.tensor([n], device=dev, dtype=idt))
    idx_in = ar - gstart[gid]
    shared = idx_in < (gsize[gid] // 2) * 2
    first = shared & (idx_in % 2 == 0)
    del gid, gstart, gsize, new_group, idx_in
    # the second entry's slot rides in the first entry's word
    nxt_slot = torch.zeros_like(meta_s)
    nxt_slot[:-1] = (meta_s[1:] >> 8) & slot_mask
    meta_s = torch.where(first, meta_s | (nxt_slot << 16), meta_s)
    del nxt_slot, first
    meta_s = torch.where(shared, meta_s | torch.tensor(TILE_PAIR_FLAG, dtype=torch.int32, device=dev), meta_s)
    # per-segment counts WITHOUT a histogram (an atomic histogram of 8e7 keys costs 3.5 ms, twice): the entries are sorted by
    # segment, so a segment is the index range between two binary searches, and its number of shared entries is a difference of
    # the running sum below
    bounds = torch.searchsorted(seg_s, torch.arange(n_seg + 1, device=dev, dtype=seg_s.dtype))          # int64 [n_seg + 1]
    n_all = bounds[1:] - bounds[:-1]
    sh = shared.to(idt)
    cs_sh = torch.zeros(n + 1, dtype=idt, device=dev)
    torch.cumsum(sh, 0, out=cs_sh[1:])                                    # cs_sh[i] = shared entries among the first i
    del sh
    base_sh = cs_sh[bounds[:-1]].long()
    n_sh = cs_sh[bounds[1:]].long() - base_sh
    pad = (n_all - n_sh) & 1                                              # an odd segment is padded to even (ABI 0.2.5; the pair
                                                                          # count is even, so this is n_all & 1: a segment's size
                                                                          # follows from its entry count alone - csrc/wgnn_plan.hip)
    seg_ptr = torch.zeros(n_seg + 1, dtype=torch.int64, device=dev)
    torch.cumsum(n_all + pad, 0, out=seg_ptr[1:])
    # position = seg_ptr[s] + (unshared: entries of s in front of me that are unshared | shared: all unshared of s + pad + shared
    # ones in front of me), folded into ONE per-segment offset for either kind (two gathers by segment instead of four):
    #   shared   : A[s] + c      A = seg_ptr + n_unshared + pad - base_sh      (c = shared entries among ALL entries in front of me)
    #   unshared : B[s] + i - c  B = seg_ptr - bounds + base_sh                (i = my index in sorted order)
    A = seg_ptr[:-1] + (n_all - n_sh + pad) - base_sh
    B = seg_ptr[:-1] - bounds[:-1] + base_sh
    seg_l = seg_s.long()
    c = cs_sh[:-1].long()
    del cs_sh
    pos = torch.where(shared, A[seg_l] + c, B[seg_l] + (ar.long() - c))
    del c, seg_l, A, B
    total = int(seg_ptr[-1])
    entries = torch.zeros((total, 2), dtype=torch.int32, device=dev)
    entries[pos] = torch.stack([meta_s, val_s], 1)                        # one 8-byte row scatter
    del pos
    pseg = torch.nonzero(pad).squeeze(1)
    if pseg.numel():
        ppos = seg_ptr[pseg] + (n_all - n_sh)[pseg]                       # right behind the (odd number of) unshared entries
        entries[ppos, 0] = (entries[ppos - 1, 0] & 0xFFFF) | TILE_PAD_FLAG
    return entries, seg_ptr


@dataclass
class TilePlan:
    items: torch.Tensor        # int32 [n_tiles, 256, 4]
    hdr: torch.Tensor          # int32 [n_tiles, 2]
    long_rows: torch.Tensor    # int32 [n_long, 4]
    n_partials: int
    n_row_tiles: int
    n_col_splits: int
    n_loaders: int = 0                         # waves 0..n_loaders-1 of every tile own no rows (dedicated loader waves)
    entries: Optional[torch.Tensor] = None     # int32 [nnz, 2]  {dst_slot<<8 | src_local, weight bits}
    seg_ptr: Optional[torch.Tensor] = None     # int32 [n_tiles*nblk_max*16 + 1]
    nblk_max: int = 0
    block_rows: int = 64
    geom: TileGeom = GEOM_FLAT

    @property
    def n_tiles(self) -> int:
        return self.items.shape[0]

    @property
    def block_rows_arg(self) -> int:
        """The C ABI's ``block_rows`` argument: rows per LDS block | plan geometry bits (WGNN_PLAN_TALL)."""
        return self.block_rows | (_lib.PLAN_TALL if self.geom.tall else 0)


def _snake(p: torch.Tensor, n: int) -> torch.Tensor:
    """Boustrophedon dealing: position p of a descending-sorted list -> bin, so that bins get balanced sums."""
    r, q = p // n, p % n
    return torch.where(r % 2 == 0, q, n - 1 - q)


ONE_ROUND_MIN_NNZ = 0               # one round wins or ties at every size measured: 2.0 M edges 62 vs 75 us, 11.9 M even, 39.8 M -20 %, 79.8 M -14 %, 159.8 M -13 % (profiles/r02_issue_analysis.md, r04_issue_analysis.md §4)


def auto_tile_geometry(n_rows: int, n_cols: int, n_cus: int = 256, nnz: Optional[int] = None,
                       rows_cap: int = 256, tile_rows: int = TILE_ROWS) -> Tuple[int, int]:
    """(n_row_tiles, n_col_splits), from sweeps at 10k..100k cells (profiles/r01_issue_analysis.md, r02_issue_analysis.md):
    * many rows (>= 40k): whole rounds of ~195..256-row tiles over the CUs, no column split;
    * fewer rows (the gene side; the cell side of small graphs): ~250-row tiles, and the source axis split so that
      hub rows spread over several workgroups and the launch has ~nnz/50k tiles (between 160 and five full rounds)."""
    min_tiles = -(-n_rows // tile_rows)
    if n_rows >= 40_000:
        # `rows_cap` < 256 when dedicated loader waves are on: a tile's rows must fit the computing waves' accumulators
        # (cfg3: 447 -> 512 tiles either way; cfg5's 764,741 rows: 14 rounds of 213-row tiles instead of 12 of 249)
        min_tiles = -(-n_rows // max(1, min(tile_rows, rows_cap)))
        return -(-min_tiles // n_cus) * n_cus, 1
    full = tile_rows * 250 // 256                         # ~98 % of a tile: 250 of 256 rows, 390 of 400
    n_row_tiles = max(1, -(-n_rows // full))
    if nnz is not None and nnz >= ONE_ROUND_MIN_NNZ and n_row_tiles <= n_cus:
        # big operands: ONE round of <= n_cus workgroups.  Every extra column split costs one more partial-sum row per
        # destination row (written, then re-read by agg_finalize); at cfg3 the genes<-cells pass went from 80 x 16
        # tiles / 335 MB of partial sums / 1.42 ms to 85 x 3 / 63 MB / 1.22 ms (profiles/r02_issue_analysis.md).
        def one_round(rows_per_tile):
            rt = max(1, -(-n_rows // rows_per_tile))
            sp = max(1, min(n_cus // rt, n_cols // 512 or 1))
            # fill the round with slightly smaller row tiles (82 -> 85 x 3 = 255 tiles), never by shredding a small operand
            return max(rt, min(n_cus // sp, -(-rt * 115 // 100))), sp
        plain = one_round(full)
        if rows_cap < full:
            # tiles that leave room for loader waves (<= rows_cap rows) - taken when they still fill the round (a 12.5k-row
            # shard: 64 x 4 = 256 tiles of 195 rows instead of 51 x 5 of 245; cfg3's gene side would drop to 105 x 2 = 210
            # tiles of 21 % more work each and keeps 85 x 3)
            lean = one_round(rows_cap)
            if lean[0] * lean[1] >= 0.95 * plain[0] * plain[1] and lean[1] <= plain[1]:
                return lean
        return plain
    target = 5 * n_cus if nnz is None else min(5 * n_cus, max(160, nnz // 50_000))
    splits = max(1, min(round(target / n_row_tiles), 5 * n_cus // n_row_tiles, n_cols // 512 or 1))
    return n_row_tiles, splits


TILE_ORDER = "xcd"       # "xcd" | "split_major" (round-1 order; kept for A/B timing)
# Round 5: the entries of a plan are built by the library's plan walk (csrc/wgnn_plan.hip: one wavefront per (tile, wave), no
# sort) when the operand lives on a GPU; False (or a CPU operand - the unit tests) = the framework index arithmetic below.
TILE_PLAN_KERNEL = __import__("os").environ.get("WGNN_TILE_PLAN_KERNEL", "1") == "1"


def _flat_tile_index(n_col_splits: int, n_row_tiles: int, dev) -> torch.Tensor:
    """[K, R] -> launch position (blockIdx.x) of tile (column split k, row tile t).  Workgroup b runs on XCD b % 8 and
    every XCD has its own 4 MiB L2, so tiles that stream the SAME source range should share an XCD and run at the same
    time.  The split-major tile list (k*R + t) is cut into 8 contiguous chunks, one per XCD, and XCD x's j-th tile is
    launched at b = 8*j + x: every XCD works through at most ceil(K/8)+1 source ranges, in order.  The round-1 order
    (b = k*R + t) spread every split over all 8 XCDs, so each L2 fetched the whole source table: 8 x 102 MB per
    genes<-cells pass at cfg3 instead of ~1.4 x."""
    K, R = n_col_splits, n_row_tiles
    k = torch.arange(K, device=dev).unsqueeze(1).expand(K, R)
    t = torch.arange(R, device=dev).unsqueeze(0).expand(K, R)
    p = k * R + t
    if TILE_ORDER == "split_major":
        return p
    chunk = -(-(K * R) // 8)
    flat = (p % chunk) * 8 + p // chunk                     # holes when 8 does not divide K*R -> compact by rank
    order = torch.argsort(flat.reshape(-1))
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=dev)
    return rank.reshape(K, R)


def build_tile_plan(csr: AggCsr, n_row_tiles: Optional[int] = None, n_col_splits: Optional[int] = 1,
                    n_cus: int = 256, block_rows: int = 64, balance: bool = True, n_loaders: int = 0,
                    geom: TileGeom = GEOM_FLAT) -> TilePlan:
    """Group the rows of ``csr`` into tiles of <= 256 rows (nnz-balanced across tiles and across the 16
    waves of a tile) and optionally split the column (source) range so hub rows spread over several
    workgroups.  Pure index arithmetic on the device; runs once per graph.

    ``balance`` (few-row operands only, i.e. the gene side): a row heavier than half the average wave's share (a hub gene) would make its wave the straggler at
    every per-block barrier, so it is dealt as k "virtual rows" - its non-zeros round-robin, i.e. evenly inside every
    source block - that land in different waves (and tiles); each writes a partial sum that ``agg_finalize`` folds in
    fixed order, exactly like the partial sums of column splits.

    Precondition: the columns of every row of ``csr`` ascend (the device plan walk binary-searches a row and takes wave minima
    of the pending columns).  ``CellGeneGraph.from_device_csr`` establishes it for caller-supplied CSRs (``sorted_columns``);
    the transposes, sampled blocks and shards built in this package keep it by construction.

    ``n_col_splits=None``: heuristic geometry (``auto_tile_geometry`` on the number of virtual rows).
    ``geom``: GEOM_FLAT (16 waves x 16 rows) or GEOM_TALL (8 waves x 49 rows, no dedicated loader waves)."""
    dev = csr.device
    TILE_ROWS, TILE_WAVES = geom.rows, geom.waves         # (shadow the module constants: everything below is per geometry)
    if geom.tall:
        n_loaders = 0                                     # 8 waves: none to spare, every wave streams its share
    R, S = csr.n_rows, csr.n_cols
    nnz = (csr.rowptr[1:] - csr.rowptr[:-1]).long()
    total = int(nnz.sum())
    # ---- virtual rows (+ heuristic geometry)
    n_loaders = int(n_loaders or 0)
    few_rows = balance and 0 < R < 40_000 and total > 0    # many-row operands: every row is a small fraction of a tile
    rpw = TILE_ROWS // TILE_WAVES
    auto_geom = n_col_splits is None
    if few_rows and n_loaders and auto_geom:
        # Few-row operands run ONE round of ~250-row tiles.  Tried in this order: the requested loader waves as they are (a
        # 25k-row shard's 214-row tiles hold two); then ONE loader wave, which fits when the tile holds <= 240 rows - cfg3's
        # gene side has 20 600 virtual rows for 85 tiles (243 each), and letting a row grow a little heavier before it is
        # dealt as virtual rows brings that under 85 x 240.  One loader wave is enough there because that stream is shorter
        # (8.7 GB; at the cell side's 10.5 GB one loader is the limit): genes<-cells 1.145-1.16 -> 1.11 ms.
        candidates = [(n_loaders, VIRTUAL_ROW_SHARE)] + [(1, sh) for sh in (VIRTUAL_ROW_SHARE, 0.65, 0.8, 1.0, 1.3)]
    else:
        candidates = [(n_loaders, VIRTUAL_ROW_SHARE)]
    candidates.append((0 if few_rows else n_loaders, VIRTUAL_ROW_SHARE))    # nothing fitted: plain plan, every wave streams
    req_tiles, req_splits = n_row_tiles, n_col_splits
    for attempt, (n_loaders, share) in enumerate(candidates):
        last = attempt == len(candidates) - 1
        n_row_tiles, n_col_splits = req_tiles, req_splits
        k_r = torch.ones(R, dtype=torch.int64, device=dev)
        if few_rows:
            tiles_guess = max(n_row_tiles or 0, -(-R // TILE_ROWS), 1)
            cap = max(64.0, share * total / (tiles_guess * TILE_WAVES))      # (half) the average wave's share of a tile
            k_r = torch.clamp(torch.ceil(nnz.double() / cap).long(), 1, 16)
        vbase = torch.zeros(R + 1, dtype=torch.int64, device=dev)
        torch.cumsum(k_r, 0, out=vbase[1:])
        V = int(vbase[-1])                                                  # number of virtual rows
        min_tiles = -(-V // TILE_ROWS)
        if auto_geom:
            n_row_tiles, n_col_splits = auto_tile_geometry(V, S, n_cus, total, rpw * (TILE_WAVES - n_loaders), TILE_ROWS)
        if not few_rows or last or n_loaders == 0:
            break
        rt = max(n_row_tiles if n_row_tiles is not None else 0, min_tiles, 1)
        if -(-V // rt) <= rpw * (TILE_WAVES - n_loaders):
            break
    vrow = torch.repeat_interleave(torch.arange(R, device=dev), k_r)        # virtual row -> row
    vpart = torch.arange(V, device=dev) - vbase[vrow]
    vnnz = nnz[vrow] // k_r[vrow] + (vpart < nnz[vrow] % k_r[vrow]).long()  # round-robin share of the row's non-zeros
    if n_row_tiles is None:
        per = max(1, n_cus // max(1, n_col_splits))
        n_row_tiles = -(-min_tiles // per) * per                 # whole number of rounds over the CUs
    n_row_tiles = max(n_row_tiles, min_tiles)
    order = torch.sort(vnnz, descending=True, stable=True).indices          # virtual row ids, longest first
    p = torch.arange(V, device=dev)
    tile = _snake(p, n_row_tiles)
    rnd = p // n_row_tiles                                                  # local index inside the tile (desc. nnz)
    if V and int(rnd.max()) >= TILE_ROWS:
        raise ValueError("tile overflow")
    wave = _snake(rnd, TILE_WAVES)
    slot_in_wave = rnd // TILE_WAVES
    rows_per_tile = -(-V // n_row_tiles) if V else 0
    if n_loaders and auto_geom and few_rows:             # (an explicit geometry takes the requested loader count as given)
        # On a one-round operand a lone loader wave has a floor (its ~4 k clk per block of 78 pieces x the blocks of ONE tile):
        # with few entries per wave and block all 16 waves streaming 5 pieces each is faster.
        nblk_est = max(1, -(-(-(-S // n_col_splits)) // block_rows))
        per_wave_block = total / max(1, n_row_tiles * n_col_splits * nblk_est * (TILE_WAVES - n_loaders))
        if per_wave_block < LOADER_MIN_ENTRIES.get(n_loaders, 1e30):
            n_loaders = 0
    if n_loaders and rows_per_tile <= rpw * (TILE_WAVES - n_loaders):
        # dedicated loader waves: waves 0..L-1 of every tile get no rows (they issue the tile's whole global->LDS stream);
        # the rows are dealt in snake order over the computing waves.  Measured and dropped (profiles/r03_issue_analysis.md):
        # equal edge shares per SIMD instead of per wave, a few light rows on the loader waves, loader waves at s_setprio 3.
        cw = TILE_WAVES - n_loaders
        wave = n_loaders + _snake(rnd, cw)
        slot_in_wave = rnd // cw
    else:
        n_loaders = 0                                                        # does not fit: every wave computes and streams
    local = wave * (TILE_ROWS // TILE_WAVES) + slot_in_wave
    # column splits on block boundaries
    blk = block_rows
    per_split = -(-(-(-S // blk)) // n_col_splits) * blk
    bounds = torch.arange(n_col_splits + 1, device=dev, dtype=torch.int64) * per_split
    bounds[-1] = S
    bounds = bounds.clamp(max=S)
    flat_of = _flat_tile_index(n_col_splits, n_row_tiles, dev)            # [K, R] -> blockIdx.x
    n_flat = n_col_splits * n_row_tiles
    items = torch.full((n_flat, TILE_ROWS, 4), -1, dtype=torch.int32, device=dev)
    items[..., 1:3] = 0
    # partial-sum slots: a row needs them when it is split along the columns or into virtual rows
    n_parts_r = k_r * n_col_splits
    needs = n_parts_r > 1
    pbase = torch.zeros(R + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.where(needs, n_parts_r, torch.zeros_like(n_parts_r)), 0, out=pbase[1:])
    n_part = int(pbase[-1])
    o_row, o_part = vrow[order], vpart[order]
    for k in range(n_col_splits):
        f = flat_of[k][tile]
        items[f, local, 0] = o_row.to(torch.int32)
        pslot = pbase[o_row] + o_part * n_col_splits + k
        items[f, local, 3] = torch.where(needs[o_row], pslot, torch.full_like(pslot, -1)).to(torch.int32)
    hdr = torch.empty((n_flat, 2), dtype=torch.int32, device=dev)
    hdr[flat_of.reshape(-1), 0] = bounds[:-1].to(torch.int32).unsqueeze(1).expand(n_col_splits, n_row_tiles).reshape(-1)
    hdr[flat_of.reshape(-1), 1] = bounds[1:].to(torch.int32).unsqueeze(1).expand(n_col_splits, n_row_tiles).reshape(-1)
    lr = torch.nonzero(needs).squeeze(1)
    long_rows = torch.stack([lr, pbase[lr], n_parts_r[lr], torch.zeros_like(lr)], 1).to(torch.int32).contiguous() \
        if lr.numel() else torch.empty((0, 4), dtype=torch.int32, device=dev)
    # ---- entries: the CSR re-ordered by (tile, block, wave, destination slot)
    nblk_max = max(1, per_split // blk)
    n_seg = n_col_splits * n_row_tiles * nblk_max * TILE_WAVES
    if (TILE_PLAN_KERNEL and TILE_SHARED_PAIRS and dev.type == "cuda" and total < 2 ** 31 - 1 and csr.rowptr.dtype == torch.int32
            and csr.col.dtype == torch.int32 and csr.val.dtype == torch.float32 and n_seg < 2 ** 31 - 1):
        # the library's plan walk: count, prefix sum, fill (csrc/wgnn_plan.hip)
        slot_vrow = torch.full((n_row_tiles * TILE_ROWS,), -1, dtype=torch.int32, device=dev)
        slot_vrow[tile * TILE_ROWS + local] = order.to(torch.int32)
        vrow32, vpart32, vk32 = vrow.to(torch.int32), vpart.to(torch.int32), k_r[vrow].to(torch.int32)
        flat_t = torch.empty(n_flat, dtype=torch.int32, device=dev)
        flat_t[flat_of.reshape(-1)] = torch.arange(n_row_tiles, device=dev, dtype=torch.int32).repeat(n_col_splits)
        hdr_c = hdr.contiguous()
        seg_total = torch.zeros(n_seg, dtype=torch.int32, device=dev)
        rowptr_c, col_c, val_c = csr.rowptr.contiguous(), csr.col.contiguous(), csr.val.contiguous()
        common = (_ptr(slot_vrow), _ptr(vrow32), _ptr(vpart32), _ptr(vk32), _ptr(flat_t), _ptr(hdr_c), n_flat, TILE_WAVES,
                  TILE_ROWS // TILE_WAVES, nblk_max, blk)
        _lib.check(_lib.call(dev, "wgnn_tile_plan_count", _ptr(rowptr_c), _ptr(col_c), *common, _ptr(seg_total), None,
                             _stream(dev)), "wgnn_tile_plan_count")
        seg_ptr = torch.zeros(n_seg + 1, dtype=torch.int64, device=dev)
        torch.cumsum(seg_total + (seg_total & 1), 0, out=seg_ptr[1:])        # an odd segment is padded to even (ABI 0.2.5)
        n_entries = int(seg_ptr[-1])
        if n_entries >= 2 ** 31 - 1:
            raise ValueError("tile plan with >= 2^31 entries: shard the operand")
        seg_ptr32 = seg_ptr.to(torch.int32)
        entries = torch.empty((n_entries, 2), dtype=torch.int32, device=dev)
        _lib.check(_lib.call(dev, "wgnn_tile_plan_fill", _ptr(rowptr_c), _ptr(col_c), _ptr(val_c), *common, _ptr(seg_total),
                             None, _ptr(seg_ptr32), _ptr(entries), _stream(dev)), "wgnn_tile_plan_fill")
        return TilePlan(items.contiguous(), hdr_c, long_rows, n_part,
                        n_row_tiles, n_col_splits, n_loaders, entries, seg_ptr32, nblk_max, block_rows, geom)
    tile_v = torch.empty(V, dtype=torch.int64, device=dev); tile_v[order] = tile
    wave_v = torch.empty(V, dtype=torch.int64, device=dev); wave_v[order] = wave
    slot_v = torch.empty(V, dtype=torch.int64, device=dev); slot_v[order] = slot_in_wave
    # (round 5: 32-bit index arithmetic and one gather per non-zero where the operand allows it - these are 8e7-element passes)
    i32 = total < 2 ** 31 - 1 and n_col_splits * n_row_tiles * nblk_max * TILE_WAVES < 2 ** 31 - 1
    it = torch.int32 if i32 else torch.int64
    rows_of = torch.repeat_interleave(torch.arange(R, device=dev, dtype=it), nnz)
    if few_rows:
        pos = torch.arange(rows_of.shape[0], device=dev, dtype=it) - csr.rowptr.to(it)[rows_of]
        virt = vbase.to(it)[rows_of] + pos % k_r.to(it)[rows_of]             # the virtual row every non-zero belongs to
        del pos
    else:
        virt = rows_of                                                        # k_r == 1 everywhere: virtual row == row
    del rows_of
    colv = csr.col.to(it)
    if n_col_splits == 1:
        blk_of = torch.div(colv, blk, rounding_mode="floor")
        src_local = colv - blk_of * blk
        base_v = ((flat_of[0][tile_v] * nblk_max) * TILE_WAVES + wave_v).to(it)     # per virtual row: its segment at block 0
        key = base_v[virt] + blk_of * TILE_WAVES
        del blk_of, base_v
    else:
        ksplit = torch.div(colv, per_split, rounding_mode="floor")
        rel = colv - ksplit * per_split
        src_local = rel % blk
        key = ((flat_of.to(it)[ksplit.long(), tile_v[virt]] * nblk_max + torch.div(rel, blk, rounding_mode="floor")) * TILE_WAVES
               + wave_v.to(it)[virt])
        del ksplit, rel
    meta = ((slot_v.to(torch.int32)[virt] << 8) | src_local.to(torch.int32))
    del colv, src_local
    val_bits = csr.val.view(torch.int32)
    if not TILE_SHARED_PAIRS:
        key = key.long() * (TILE_ROWS // TILE_WAVES) + slot_v[virt]
        del virt
        perm = torch.sort(key, stable=True).indices
        counts = torch.bincount(torch.div(key, TILE_ROWS // TILE_WAVES, rounding_mode="floor"), minlength=n_seg)
        del key
        seg_ptr = torch.zeros(n_seg + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=seg_ptr[1:])
        entries = torch.stack([meta[perm], val_bits[perm]], 1).contiguous()
        del perm, meta
    else:
        entries, seg_ptr = _pair_segment_entries(key, meta, val_bits, n_seg, 0x3F if geom.tall else 0xF)
        del key, meta, virt
    return TilePlan(items.contiguous(), hdr.contiguous(), long_rows, n_part,
                    n_row_tiles, n_col_splits, n_loaders, entries, seg_ptr.to(torch.int32), nblk_max, block_rows, geom)
