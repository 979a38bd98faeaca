"""Seeded device-side neighbour sampler: the ``num_neighbors > 0`` training mode of the reference
(``train.py:37-40,71-78``: ``NeighborSampler(expand_factor=num_neighbors, neighbor_type='in', num_hops=n_layers)``).

DGL 0.4.3 semantics restated: a NodeFlow is built backwards from the seed cells; for every node of layer i+1 at most
``num_neighbors`` of its in-edges are drawn uniformly without replacement - the unit self-loop is one of those edges
(it was added to the graph explicitly, preprocess_internal.py:213-214) - and ``fn.mean`` then divides by the number
of DRAWN edges.  Layer i is the set of sources of the drawn edges.

Here a drawn block is a small destination-major CSR over the *global* source ids (no node relabelling: the source
tables stay full-size), so it runs through the same K1 / K2 / K3 kernels as the full graph.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from .graph import AggCsr, CellGeneGraph, device_plan


@dataclass
class SampledBlock:
    rows: torch.Tensor          # int64 [n]  destination ids (cell index or gene index) in NodeFlow order
    csr: AggCsr                 # n rows; cols = global ids of the other node type; inv_deg = 1 / #drawn edges
    self_drawn: torch.Tensor    # float32 [n] 1.0 where the self-loop edge was drawn


def sample_block(csr: AggCsr, rows: torch.Tensor, k: int, gen: Optional[torch.Generator]) -> SampledBlock:
    """Draw min(k, deg+1) of the deg+1 in-edges (deg real edges + the self-loop) of every row in ``rows``."""
    dev = csr.device
    rows = rows.to(dev).long()
    n = rows.shape[0]
    rp = csr.rowptr.long()
    beg, deg = rp[rows], rp[rows + 1] - rp[rows]
    cand = deg + 1                                              # + the self-loop
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cand, 0, out=off[1:])
    total = int(off[-1])
    owner = torch.repeat_interleave(torch.arange(n, device=dev), cand)
    pos = torch.arange(total, device=dev) - off[owner]         # 0..deg ; pos == deg is the self-loop
    key = torch.rand(total, device=dev, generator=gen, dtype=torch.float64)
    order = torch.sort(owner.double() + key).indices           # random order inside every row
    rank = torch.empty(total, dtype=torch.int64, device=dev)
    rank[order] = torch.arange(total, device=dev) - off[owner[order]]
    keep = rank < k
    is_self = pos == deg[owner]
    self_drawn = torch.zeros(n, dtype=torch.float32, device=dev)
    self_drawn[owner[keep & is_self]] = 1.0
    real = keep & ~is_self
    m_real = torch.bincount(owner[real], minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(m_real, 0, out=rowptr[1:])
    eidx = (beg[owner] + pos)[real]                            # edge positions in the parent CSR (row-major, ascending)
    inv = 1.0 / (m_real.float() + self_drawn).clamp(min=1.0)
    rowptr32 = rowptr.to(torch.int32)
    bound = max(1, min(int(k), csr.max_row_nnz))               # host-known bound on the drawn row length
    # plan on the device (every row holds <= k drawn edges): no row-pointer read-back.  What remains host-visible per
    # block is the number of candidate edges (`total`, sizes the random draw) - a NodeFlow's sizes are data-dependent.
    sub = AggCsr(rowptr32, csr.col[eidx].contiguous(), csr.val[eidx].contiguous(), inv.contiguous(), n, csr.n_cols,
                 device_plan(rowptr32, n, bound, max(1, csr.plan.chunk)), None)
    sub._max_row_nnz = bound
    return SampledBlock(rows, sub, self_drawn)


class DeviceSampler:
    """State of the sync-free sampler (``wgnn_sample_rows``): a seed and a DEVICE step counter.  Every drawn NodeFlow
    advances the counter on the stream, so a captured training step draws a fresh sample at every replay."""

    def __init__(self, seed: int, device):
        self.seed = int(seed) & ((1 << 63) - 1)
        self.step = torch.zeros(1, dtype=torch.int64, device=device)

    def advance(self) -> None:
        self.step += 1


def sample_block_static(csr: AggCsr, rows: Optional[torch.Tensor], k: int, sampler: DeviceSampler, stream_id: int) -> SampledBlock:
    """K5 (``wgnn_sample_rows``): draw min(k, deg+1) of the deg+1 in-edges of every row in ``rows`` (None = all rows) with
    static output shapes and no host synchronisation.  The block is an ELL-padded ``AggCsr`` (``ell_k`` slots per row)."""
    from . import _lib
    from .graph import _ptr, _stream, Plan
    dev = csr.device
    kk = max(1, min(int(k), csr.max_row_nnz + 1))               # host-known bound on the draws per row (same draw law)
    if kk > 256:
        raise _lib.WgnnError("wgnn_sample_rows draws at most 256 edges per row")
    if rows is None:
        n, ids32, rows_l = csr.n_rows, None, torch.arange(csr.n_rows, device=dev)
    else:
        rows_l = rows.to(dev).long()
        n, ids32 = rows_l.shape[0], rows_l.to(torch.int32).contiguous()
    out_col = torch.zeros(max(1, n * kk), dtype=torch.int32, device=dev)
    out_val = torch.zeros(max(1, n * kk), dtype=torch.float32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    self_drawn = torch.zeros(n, dtype=torch.float32, device=dev)
    inv = torch.ones(n, dtype=torch.float32, device=dev)
    _lib.check(_lib.call(dev, "wgnn_sample_rows", _ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.val), _ptr(ids32), n, kk,
                         sampler.seed, _ptr(sampler.step), int(stream_id), _ptr(out_col), _ptr(out_val), _ptr(cnt),
                         _ptr(self_drawn), _ptr(inv), _stream(dev)), "wgnn_sample_rows")
    slot = torch.arange(n, device=dev, dtype=torch.int32)
    items = torch.stack([slot, slot * kk, slot * kk + cnt, torch.full_like(slot, -1)], 1).contiguous()
    starts = torch.arange(n + 1, device=dev, dtype=torch.int32) * kk
    sub = AggCsr(starts, out_col[: n * kk], out_val[: n * kk], inv, n, csr.n_cols,
                 Plan(items, torch.empty((0, 4), dtype=torch.int32, device=dev), 0, kk), None, ell_k=kk, ell_cnt=cnt)
    sub._max_row_nnz = kk
    return SampledBlock(rows_l, sub, self_drawn)


def sample_nodeflow_static(g: CellGeneGraph, seed_cells: torch.Tensor, n_layers: int, k: int,
                           sampler: DeviceSampler) -> "NodeFlow":
    """NodeFlow with static shapes: the last block covers the seed cells; every lower block covers ALL cells and ALL
    genes (each node draws its own <= k in-edges, exactly like the nodes a DGL NodeFlow would contain - the others are
    simply never read).  O((G + C) k) work per layer instead of a data-dependent closure, no ``unique``, no host
    synchronisation: a whole sampled training step can be captured in a hipGraph."""
    blocks: List[Tuple[SampledBlock, Optional[SampledBlock]]] = []
    for i in range(n_layers):
        if i == n_layers - 1:
            blocks.append((sample_block_static(g.cg, seed_cells, k, sampler, 2 * i), None))
        else:
            blocks.append((sample_block_static(g.cg, None, k, sampler, 2 * i),
                           sample_block_static(g.gc, None, k, sampler, 2 * i + 1)))
    sampler.advance()
    return NodeFlow(blocks)


@dataclass
class NodeFlow:
    """blocks[i] = (cell block, gene block or None) feeding layer i+1; layer sets are implied by the blocks."""
    blocks: List[Tuple[SampledBlock, Optional[SampledBlock]]]


def sample_nodeflow(g: CellGeneGraph, seed_cells: torch.Tensor, n_layers: int, k: int,
                    gen: Optional[torch.Generator] = None) -> NodeFlow:
    """``seed_cells``: cell indices (0-based, i.e. node id - G) in the order the logits are wanted."""
    dev = g.device
    cells = seed_cells.to(dev).long()
    genes = torch.empty(0, dtype=torch.int64, device=dev)
    blocks: List[Tuple[SampledBlock, Optional[SampledBlock]]] = []
    for _ in range(n_layers):
        cb = sample_block(g.cg, cells, k, gen)
        gb = sample_block(g.gc, genes, k, gen) if genes.numel() else None
        blocks.insert(0, (cb, gb))
        # next (lower) layer: sources of the drawn edges, plus nodes whose self-loop was drawn
        new_genes = [cb.csr.col.long()]
        new_cells = [cells[cb.self_drawn > 0]]
        if gb is not None:
            new_cells.append(gb.csr.col.long())
            new_genes.append(genes[gb.self_drawn > 0])
        genes = torch.unique(torch.cat(new_genes))
        cells = torch.unique(torch.cat(new_cells))
    return NodeFlow(blocks)
