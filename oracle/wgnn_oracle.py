"""CPU restatement ("oracle") of scDeepSort's weighted-GNN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``scdeepsort_amd/`` imports this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may use it, and only as the checker.

PARITY STATUS - pinned against the reference's own Python code, UNPINNED against DGL.  The reference ships no tests,
golden vectors or checkpoints for this path, and the reduce / NodeFlow part of its arithmetic lives in DGL 0.4.3.post2
(requirements.txt:5), which is neither vendored under /root/reference nor installable here.  What could be done, and
is: the reference's OWN code for the path - ``GNN.forward`` / ``message_func`` / ``NodeUpdate`` (models/gnn.py:10-68)
and ``normalize_weight`` (utils/preprocess_internal.py:15-23) - was imported from /root/reference in the build
container and EXECUTED over a ~70-line stand-in for the DGL objects it touches (NodeFlow frames + ``block_compute`` +
``fn.mean`` with their documented semantics); the resulting logits and normalised edge weights are committed as
``tests/golden/refcode_*.npz`` (script: ``tests/golden/make_refcode_golden.py``) and both formulations below reproduce
them to 2e-6.  What stays restated-from-documentation is DGL's side: ``fn.mean`` = sum over in-edges / in-degree, a
NodeFlow block = all parent in-edges when expand_factor >= degree, ``edata[...]`` writes through.  Further pins:
hand-derived known answers (tests/golden/kat_*.json), agreement of two independent formulations (edge-list vs CSR),
algebraic properties, finite-difference gradients.

Two formulations live here:

* **edge-list** (``build_reference_graph`` + ``nodeflow_forward``): mirrors the
  reference literally - genes then cells as nodes, an ``id`` node field, one
  edge per direction, per-destination weight normalisation, self-loops added
  afterwards, per-edge materialised messages, mean over in-edges, mini-batch
  NodeFlow closure.  Written with torch ops so ``autograd`` supplies the
  gradient oracle.
* **CSR / full-graph** (``csr_forward``): the layer-wise full-graph form the
  HIP path implements ("equivalent full-graph
  formulation"), evaluated with scipy in fp32 or fp64.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp
import torch


# ----------------------------------------------------------------------------
# graph construction   (utils/preprocess_internal.py:15-23,107-110,160-173,210-215
#                       utils/preprocess.py:126-134,184-187,212-221)
# ----------------------------------------------------------------------------
@dataclass
class RefGraph:
    """Edge-list graph exactly as the reference lays it out in a DGLGraph."""
    num_genes: int
    num_cells: int
    node_id: np.ndarray      # [N] int32: gene index for genes, -1 for cells   (preprocess_internal.py:109,168)
    src: np.ndarray          # [E] int64
    dst: np.ndarray          # [E] int64
    weight: np.ndarray       # [E] float32, normalised; self-loops (w=1) are the last N edges

    @property
    def num_nodes(self) -> int:
        return self.num_genes + self.num_cells


def build_reference_graph(expr: sp.csr_matrix, support_mask: Optional[np.ndarray] = None) -> RefGraph:
    """Mirror of the reference graph build.

    ``expr`` is the (cells x genes) matrix of raw expression values already
    filtered by ``> threshold`` (preprocess_internal.py:158).  Every stored
    entry yields a gene->cell edge; cells with ``support_mask[c]`` True
    (all cells of a training graph) also yield the cell->gene edge
    (preprocess_internal.py:170-173).  Test cells of a predict graph carry
    gene->cell edges only (preprocess.py:184-187).
    """
    expr = sp.csr_matrix(expr).astype(np.float32)
    expr.sort_indices()
    C, G = expr.shape
    if support_mask is None:
        support_mask = np.ones(C, dtype=bool)
    support_mask = np.asarray(support_mask, dtype=bool)
    coo = expr.tocoo()
    cell = coo.row.astype(np.int64) + G        # cells are appended after the genes (preprocess_internal.py:160)
    gene = coo.col.astype(np.int64)
    w = coo.data.astype(np.float32)
    sup = support_mask[coo.row]
    # cell->gene (support cells only) then gene->cell, raw weight = expression value on both
    src = np.concatenate([cell[sup], gene])
    dst = np.concatenate([gene[sup], cell])
    wt = np.concatenate([w[sup], w])
    N = G + C
    # normalize_weight (preprocess_internal.py:15-23): for every node with >=1 in-edge,
    #   w_in <- in_degree * w_in / sum(w_in)        (before self-loops are added, :211-214)
    wt_t = torch.from_numpy(wt)
    dst_t = torch.from_numpy(dst)
    deg = torch.zeros(N, dtype=torch.int64).index_add_(0, dst_t, torch.ones_like(dst_t))
    # torch.sum(edge_w) (:23) is a pairwise fp32 reduction; a float64 accumulation rounded to fp32 is its
    # closest order-independent stand-in (a sequential fp32 index_add_ would be ~1e-5 off on 10^3-10^4 terms)
    ssum = torch.zeros(N, dtype=torch.float64).index_add_(0, dst_t, wt_t.double()).float()
    wt_n = (deg[dst_t] * wt_t / ssum[dst_t]).numpy().astype(np.float32)
    # self-loop on every node, weight 1 (preprocess_internal.py:213-214)
    loops = np.arange(N, dtype=np.int64)
    src = np.concatenate([src, loops])
    dst = np.concatenate([dst, loops])
    wt_n = np.concatenate([wt_n, np.ones(N, dtype=np.float32)])
    node_id = np.concatenate([np.arange(G, dtype=np.int32), -np.ones(C, dtype=np.int32)])
    return RefGraph(G, C, node_id, src, dst, wt_n)


# ----------------------------------------------------------------------------
# model parameters   (models/gnn.py:10-16,29-45)
# ----------------------------------------------------------------------------
def init_params(in_feats: int, n_hidden: int, n_classes: int, n_layers: int, gene_num: int,
                seed: int = 0, random_alpha: bool = True, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """state_dict with the reference's key names and shapes (gnn.py:13,37-45)."""
    g = torch.Generator().manual_seed(seed)
    gain = torch.nn.init.calculate_gain('relu')                       # gnn.py:16
    sd: Dict[str, torch.Tensor] = {}

    def xavier(out_f, in_f):
        bound = gain * (6.0 / (in_f + out_f)) ** 0.5
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound

    dims = [in_feats] + [n_hidden] * n_layers
    for i in range(n_layers):
        sd[f'layers.{i}.fc_neigh.weight'] = xavier(dims[i + 1], dims[i])
        b = 1.0 / dims[i] ** 0.5                                     # torch nn.Linear default bias init
        sd[f'layers.{i}.fc_neigh.bias'] = (torch.rand(dims[i + 1], generator=g) * 2 - 1) * b
    if random_alpha:   # the reference initialises alpha to ones (gnn.py:43); U(0.5,1.5) exercises the alpha path
        sd['alpha'] = torch.rand(gene_num + 2, 1, generator=g) + 0.5
    else:
        sd['alpha'] = torch.ones(gene_num + 2, 1)
    sd['linear.weight'] = xavier(n_classes, n_hidden)
    sd['linear.bias'] = (torch.rand(n_classes, generator=g) * 2 - 1) / n_hidden ** 0.5
    return {k: v.to(dtype) for k, v in sd.items()}


# ----------------------------------------------------------------------------
# edge-list formulation: message / mean / node update   (gnn.py:47-56,65,18-25)
# ----------------------------------------------------------------------------
def alpha_index(src_id: np.ndarray, dst_id: np.ndarray, gene_num: int) -> np.ndarray:
    """k(e) exactly as gnn.py:49-53 (np.where cascade, later rules override)."""
    idx = np.full(src_id.shape, gene_num + 1, dtype=np.int64)                 # :49  default = cell self-loop
    idx = np.where((src_id >= 0) & (dst_id < 0), src_id, idx)                # :51  gene -> cell
    idx = np.where((dst_id >= 0) & (src_id < 0), dst_id, idx)                # :52  cell -> gene
    idx = np.where((dst_id >= 0) & (src_id >= 0), gene_num, idx)             # :53  gene -> gene (self-loop)
    return idx


def block_compute(h_src_layer: torch.Tensor, e_src: np.ndarray, e_dst: np.ndarray, e_w: torch.Tensor,
                  src_node_id: np.ndarray, dst_node_id: np.ndarray, n_dst: int,
                  alpha: torch.Tensor, gene_num: int) -> torch.Tensor:
    """One ``nf.block_compute(i, message_func, fn.mean('m','neigh'), ...)`` up to 'neigh'.

    message (gnn.py:54,56):  m_e = (h[src] * alpha[k(e)]) * w_e
    reduce  [DGL fn.mean]:   neigh[v] = sum_{e into v} m_e / in_degree_block(v)
    """
    k = alpha_index(src_node_id[e_src], dst_node_id[e_dst], gene_num)
    a = alpha[torch.from_numpy(k)]                                           # [E,1]
    m = (h_src_layer[torch.from_numpy(e_src)] * a) * e_w.unsqueeze(-1)       # [E,D]
    dst_t = torch.from_numpy(e_dst)
    s = torch.zeros(n_dst, h_src_layer.shape[1], dtype=m.dtype).index_add_(0, dst_t, m)
    deg = torch.zeros(n_dst, dtype=m.dtype).index_add_(0, dst_t, torch.ones(len(e_dst), dtype=m.dtype))
    return s / deg.unsqueeze(-1)


def node_update(neigh: torch.Tensor, W: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """NodeUpdate.forward with activation=F.relu, norm=None (gnn.py:18-25; train.py:31)."""
    return torch.relu(torch.nn.functional.linear(neigh, W, b))


class _InEdges:
    """dst-major index of a RefGraph (what DGL's sampler walks)."""

    def __init__(self, g: RefGraph):
        order = np.argsort(g.dst, kind='stable')
        self.src = g.src[order]
        self.w = g.weight[order]
        counts = np.bincount(g.dst, minlength=g.num_nodes)
        self.ptr = np.concatenate([[0], np.cumsum(counts)])

    def of(self, nodes: np.ndarray, rng: Optional[np.random.Generator], num_neighbors: int, picker=None, block=0):
        srcs, dsts, ws = [], [], []
        for j, v in enumerate(nodes):
            lo, hi = self.ptr[v], self.ptr[v + 1]
            sel = np.arange(lo, hi)
            if picker is not None:                              # an externally drawn sample (parity tests)
                sel = sel[np.isin(self.src[sel], picker(block, int(v)))]
            elif num_neighbors and num_neighbors < hi - lo:     # train.py:37-40: uniform w/o replacement
                sel = np.sort(rng.choice(sel, size=num_neighbors, replace=False))
            srcs.append(self.src[sel]); ws.append(self.w[sel]); dsts.append(np.full(len(sel), j, dtype=np.int64))
        return np.concatenate(srcs), np.concatenate(dsts), np.concatenate(ws)


def nodeflow_forward(sd: Dict[str, torch.Tensor], g: RefGraph, features: torch.Tensor, seeds: Sequence[int],
                     n_layers: int, num_neighbors: int = 0, rng: Optional[np.random.Generator] = None,
                     dropout_masks: Optional[List[torch.Tensor]] = None, picker=None) -> torch.Tensor:
    """GNN.forward on the NodeFlow of one seed batch (gnn.py:58-68 + train.py:71-81).

    NodeFlow emulation [DGL-ext semantics]: layer L = seeds; layer i = unique
    sources of the (sampled) parent in-edges of layer i+1; block i = those edges.
    With ``num_neighbors == 0`` every in-edge is taken (expand_factor >= max degree,
    train.py:37-38).  ``dropout_masks[i]`` (already scaled by 1/(1-p), indexed by
    *parent* node id) reproduces ``self.dropout(h)`` on layer i's rows (gnn.py:62-63).
    ``picker(block, v)`` (optional) returns the parent ids of the in-edge sources drawn for node ``v`` in block
    ``block`` (v itself = its self-loop): lets a test replay a sample drawn elsewhere instead of ``rng``.
    Returns logits for ``seeds`` in the given order.
    """
    ine = _InEdges(g)
    layers = [np.asarray(seeds, dtype=np.int64)]
    blocks = []
    for t in range(n_layers):
        e_src_parent, e_dst_local, e_w = ine.of(layers[0], rng, num_neighbors, picker, n_layers - 1 - t)
        prev, e_src_local = np.unique(e_src_parent, return_inverse=True)
        blocks.insert(0, (e_src_local, e_dst_local, torch.from_numpy(e_w).to(features.dtype)))
        layers.insert(0, prev)
    alpha = sd['alpha']
    h = features[torch.from_numpy(layers[0])]                         # gnn.py:59  layer-0 'features'
    for i in range(n_layers):
        if dropout_masks is not None:
            h = h * dropout_masks[i][torch.from_numpy(layers[i])]
        e_src, e_dst, e_w = blocks[i]
        neigh = block_compute(h, e_src, e_dst, e_w, g.node_id[layers[i]], g.node_id[layers[i + 1]],
                              len(layers[i + 1]), alpha, g.num_genes)
        h = node_update(neigh, sd[f'layers.{i}.fc_neigh.weight'], sd[f'layers.{i}.fc_neigh.bias'])
    return torch.nn.functional.linear(h, sd['linear.weight'], sd['linear.bias'])     # gnn.py:66-67


def edgelist_full_forward(sd, g: RefGraph, features: torch.Tensor, n_layers: int,
                          return_hidden: bool = False):
    """Edge-list formulation over ALL nodes at once (every node is a seed)."""
    src_id = g.node_id
    h = features
    e_w = torch.from_numpy(g.weight).to(features.dtype)
    hs = []
    for i in range(n_layers):
        neigh = block_compute(h, g.src, g.dst, e_w, src_id, src_id, g.num_nodes, sd['alpha'], g.num_genes)
        h = node_update(neigh, sd[f'layers.{i}.fc_neigh.weight'], sd[f'layers.{i}.fc_neigh.bias'])
        hs.append(h)
    logits = torch.nn.functional.linear(h, sd['linear.weight'], sd['linear.bias'])
    return (logits, hs) if return_hidden else logits


# ----------------------------------------------------------------------------
# CSR / full-graph formulation   
# ----------------------------------------------------------------------------
@dataclass
class CsrGraph:
    num_genes: int
    num_cells: int
    A_cg: sp.csr_matrix     # [C,G] normalised gene->cell weights (row = destination cell)
    A_gc: sp.csr_matrix     # [G,C] normalised cell->gene weights (row = destination gene)
    deg_c: np.ndarray       # [C] in-degree of cells incl. self-loop
    deg_g: np.ndarray       # [G] in-degree of genes incl. self-loop


def build_csr_graph(expr: sp.csr_matrix, support_mask: Optional[np.ndarray] = None, dtype=np.float32) -> CsrGraph:
    """Same operand as ``build_reference_graph`` but as two dst-major CSRs.

    A_cg[c,g] = deg_c * x[c,g] / sum_g x[c,g];  A_gc[g,c] = deg_g * x[c,g] / sum_{c in support} x[c,g]
    (normalize_weight, preprocess_internal.py:17-23).  Self-loops stay implicit.
    """
    expr = sp.csr_matrix(expr).astype(np.float32)
    expr.sort_indices()
    C, G = expr.shape
    if support_mask is None:
        support_mask = np.ones(C, dtype=bool)
    support_mask = np.asarray(support_mask, dtype=bool)
    nnz_c = np.diff(expr.indptr)
    row_sum = np.asarray(expr.astype(np.float64).sum(axis=1)).ravel().astype(np.float32)
    A_cg = expr.copy().astype(dtype)
    rows = np.repeat(np.arange(C), nnz_c)
    with np.errstate(divide='ignore', invalid='ignore'):
        A_cg.data = (nnz_c[rows].astype(np.float32) * expr.data / row_sum[rows]).astype(dtype)   # fp32 like the reference
    sup = sp.diags(support_mask.astype(np.float32)) @ expr
    sup = sp.csr_matrix(sup); sup.eliminate_zeros()
    XT = sp.csr_matrix(sup.T); XT.sort_indices()
    nnz_g = np.diff(XT.indptr)
    col_sum = np.asarray(XT.astype(np.float64).sum(axis=1)).ravel().astype(np.float32)
    A_gc = XT.copy().astype(dtype)
    rows = np.repeat(np.arange(G), nnz_g)
    A_gc.data = (nnz_g[rows].astype(np.float32) * XT.data / col_sum[rows]).astype(dtype)
    return CsrGraph(G, C, A_cg, A_gc, (nnz_c + 1).astype(np.int64), (nnz_g + 1).astype(np.int64))


def csr_aggregate(cg: CsrGraph, alpha: np.ndarray, Hg: np.ndarray, Hc: np.ndarray, want_genes: bool = True):
    """neigh for cells and genes:  Z_c = (A_cg diag(a) H_g + alpha[G+1] H_c)/d_c ;
    Z_g = (diag(a) A_gc H_c + alpha[G] H_g)/d_g      (gnn.py:47-56,65)."""
    G = cg.num_genes
    a = alpha.reshape(-1)[:G].astype(Hg.dtype)
    Zc = (cg.A_cg.astype(Hg.dtype) @ (Hg * a[:, None]) + alpha.reshape(-1)[G + 1] * Hc) / cg.deg_c[:, None].astype(Hg.dtype)
    Zg = None
    if want_genes:
        Zg = (a[:, None] * (cg.A_gc.astype(Hg.dtype) @ Hc) + alpha.reshape(-1)[G] * Hg) / cg.deg_g[:, None].astype(Hg.dtype)
    return Zc, Zg


def csr_forward(sd: Dict[str, torch.Tensor], cg: CsrGraph, features: np.ndarray, n_layers: int,
                dtype=np.float32, return_hidden: bool = False):
    """Layer-wise full-graph forward; returns logits for all cells (rows = cells in order)."""
    G = cg.num_genes
    p = {k: v.detach().cpu().numpy().astype(dtype) for k, v in sd.items()}
    Hg = features[:G].astype(dtype)
    Hc = features[G:].astype(dtype)
    hidden = []
    for i in range(n_layers):
        last = i == n_layers - 1
        Zc, Zg = csr_aggregate(cg, p['alpha'], Hg, Hc, want_genes=not last)
        W, b = p[f'layers.{i}.fc_neigh.weight'], p[f'layers.{i}.fc_neigh.bias']
        Hc = np.maximum(Zc @ W.T + b, 0)
        if not last:
            Hg = np.maximum(Zg @ W.T + b, 0)
        hidden.append((Hg if not last else None, Hc))
    logits = Hc @ p['linear.weight'].T + p['linear.bias']
    return (logits, hidden) if return_hidden else logits


# ----------------------------------------------------------------------------
# training step + post-processing   (train.py:34-36,80-87,106-113; predict.py:78-88)
# ----------------------------------------------------------------------------
def loss_and_grads(sd: Dict[str, torch.Tensor], g: RefGraph, features: torch.Tensor, seeds: Sequence[int],
                   labels: torch.Tensor, n_layers: int, dtype=torch.float64):
    """CrossEntropyLoss(reduction='sum') on one seed batch and autograd gradients of every
    parameter (train.py:36,80-84), evaluated through the edge-list formulation."""
    p = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
    logits = nodeflow_forward(p, g, features.to(dtype), seeds, n_layers)
    loss = torch.nn.functional.cross_entropy(logits, labels, reduction='sum')
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in p.items()}, logits.detach()


def postprocess(logits: np.ndarray, unsure_rate: float) -> Tuple[np.ndarray, np.ndarray]:
    """softmax -> 'unsure' (-1) iff max_prob < unsure_rate/num_classes, else argmax
    (predict.py:78-88; train.py:106-113)."""
    z = logits - logits.max(axis=1, keepdims=True)
    prob = np.exp(z); prob /= prob.sum(axis=1, keepdims=True)
    pred = prob.argmax(axis=1)
    unsure = prob.max(axis=1) < unsure_rate / logits.shape[1]
    return np.where(unsure, -1, pred), prob
