"""CPU baselines of SURVEY.md section 8d, timed by ``bench.py``'s ``cpu_baseline`` leg beside the C/OpenMP port.

TEST INFRASTRUCTURE ONLY (same rule as the rest of ``oracle/``): nothing under ``scdeepsort_amd/`` imports this.

The reference itself cannot run here (DGL 0.4.3.post2 absent), so these are RESTATEMENTS of how it executes:

* **B1 "reference-style"** - what ``train.py:71-80`` + ``models/gnn.py:47-65`` do per seed batch: build the batch's
  block (all in-edges of the seeds + their self-loops), MATERIALISE the per-edge messages ``(h[src]*alpha[k])*w`` as an
  ``[E_b, D]`` tensor (gnn.py:54-56), sum them per destination (DGL's reduce; ``index_add_`` here), divide by the
  in-degree, apply ``NodeUpdate`` (gnn.py:18-25).  Uses ``wgnn_oracle.block_compute`` - the literal mirror of
  message_func / fn.mean - on 1-hop blocks of ``batch`` seed cells.  A 2-hop closure of 500 cells at cfg3 is the whole
  graph (one message tensor = 8e7 x 400 x 4 B = 128 GB), so the 2-layer forward is EXTRAPOLATED from the measured
  seconds per edge-float: L1 genes<-cells and L1 cells<-genes at D = dense_dim, L2 cells<-genes at D = hidden.
* **B2 "strong"** - the same math as ``torch`` CSR SpMM over the full graph + ``F.linear`` (project-first, i.e. the
  cheapest order), all host threads.
"""
from __future__ import annotations

import time
from typing import Dict

import numpy as np
import torch

from . import wgnn_oracle as O


def b2_torch_csr_forward(sd: Dict[str, torch.Tensor], cg: "O.CsrGraph", feats: np.ndarray, n_layers: int) -> np.ndarray:
    """Full-graph forward with torch.sparse CSR SpMM (MKL) + F.linear, fp32, project-first."""
    G = cg.num_genes
    F = torch.nn.functional

    def tcsr(m):
        return torch.sparse_csr_tensor(torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int64)),
                                       torch.from_numpy(m.data.astype(np.float32)), size=m.shape)
    cache = getattr(cg, "_torch_csr", None)
    if cache is None:
        cache = (tcsr(cg.A_cg), tcsr(cg.A_gc), torch.from_numpy(1.0 / cg.deg_c.astype(np.float32)).unsqueeze(1),
                 torch.from_numpy(1.0 / cg.deg_g.astype(np.float32)).unsqueeze(1))
        cg._torch_csr = cache
    A_cg, A_gc, inv_c, inv_g = cache
    a = sd["alpha"].reshape(-1).float()
    h_g, h_c = torch.from_numpy(feats[:G]).float(), torch.from_numpy(feats[G:]).float()
    with torch.no_grad():
        for i in range(n_layers):
            W, b = sd[f"layers.{i}.fc_neigh.weight"].float(), sd[f"layers.{i}.fc_neigh.bias"].float()
            last = i == n_layers - 1
            p_g, p_c = F.linear(h_g, W), F.linear(h_c, W)
            n_c = torch.relu((torch.sparse.mm(A_cg, p_g * a[:G, None]) + a[G + 1] * p_c) * inv_c + b)
            if not last:
                h_g = torch.relu((a[:G, None] * torch.sparse.mm(A_gc, p_c) + a[G] * p_g) * inv_g + b)
            h_c = n_c
        return F.linear(h_c, sd["linear.weight"].float(), sd["linear.bias"].float()).numpy()


def b1_reference_style(sd: Dict[str, torch.Tensor], cg: "O.CsrGraph", feats: np.ndarray, n_layers: int, hidden: int,
                       batch: int = 500, max_batches: int = 8, budget_s: float = 8.0, seed: int = 0) -> dict:
    """Times 1-hop seed batches executed the reference's way and extrapolates to the ``n_layers``-layer full forward.
    Returns {"s_per_forward", "batches", "edges", "s_per_edge_float", "check_err"} - ``check_err`` = max abs deviation of
    the sampled rows' layer-1 output from the CSR formulation (the two restatements must agree)."""
    G, C = cg.num_genes, cg.num_cells
    D_in = feats.shape[1]
    alpha = sd["alpha"].float()
    W, b = sd["layers.0.fc_neigh.weight"].float(), sd["layers.0.fc_neigh.bias"].float()
    h = torch.from_numpy(feats).float()
    node_id = np.concatenate([np.arange(G, dtype=np.int32), -np.ones(C, dtype=np.int32)])
    rng = np.random.default_rng(seed)
    order = rng.permutation(C)                                   # shuffle=True, train.py:76
    indptr, indices, data = cg.A_cg.indptr, cg.A_cg.indices, cg.A_cg.data
    t_total, e_total, nb, err = 0.0, 0, 0, 0.0
    with torch.no_grad():
        while nb < max_batches and t_total < budget_s and nb * batch < C:
            cells = np.sort(order[nb * batch:(nb + 1) * batch])
            t0 = time.perf_counter()
            # block of the batch: every in-edge of the seeds (expand_factor = all nodes, train.py:37-38) + self-loops
            lo, hi = indptr[cells], indptr[cells + 1]
            cnt = hi - lo
            e_dst = np.concatenate([np.repeat(np.arange(len(cells)), cnt), np.arange(len(cells))])
            pos = np.concatenate([np.arange(l, r) for l, r in zip(lo, hi)]) if len(cells) else np.zeros(0, np.int64)
            src_parent = np.concatenate([indices[pos].astype(np.int64), cells.astype(np.int64) + G])
            e_w = torch.from_numpy(np.concatenate([data[pos].astype(np.float32), np.ones(len(cells), np.float32)]))
            layer0, e_src = np.unique(src_parent, return_inverse=True)        # NodeFlow layer 0 = unique sources
            hb = h[torch.from_numpy(layer0)]                                    # copy_from_parent (train.py:79)
            neigh = O.block_compute(hb, e_src, e_dst, e_w, node_id[layer0], node_id[cells + G], len(cells), alpha, G)
            out = O.node_update(neigh, W, b)
            t_total += time.perf_counter() - t0
            e_total += len(e_dst); nb += 1
            if nb == 1:                                                          # restatements agree on these rows
                a = alpha.reshape(-1).numpy()
                Zc = (cg.A_cg[cells] @ (feats[:G] * a[:G, None]) + a[G + 1] * feats[G + cells]) / cg.deg_c[cells, None]
                ref = np.maximum(Zc.astype(np.float32) @ W.numpy().T + b.numpy(), 0)
                err = float(np.abs(out.numpy() - ref).max())
    s_per_edge_float = t_total / max(1, e_total * D_in)
    E_cg, E_gc = cg.A_cg.nnz + C, cg.A_gc.nnz + G
    floats = 0
    d = D_in
    for i in range(n_layers):
        if i < n_layers - 1:
            floats += E_gc * d
        floats += E_cg * d
        d = hidden
    return {"s_per_forward": s_per_edge_float * floats, "batches": nb, "batch": batch, "edges": int(e_total),
            "s_measured": t_total, "s_per_edge_float": s_per_edge_float, "check_err": err}
