"""Generates the committed golden fixtures.  Run in the BUILD container only
(`python tests/golden/make_golden.py`): it reads the reference's demo DATA files
(/root/reference/test/...), never its source, and runs the repo's own CPU
restatement (oracle/wgnn_oracle.py) on them.

Outputs (tests/golden/):
  kat_2x2.json          hand-derived known answer - written by hand below, NOT computed
  testis199.npz         expression CSR of mouse_Testis199 (199 cells x 9339 genes, 75,168 nnz) + oracle outputs
  pancreas11.npz        expression CSR of human_Pancreas11 (11 cells x 2650 genes, dense)   + oracle outputs
DGL 0.4.3 is absent, so THESE vectors pin the restatement against itself over time; the
fixtures that come from executing the reference's own code are made by make_refcode_golden.py.
"""
import json
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import scipy.sparse as sp
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle import wgnn_oracle as O  # noqa: E402

REF = Path("/root/reference")


def golden_features(n_nodes, dim, seed):
    return (0.5 * np.random.default_rng(seed).standard_normal((n_nodes, dim))).astype(np.float32)


def make_case(name, expr, dim, hidden, n_classes, n_layers, support_mask=None):
    C, G = expr.shape
    sd = O.init_params(dim, hidden, n_classes, n_layers, G, seed=7)
    feats = golden_features(G + C, dim, seed=11)
    g = O.build_reference_graph(expr, support_mask)
    logits32, hs32 = O.edgelist_full_forward(sd, g, torch.from_numpy(feats), n_layers, return_hidden=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    logits64 = O.edgelist_full_forward(sd64, g, torch.from_numpy(feats).double(), n_layers)
    cg = O.build_csr_graph(expr, support_mask)
    csr32 = O.csr_forward(sd, cg, feats, n_layers)
    print(f"{name}: edge-list fp32 vs fp64 {np.abs(logits32.numpy() - logits64.numpy()).max():.3e}; "
          f"CSR fp32 vs edge-list fp64 {np.abs(csr32 - logits64.numpy()[G:]).max():.3e}")
    # gradient oracle on a 32-seed batch
    seeds = np.arange(G, G + min(C, 32))
    labels = torch.from_numpy(np.random.default_rng(3).integers(0, n_classes, len(seeds)))
    loss, grads, _ = O.loss_and_grads(sd, g, torch.from_numpy(feats), seeds, labels, n_layers)
    out = dict(indptr=expr.indptr.astype(np.int64), indices=expr.indices.astype(np.int32), data=expr.data.astype(np.float32),
               shape=np.array(expr.shape), dim=dim, hidden=hidden, n_classes=n_classes, n_layers=n_layers,
               feat_seed=11, feat_checksum=float(feats.astype(np.float64).sum()),
               logits_f32=logits32.numpy()[G:], logits_f64=logits64.numpy()[G:],
               hidden_last_f32=hs32[-1].numpy()[G:], labels=labels.numpy(), seeds=seeds, loss=float(loss),
               support_mask=np.ones(C, bool) if support_mask is None else support_mask)
    for k, v in sd.items():
        out["param." + k] = v.numpy()
    for k, v in grads.items():
        out["grad." + k] = v.numpy()
    np.savez_compressed(HERE / f"{name}.npz", **out)


def main():
    # ---- hand-derived KAT (derived on paper from preprocess_internal.py:17-23,170-173,213-214
    #      and gnn.py:47-56,65).  D = 1.
    kat = {
        "genes": 2, "cells": 2,
        "expr_rows": [[1.0, 3.0], [2.0, 0.0]],            # cell0 = {g0:1, g1:3}, cell1 = {g0:2}
        "alpha": [2.0, 0.5, 3.0, 0.25],                    # [a_g0, a_g1, a_G (gene self), a_G+1 (cell self)]
        "features": [1.0, 10.0, 100.0, 1000.0],            # nodes g0, g1, c0, c1
        "norm_w_into_g0": {"c0": 2.0 / 3.0, "c1": 4.0 / 3.0},
        "norm_w_into_g1": {"c0": 1.0},
        "norm_w_into_c0": {"g0": 0.5, "g1": 1.5},
        "norm_w_into_c1": {"g0": 1.0},
        "neigh": {"g0": 2803.0 / 3.0, "g1": 40.0, "c0": 33.5 / 3.0, "c1": 126.0},
    }
    (HERE / "kat_2x2.json").write_text(json.dumps(kat, indent=1))

    # ---- mouse_Testis199: (genes x cells) csv.gz -> (cells x genes) CSR of values > 0 (threshold 0, train.py:147)
    df = pd.read_csv(REF / "test/mouse/mouse_Testis199_data.gz", compression="gzip", index_col=0)
    arr = df.to_numpy(dtype=np.float32).T
    expr = sp.csr_matrix(np.where(arr > 0, arr, 0).astype(np.float32)); expr.sort_indices()
    print("testis199", expr.shape, expr.nnz)
    # predict-graph asymmetry (preprocess.py:184-187): last 40 cells act as test cells (gene->cell edges only)
    mask = np.ones(expr.shape[0], bool); mask[-40:] = False
    make_case("testis199", expr, dim=32, hidden=24, n_classes=8, n_layers=2, support_mask=mask)

    df = pd.read_csv(REF / "test/human/human_Pancreas11_data.csv", index_col=0)
    arr = df.to_numpy(dtype=np.float32).T
    expr = sp.csr_matrix(np.where(arr > 0, arr, 0).astype(np.float32)); expr.sort_indices()
    print("pancreas11", expr.shape, expr.nnz)
    make_case("pancreas11", expr, dim=8, hidden=12, n_classes=4, n_layers=2)


if __name__ == "__main__":
    main()
