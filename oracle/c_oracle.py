"""ctypes loader for oracle/liboracle.so (TEST INFRASTRUCTURE ONLY; see wgnn_oracle.c)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liboracle.so"


def build(force=False):
    # built for x86-64-v3 (AVX2+FMA) so the .so made in the build container also runs on the GPU box host
    if force or not LIB.exists() or LIB.stat().st_mtime < (HERE / "wgnn_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "-B", "liboracle.so"], check=True, capture_output=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def aggregate(rowptr, col, val, alpha, mode, self_idx, h_src, h_self):
    rowptr = np.ascontiguousarray(rowptr, np.int32); col = np.ascontiguousarray(col, np.int32)
    val = np.ascontiguousarray(val, np.float32); alpha = np.ascontiguousarray(alpha, np.float32).ravel()
    h_src = np.ascontiguousarray(h_src, np.float32); h_self = np.ascontiguousarray(h_self, np.float32)
    R, D = h_self.shape
    out = np.empty((R, D), np.float32)
    lib().oracle_aggregate(_p(rowptr), _p(col), _p(val), _p(alpha), C.c_int(mode), C.c_int32(self_idx),
                           _p(h_src), C.c_int64(h_src.shape[1]), _p(h_self), C.c_int64(D), _p(out), C.c_int64(D),
                           C.c_int64(R), C.c_int32(D))
    return out


def aggregate_blocked(rowptr, col, val, alpha, mode, self_idx, h_src, h_self, tile_rows=None, block_rows=256):
    """Cache-blocked form of :func:`aggregate` (tile of destination rows x block of source rows, see wgnn_oracle.c)."""
    rowptr = np.ascontiguousarray(rowptr, np.int32); col = np.ascontiguousarray(col, np.int32)
    val = np.ascontiguousarray(val, np.float32); alpha = np.ascontiguousarray(alpha, np.float32).ravel()
    h_src = np.ascontiguousarray(h_src, np.float32); h_self = np.ascontiguousarray(h_self, np.float32)
    R, D = h_self.shape
    if tile_rows is None:          # measured on the GPU box's host (profiles/r03_issue_analysis.md): 128 for the many-row cell side
        tile_rows = 128 if R >= 40_000 else 256
    out = np.empty((R, D), np.float32)
    pre = np.empty_like(h_src) if mode == 0 else h_src
    lib().oracle_aggregate_blocked(_p(rowptr), _p(col), _p(val), _p(alpha), C.c_int(mode), C.c_int32(self_idx),
                                   _p(h_src), C.c_int64(h_src.shape[1]), _p(h_self), C.c_int64(D), _p(out), C.c_int64(D),
                                   C.c_int64(R), C.c_int64(h_src.shape[0]), C.c_int32(D), C.c_int32(tile_rows),
                                   C.c_int32(block_rows), _p(pre))
    return out


def normalize_rows(rowptr, val):
    rowptr = np.ascontiguousarray(rowptr, np.int32); val = np.ascontiguousarray(val, np.float32)
    out = np.empty_like(val)
    lib().oracle_normalize_rows(_p(rowptr), _p(val), _p(out), C.c_int64(len(rowptr) - 1))
    return out


def num_threads():
    return int(lib().oracle_num_threads())


def forward(sd, cg, features, n_layers, order="aggregate_first"):
    """2-layer (n-layer) full-graph forward.  ``aggregate_first`` = the reference's order (aggregate, then Linear+ReLU via
    torch, as the reference's nn.Linear does on CPU); ``project_first`` = the cheaper algebraically equal order the GPU
    path uses when H <= D_in (P = h W^T once, aggregate the H-wide rows, + bias, ReLU) - timed as the stronger CPU
    baseline.  cg: oracle.wgnn_oracle.CsrGraph.  Returns logits of all cells."""
    import torch
    import torch.nn.functional as F
    G = cg.num_genes
    alpha = sd["alpha"].detach().numpy().ravel().astype(np.float32)
    Hg = np.ascontiguousarray(features[:G], np.float32); Hc = np.ascontiguousarray(features[G:], np.float32)
    A_cg, A_gc = cg.A_cg, cg.A_gc
    if order in ("project_first", "project_first_blocked"):
        agg = aggregate if order == "project_first" else aggregate_blocked
        for i in range(n_layers):
            last = i == n_layers - 1
            W, b = sd[f"layers.{i}.fc_neigh.weight"].float(), sd[f"layers.{i}.fc_neigh.bias"].float().numpy()
            Pg = F.linear(torch.from_numpy(Hg), W).numpy(); Pc = F.linear(torch.from_numpy(Hc), W).numpy()
            Zc = agg(A_cg.indptr, A_cg.indices, A_cg.data, alpha, 0, G + 1, Pg, Pc)
            if not last:
                Zg = agg(A_gc.indptr, A_gc.indices, A_gc.data, alpha, 1, G, Pc, Pg)
                Hg = np.maximum(Zg + b, 0, out=Zg)
            Hc = np.maximum(Zc + b, 0, out=Zc)
        return F.linear(torch.from_numpy(Hc), sd["linear.weight"].float(), sd["linear.bias"].float()).numpy()
    for i in range(n_layers):
        last = i == n_layers - 1
        W, b = sd[f"layers.{i}.fc_neigh.weight"].float(), sd[f"layers.{i}.fc_neigh.bias"].float()
        Zc = aggregate(A_cg.indptr, A_cg.indices, A_cg.data, alpha, 0, G + 1, Hg, Hc)
        if not last:
            Zg = aggregate(A_gc.indptr, A_gc.indices, A_gc.data, alpha, 1, G, Hc, Hg)
            Hg = F.relu(F.linear(torch.from_numpy(Zg), W, b)).numpy()
        Hc = F.relu(F.linear(torch.from_numpy(Zc), W, b)).numpy()
    return F.linear(torch.from_numpy(Hc), sd["linear.weight"].float(), sd["linear.bias"].float()).numpy()
