"""Synthetic single-cell expression graphs of the shape BASELINE.json names.

Statistics follow the reference's demo file ``test/mouse/mouse_Testis199_data.gz`` (199 cells x 9339 genes, 4.04 % dense):

* RNG ``numpy.random.default_rng(seed)``, reference default seed 10086 (``train.py:128``) - the SAME graph on every host,
  with or without a GPU; per-chunk child streams (``Generator.spawn``) make the result independent of the thread count;
* per-cell non-zero count ``k_c = clip(round(lognormal(ln(density*G) - s^2/2, s = 0.52)), 16, G)``;
* gene popularity (``popularity="testis199"``, the default): the fraction of cells a gene of popularity rank r is expressed
  in follows the demo file's curve - 0.98 / 0.67 / 0.32 / 0.09 / 0.035 at rank 1 / 10 / 100 / 1000 / 3000 of 9339 (a
  rank^-0.9 law whose head is flattened: the top gene cannot be in more than every cell), the median gene in 1.5 % of the
  cells.  For G != 9339 the curve is applied at RELATIVE rank r*9339/G: the inclusion probabilities must sum to the mean
  non-zero count per cell (density*G), which pins the curve's integral - keeping the absolute ranks at G = 20 000 would leave
  510 of the 800 non-zeros per cell to ranks > 3000, i.e. a flat 3 % tail instead of "the median gene in 1-2 % of cells";
* a cell's genes are drawn WITHOUT replacement proportionally to per-gene weights (exponential clocks: gene g fires at
  ``Exp(1)/w_g``, the k_c earliest are kept == successive sampling ~ w); the weights are solved so that the inclusion
  probability, averaged over the k_c law, is the curve above (``sampling_weights``);
* values ``clip(N(3.0, 0.9), 0.5, 7.0)`` fp32; gene ids shuffled so hub genes are scattered over the id range as in real data.

``popularity="dense_head"`` keeps rounds 1-5's generator (weights = rank^-0.9 taken literally, torch's generator on the
target device): its head is denser (inclusion 0.99 / 0.61 at rank 10 / 100); kept for A/B only.
"""
from __future__ import annotations

import functools
import math
import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

REFERENCE_SEED = 10086          # train.py:128 default --random_seed


@dataclass
class Config:
    name: str
    cells: int
    genes: int
    hidden: int
    dense_dim: int = 400        # train.py:141
    n_classes: int = 16
    n_layers: int = 2
    density: float = 0.04
    feature_dtype: torch.dtype = torch.float32     # storage type of the node features (cfg5: fp16)
    total_cells: bool = False                      # True: `cells` is the WHOLE job, sharded over the ranks (strong scaling)


CONFIGS = {
    "tiny": Config("tiny", 512, 256, 32, dense_dim=40),
    "tiny_atlas": Config("tiny_atlas", 3_001, 256, 32, dense_dim=40, feature_dtype=torch.float16, total_cells=True),   # cfg5's shape in small
    "cfg2": Config("cfg2", 10_000, 5_000, 128),
    "cfg3": Config("cfg3", 100_000, 20_000, 256),
    "cfg5": Config("cfg5", 764_741, 20_000, 256, feature_dtype=torch.float16, total_cells=True),
}


# Inclusion frequency by popularity rank in mouse_Testis199 (9339 genes, 199 cells; measured from the demo matrix, smoothed
# over a +-30 % rank window).  SURVEY 8d quotes the five ranks 1 / 10 / 100 / 1000 / 3000.
TESTIS_GENES = 9339
_TESTIS_RANK = (1, 2, 3, 5, 10, 20, 30, 50, 100, 200, 300, 500, 1000, 1500, 2000, 3000, 4000, 5000, 6000, 7500, 9339)
_TESTIS_INCL = (0.98, 0.95, 0.87, 0.77, 0.67, 0.56, 0.50, 0.42, 0.32, 0.236, 0.196, 0.151, 0.090, 0.065, 0.050, 0.035,
                0.0215, 0.015, 0.010, 0.0062, 0.005)
NNZ_LOG_STD = 0.52
MIN_NNZ = 16


def inclusion_curve(genes: int, density: float = 0.04) -> np.ndarray:
    """Target inclusion probability of the gene of popularity rank 1..genes (float64, descending): Testis199's curve at
    relative rank, scaled by one factor (capped at 0.98) so that it sums to ``density * genes``."""
    x = np.arange(1, genes + 1, dtype=np.float64) * (TESTIS_GENES / genes)
    pi0 = np.exp(np.interp(np.log(np.clip(x, 1.0, TESTIS_GENES)), np.log(_TESTIS_RANK), np.log(_TESTIS_INCL)))
    target = min(density * genes, 0.98 * genes)
    lo, hi = 0.0, 1.0
    while np.minimum(0.98, hi * pi0).sum() < target and hi < 1e6:
        hi *= 2
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if np.minimum(0.98, mid * pi0).sum() < target:
            lo = mid
        else:
            hi = mid
    return np.minimum(0.98, hi * pi0)


def _nnz_quantiles(genes: int, density: float, m: int = 24) -> np.ndarray:
    """m equally likely values of the per-cell non-zero count law (the mid-quantiles of the clipped log-normal)."""
    from statistics import NormalDist
    z = np.array([NormalDist().inv_cdf((i + 0.5) / m) for i in range(m)])
    mu = math.log(density * genes) - NNZ_LOG_STD ** 2 / 2
    ks = np.clip(np.exp(mu + NNZ_LOG_STD * z), min(MIN_NNZ, genes), 0.999 * genes)
    return ks * (min(density, 0.98) * genes / ks.mean())           # mid-quantiles lose the tails' share of the mean: restore it


@functools.lru_cache(maxsize=8)
def sampling_weights(genes: int, density: float = 0.04) -> np.ndarray:
    """Per-rank weights w such that drawing k_c genes without replacement ~ w gives the gene of rank r the inclusion
    probability ``inclusion_curve(genes)[r]`` on average over cells.  With exponential clocks a gene is among the k earliest with
    probability 1 - exp(-t_k * w) (t_k = the k-th firing time, sharply concentrated for k >> 1), t_k solving
    sum_g 1 - exp(-t_k w_g) = k; fixed point on w in the rate domain."""
    pi = inclusion_curve(genes, density)
    ks = _nnz_quantiles(genes, density)
    w = -np.log1p(-pi)
    for _ in range(40):
        t = np.ones_like(ks)
        for _ in range(50):                                        # Newton on t (one root: the left side is concave increasing)
            e = np.exp(-np.outer(t, w))
            f = (1.0 - e).sum(1) - ks
            t = np.maximum(t - f / np.maximum((e * w).sum(1), 1e-300), t * 0.1)
            if np.abs(f).max() < 1e-9 * genes:
                break
        got = (1.0 - np.exp(-np.outer(t, w))).mean(0)
        got = np.clip(got, 1e-300, 1 - 1e-15)
        step = np.log1p(-pi) / np.log1p(-got)
        w = w * step
        if np.abs(step - 1).max() < 1e-6:
            break
    return w


def _host_threads() -> int:
    """Generator threads of THIS process: the host's cores shared among the ranks of the node (LOCAL_WORLD_SIZE), <= 64."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    return max(1, min(64, cores // ranks))


def _synth_expression_numpy(cells: int, genes: int, density: float, seed: int, shuffle_genes: bool,
                            chunk_cells: int, cell_range: Optional[Tuple[int, int]] = None
                            ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``cell_range = (lo, hi)``: only the rows lo .. hi-1 of the SAME matrix (a rank's shard of one job): the per-cell counts and
    the chunks' child streams are drawn for the whole matrix (cheap), the per-(cell, gene) work only for the chunks the range meets."""
    rng = np.random.default_rng(seed)
    mu = math.log(density * genes) - NNZ_LOG_STD ** 2 / 2
    k = np.rint(rng.lognormal(mu, NNZ_LOG_STD, size=cells)).astype(np.int64)
    k = np.clip(k, min(MIN_NNZ, genes), genes)
    gene_of_rank = rng.permutation(genes) if shuffle_genes else np.arange(genes)
    inv_w = np.empty(genes, dtype=np.float32)
    inv_w[gene_of_rank] = (1.0 / sampling_weights(genes, density)).astype(np.float32)     # indexed by gene id
    starts = list(range(0, cells, chunk_cells))
    children = rng.spawn(len(starts))

    def chunk(i):
        c0 = starts[i]
        kc = k[c0:c0 + chunk_cells]
        n = kc.shape[0]
        g = children[i]
        keys = g.standard_exponential(size=(n, genes), dtype=np.float32)
        keys *= inv_w                                              # firing time of gene g in this cell
        kmax = int(kc.max())
        early = np.partition(keys, kmax - 1, axis=1)[:, :kmax] if kmax < genes else keys.copy()
        early.sort(axis=1)
        thr = early[np.arange(n), kc - 1]                          # the k_c-th firing time of each cell
        del early
        mask = keys <= thr[:, None]
        for r in np.flatnonzero(mask.sum(1) != kc):                # fp32 ties at the threshold: keep exactly k_c, lowest id first
            mask[r] = False
            mask[r, np.argsort(keys[r], kind="stable")[:kc[r]]] = True
        col = np.nonzero(mask)[1].astype(np.int32)                 # ascending gene id inside each cell
        val = np.clip(g.normal(3.0, 0.9, size=col.shape[0]), 0.5, 7.0).astype(np.float32)
        return col, val

    lo, hi = (0, cells) if cell_range is None else (max(0, int(cell_range[0])), min(cells, int(cell_range[1])))
    wanted = [i for i, c0 in enumerate(starts) if c0 < hi and c0 + chunk_cells > lo]
    nthreads = min(_host_threads(), len(wanted))
    if nthreads > 1:
        with ThreadPoolExecutor(nthreads) as pool:
            parts = list(pool.map(chunk, wanted))
    else:
        parts = [chunk(i) for i in wanted]
    col = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, np.int32)
    val = np.concatenate([p[1] for p in parts]) if parts else np.zeros(0, np.float32)
    if cell_range is not None:                                     # cut the first / last chunk down to the range
        first = starts[wanted[0]] if wanted else lo
        kk = k[first:min(cells, (starts[wanted[-1]] + chunk_cells) if wanted else lo)]
        off = np.zeros(kk.shape[0] + 1, dtype=np.int64); np.cumsum(kk, out=off[1:])
        col, val = col[off[lo - first]:off[hi - first]], val[off[lo - first]:off[hi - first]]
        k = k[lo:hi]
    rowptr = np.zeros(k.shape[0] + 1, dtype=np.int64)
    np.cumsum(k, out=rowptr[1:])
    return rowptr, col, val


def synth_expression(cells: int, genes: int, density: float = 0.04, seed: int = REFERENCE_SEED,
                     device: torch.device | str = "cpu", shuffle_genes: bool = True,
                     chunk_cells: int | None = None, popularity: str | None = None,
                     cell_range: Optional[Tuple[int, int]] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CSR (rowptr int64 [C+1], col int32 sorted per row, val float32) of a (cells x genes) expression matrix, resident on
    ``device``.  ``popularity``: "testis199" (default; drawn on the host with numpy) or "dense_head" (rounds 1-5);
    the default can be switched with WGNN_SYNTH_POPULARITY.  ``cell_range = (lo, hi)``: the CSR of rows lo .. hi-1 of that matrix
    only (rowptr re-based to 0) - what one rank of a cell-sharded job needs; identical to slicing the whole matrix."""
    popularity = popularity or os.environ.get("WGNN_SYNTH_POPULARITY", "testis199")
    if popularity == "dense_head":
        rp, col, val = _synth_expression_dense_head(cells, genes, density, seed, device, shuffle_genes, chunk_cells or 4096)
        if cell_range is not None:
            lo, hi = int(cell_range[0]), int(cell_range[1])
            b, e = int(rp[lo]), int(rp[hi])
            rp, col, val = (rp[lo:hi + 1] - rp[lo]).clone(), col[b:e].clone(), val[b:e].clone()
        return rp, col, val
    if popularity != "testis199":
        raise ValueError(f"unknown popularity law {popularity!r} (testis199 | dense_head)")
    if chunk_cells is None:
        chunk_cells = max(16, min(1024, (1 << 24) // max(genes, 1)))   # <= 64 MiB of keys per chunk
    rowptr, col, val = _synth_expression_numpy(cells, genes, density, seed, shuffle_genes, chunk_cells, cell_range)
    device = torch.device(device)
    return (torch.from_numpy(rowptr).to(device), torch.from_numpy(col).to(device), torch.from_numpy(val).to(device))


def _synth_expression_dense_head(cells, genes, density, seed, device, shuffle_genes, chunk_cells):
    """Rounds 1-5: Gumbel top-k ~ rank^-0.9 with torch's generator ON the device (CPU and GPU give different graphs)."""
    device = torch.device(device)
    gen = torch.Generator(device=device).manual_seed(seed)
    sigma = 0.52
    mu = math.log(density * genes) - sigma * sigma / 2
    k = torch.exp(mu + sigma * torch.randn(cells, generator=gen, device=device)).round().long()
    k = k.clamp_(min=min(16, genes), max=genes)
    logw = -0.9 * torch.log(torch.arange(1, genes + 1, device=device, dtype=torch.float32))
    if shuffle_genes:
        logw = logw[torch.randperm(genes, generator=gen, device=device)]
    cols, counts = [], []
    for c0 in range(0, cells, chunk_cells):
        kc = k[c0:c0 + chunk_cells]
        n = kc.shape[0]
        u = torch.rand(n, genes, generator=gen, device=device).clamp_(1e-12, 1 - 1e-7)
        keys = logw.unsqueeze(0) - torch.log(-torch.log(u))             # Gumbel top-k == weighted sampling w/o replacement
        del u
        kmax = int(kc.max())
        top = torch.topk(keys, kmax, dim=1, sorted=True).indices
        del keys
        keep = torch.arange(kmax, device=device).unsqueeze(0) < kc.unsqueeze(1)
        mask = torch.zeros(n, genes, dtype=torch.bool, device=device)
        mask.scatter_(1, top, keep)
        cols.append(mask.nonzero()[:, 1].to(torch.int32))
        counts.append(kc)
        del mask, top, keep
    col = torch.cat(cols)
    rowptr = torch.zeros(cells + 1, dtype=torch.int64, device=device)
    torch.cumsum(torch.cat(counts), 0, out=rowptr[1:])
    val = (3.0 + 0.9 * torch.randn(col.shape[0], generator=gen, device=device)).clamp_(0.5, 7.0)
    return rowptr, col, val.float()


def synth_features(n_nodes: int, dim: int, seed: int = REFERENCE_SEED + 1, device="cpu",
                   dtype=torch.float32, rng: str | None = None) -> torch.Tensor:
    """Node features ~ 0.5 * N(0, 1), fp32 or fp16 storage.  ``rng="numpy"`` (default with the testis199 law):
    ``numpy.random.default_rng(seed)`` on the host - the same rows on every machine, with or without a GPU; blocks of 65 536 rows
    from child streams, so the result does not depend on the thread count.  ``rng="torch"`` (the dense_head A/B workload of rounds
    1-5): torch's generator on the target device."""
    rng = rng or ("torch" if os.environ.get("WGNN_SYNTH_POPULARITY", "testis199") == "dense_head" else "numpy")
    if rng == "torch":
        gen = torch.Generator(device=torch.device(device)).manual_seed(seed)
        return (0.5 * torch.randn(n_nodes, dim, generator=gen, device=device)).to(dtype)
    if rng != "numpy":
        raise ValueError(f"unknown feature rng {rng!r} (numpy | torch)")
    block = 65536
    starts = list(range(0, n_nodes, block))
    children = np.random.default_rng(seed).spawn(len(starts))
    out = np.empty((n_nodes, dim), dtype=np.float32)

    def fill(i):
        r0 = starts[i]
        n = min(block, n_nodes - r0)
        children[i].standard_normal(size=(n, dim), dtype=np.float32, out=out[r0:r0 + n])

    nthreads = min(_host_threads(), len(starts))
    if nthreads > 1:
        with ThreadPoolExecutor(nthreads) as pool:
            list(pool.map(fill, range(len(starts))))
    else:
        for i in range(len(starts)):
            fill(i)
    out *= 0.5
    return torch.from_numpy(out).to(device=torch.device(device), dtype=dtype)


def to_scipy(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, genes: int):
    import scipy.sparse as sp
    return sp.csr_matrix((val.cpu().numpy(), col.cpu().numpy(), rowptr.cpu().numpy()),
                         shape=(rowptr.shape[0] - 1, genes))
