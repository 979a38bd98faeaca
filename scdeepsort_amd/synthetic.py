"""Synthetic single-cell expression graphs of the shape BASELINE.json names.

Statistics follow the reference's demo file ``test/mouse/mouse_Testis199_data.gz``
(199 cells x 9339 genes, 4.04 % dense): per-cell non-zero count log-normal
(log-std 0.52) with mean ``density*G``; gene popularity ~ rank^-0.9 (hub genes
expressed in nearly every cell, the median gene in 1-2 % of cells); values
``clip(N(3.0, 0.9), 0.5, 7.0)``.  Genes of a cell are drawn without replacement
proportionally to popularity (Gumbel top-k).  Gene ids are shuffled so that hub
genes are scattered over the id range as in real data.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Tuple

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


def synth_expression(cells: int, genes: int, density: float = 0.04, seed: int = REFERENCE_SEED,
                     device: torch.device | str = "cpu", shuffle_genes: bool = True,
                     chunk_cells: int = 4096) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CSR (rowptr int64 [C+1], col int32 sorted per row, val float32) of a (cells x genes) expression matrix."""
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
                   dtype=torch.float32) -> torch.Tensor:
    gen = torch.Generator(device=torch.device(device)).manual_seed(seed)
    return (0.5 * torch.randn(n_nodes, dim, generator=gen, device=device)).to(dtype)


def to_scipy(rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, genes: int):
    import scipy.sparse as sp
    return sp.csr_matrix((val.cpu().numpy(), col.cpu().numpy(), rowptr.cpu().numpy()),
                         shape=(rowptr.shape[0] - 1, genes))
