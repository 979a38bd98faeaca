import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import scipy.sparse as sp
    import torch
    z = np.load(GOLDEN / f"{name}.npz")
    expr = sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    sd = {k[len("param."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    grads = {k[len("grad."):]: z[k] for k in z.files if k.startswith("grad.")}
    C, G = expr.shape
    feats = (0.5 * np.random.default_rng(int(z["feat_seed"])).standard_normal((G + C, int(z["dim"])))).astype(np.float32)
    assert abs(float(feats.astype(np.float64).sum()) - float(z["feat_checksum"])) < 1e-6, "feature RNG drifted"
    return dict(z=z, expr=expr, sd=sd, grads=grads, feats=feats, C=C, G=G, n_layers=int(z["n_layers"]),
                support_mask=z["support_mask"])


def small_case(cells=96, genes=64, dim=24, hidden=16, n_classes=5, n_layers=2, seed=0, density=0.15,
               empty_rows=True, test_cells=8):
    """Random ragged expression matrix with hub genes, empty cells and unexpressed genes."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    pop = np.minimum(1.0, density * genes / np.arange(1, genes + 1) ** 0.9 / np.sum(1 / np.arange(1, genes + 1) ** 0.9) * 1.0)
    pop[0] = 0.98
    mask = rng.random((cells, genes)) < pop[None, :]
    if empty_rows:
        mask[3, :] = False          # a cell with no expressed gene
        mask[:, 5] = False          # a gene no cell expresses
    vals = np.clip(rng.normal(3.0, 0.9, size=mask.shape), 0.5, 7.0).astype(np.float32)
    expr = sp.csr_matrix(np.where(mask, vals, 0).astype(np.float32))
    expr.sort_indices()
    support = np.ones(cells, bool)
    if test_cells:
        support[-test_cells:] = False
    feats = (0.5 * rng.standard_normal((genes + cells, dim))).astype(np.float32)
    return dict(expr=expr, support_mask=support, feats=feats, C=cells, G=genes, dim=dim, hidden=hidden,
                n_classes=n_classes, n_layers=n_layers)
