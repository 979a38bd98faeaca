"""CPU tests of the host-side mirror of the reference interface (no kernels launched)."""
import io

import numpy as np
import torch
import torch.nn.functional as F

import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S
from oracle import wgnn_oracle as O


def test_gnn_state_dict_matches_reference_contract():
    """Key names / shapes implied by models/gnn.py:13,37-45 so released checkpoints load."""
    m = sda.GNN(in_feats=400, n_hidden=200, n_classes=14, n_layers=2, gene_num=1000, activation=F.relu, dropout=0.1)
    sd = m.state_dict()
    want = {"layers.0.fc_neigh.weight": (200, 400), "layers.0.fc_neigh.bias": (200,),
            "layers.1.fc_neigh.weight": (200, 200), "layers.1.fc_neigh.bias": (200,),
            "alpha": (1002, 1), "linear.weight": (14, 200), "linear.bias": (14,)}
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    assert torch.all(sd["alpha"] == 1)                                   # gnn.py:43
    assert isinstance(m.dropout, torch.nn.Dropout) and m.dropout.p == 0.1
    assert sda.GNN(8, 8, 2, 1, 4).dropout is None                        # gnn.py:33-36
    # xavier_uniform with relu gain bound (gnn.py:16,45)
    bound = (2 ** 0.5) * (6 / (400 + 200)) ** 0.5
    assert sd["layers.0.fc_neigh.weight"].abs().max() <= bound + 1e-6


def test_checkpoint_roundtrip_reference_format():
    """{'model': state_dict, 'optimizer': ...} as written by train.py:117-123 and read by predict.py:56-59."""
    m = sda.GNN(16, 8, 3, 1, 10, activation=F.relu)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    buf = io.BytesIO()
    torch.save({"model": m.state_dict(), "optimizer": opt.state_dict()}, buf)
    buf.seek(0)
    state = torch.load(buf, map_location="cpu")
    m2 = sda.GNN(16, 8, 3, 1, 10, activation=F.relu)
    m2.load_state_dict(state["model"])
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    # an oracle-initialised state_dict (reference key set) loads strictly
    sd = O.init_params(16, 8, 3, 1, 10, seed=1)
    m2.load_state_dict(sd, strict=True)


def test_synthetic_generator_statistics():
    rp, col, val = S.synth_expression(2000, 1500, seed=1)
    k = (rp[1:] - rp[:-1]).float()
    assert abs(k.mean().item() / (0.04 * 1500) - 1) < 0.1
    assert k.min() >= 16
    # sorted, unique columns per row
    for r in (0, 7, 1999):
        c = col[rp[r]:rp[r + 1]]
        assert torch.all(c[1:] > c[:-1])
    pop = torch.bincount(col.long(), minlength=1500).float() / 2000
    assert pop.max() > 0.9 and pop.median() < 0.05
    assert val.min() >= 0.5 and val.max() <= 7.0 and abs(val.mean().item() - 3.0) < 0.1
    rp2, col2, val2 = S.synth_expression(2000, 1500, seed=1)
    assert torch.equal(col, col2) and torch.equal(val, val2)            # deterministic


def test_shard_ranges_cover_cells():
    from scdeepsort_amd.dist import shard_range
    for C, W in ((100_000, 8), (764_741, 8), (10, 4), (7, 8)):
        spans = [shard_range(C, r, W) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == C
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and b >= a
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
