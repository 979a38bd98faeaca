"""Known-answer tests derived in EXACT RATIONAL arithmetic, straight from the reference's lines - no numpy, no torch,
no code shared with ``oracle/`` or with the DGL stand-in of ``make_refcode_golden.py``:

    python tests/golden/make_kat_rational.py    ->  tests/golden/kat_2layer_predict.json

Every quantity below is a ``fractions.Fraction``; the derivation follows the cited lines one by one, per node, with
explicit loops over edges (the way one does it on paper).

Graph (predict-graph shape, utils/preprocess.py:102-134,167-187; node order preprocess_internal.py:107-110,160-168):
  genes g0..g2 = nodes 0..2 (``id`` = 0,1,2), cells c0..c3 = nodes 3..6 (``id`` = -1).  c0..c2 are SUPPORT (training)
  cells, c3 is a TEST cell.  g0 is a hub gene (expressed by every cell), g2 is expressed by the test cell only.

      expr      g0   g1   g2
      c0         1    3    .
      c1         2    .    .
      c2         4    1    .
      c3 (test)  2    2    5

  edges: every entry x[c,g] > 0 gives gene->cell (g -> c, raw weight x); support cells also give cell->gene
  (preprocess.py:126-134); the test cell gets gene->cell ONLY (preprocess.py:184-187).
  normalize_weight (preprocess_internal.py:17-23): for every node i with >= 1 in-edge,  w_e <- in_degree(i) * w_e /
  sum(w over in-edges of i), BEFORE the self-loops; then one self-loop of weight 1 per node (:213-214).  Consequence
  tested here: g2 has NO in-edge (its only expressing cell is a test cell), so neigh(g2) is its self-loop alone.

Forward (models/gnn.py:47-68, train.py:31 activation = relu), two layers, all nodes evaluated at both layers:
  k(e) = src gene id for gene->cell (:51), dst gene id for cell->gene (:52), G for a gene self-loop (:53), G+1 for a cell
  self-loop (:49 default);  m_e = (h[src] * alpha[k(e)]) * w_e (:54,56);
  neigh[v] = sum of m_e over ALL in-edges of v (self-loop included) / number of those edges   (fn.mean, :65)
  h'[v] = relu(W neigh[v] + b)   (NodeUpdate, :18-25);   logits[c] = W_out h2[c] + b_out   (:66-67).
D = dense_dim = 4, hidden = 4, n_classes = 2.  Parameters and features are small rationals chosen so that ReLU clips
some but not all units.  Gradients are not part of this KAT (the refcode_* goldens carry autograd through the executed
reference code).
"""
import json
from fractions import Fraction as Fr
from pathlib import Path

G, C = 3, 4
N = G + C
EXPR = {(0, 0): 1, (0, 1): 3, (1, 0): 2, (2, 0): 4, (2, 1): 1, (3, 0): 2, (3, 1): 2, (3, 2): 5}     # (cell, gene) -> raw value
SUPPORT = [True, True, True, False]
ALPHA = [Fr(2), Fr(1, 2), Fr(3, 2), Fr(3), Fr(1, 4)]          # a_g0, a_g1, a_g2, alpha[G] gene self-loop, alpha[G+1] cell self-loop
FEATS = [[Fr(1), Fr(-1), Fr(1, 2), Fr(2)],                    # g0
         [Fr(0), Fr(2), Fr(-1), Fr(1)],                       # g1
         [Fr(3), Fr(1), Fr(1), Fr(-2)],                       # g2
         [Fr(1, 2), Fr(1), Fr(0), Fr(-1)],                    # c0
         [Fr(-2), Fr(1, 2), Fr(1), Fr(1)],                    # c1
         [Fr(1), Fr(1), Fr(1), Fr(1)],                        # c2
         [Fr(0), Fr(-1), Fr(2), Fr(1, 2)]]                    # c3
W1 = [[Fr(1, 2), Fr(-1), Fr(0), Fr(1, 4)], [Fr(1), Fr(1, 2), Fr(-1, 2), Fr(0)],
      [Fr(-1, 4), Fr(0), Fr(1), Fr(1)], [Fr(0), Fr(1), Fr(1, 2), Fr(-1)]]
B1 = [Fr(1, 10), Fr(-1, 5), Fr(0), Fr(3, 10)]
W2 = [[Fr(1), Fr(0), Fr(-1, 2), Fr(1, 2)], [Fr(-1), Fr(1, 2), Fr(1), Fr(0)],
      [Fr(1, 4), Fr(1, 4), Fr(-1), Fr(1)], [Fr(0), Fr(-1, 2), Fr(1, 2), Fr(1)]]
B2 = [Fr(-1, 10), Fr(1, 5), Fr(1, 10), Fr(0)]
WO = [[Fr(1), Fr(-1), Fr(1, 2), Fr(0)], [Fr(-1, 2), Fr(1), Fr(0), Fr(2)]]
BO = [Fr(1, 20), Fr(-1, 20)]


def build_edges():
    """(src, dst, raw weight) in the reference's insertion order, then normalised, then self-loops."""
    edges = []
    for (c, g), x in sorted(EXPR.items()):
        if SUPPORT[c]:
            edges.append([G + c, g, Fr(x)])          # cell -> gene
        edges.append([g, G + c, Fr(x)])              # gene -> cell
    for v in range(N):                               # normalize_weight, node by node (preprocess_internal.py:17-23)
        ins = [e for e in edges if e[1] == v]
        if ins:
            s = sum(e[2] for e in ins)
            for e in ins:
                e[2] = len(ins) * e[2] / s
    edges += [[v, v, Fr(1)] for v in range(N)]       # self-loops afterwards (preprocess_internal.py:213-214)
    return edges


def node_id(v):
    return v if v < G else -1


def k_of(src, dst):
    s, d = node_id(src), node_id(dst)                # gnn.py:49-53, later rules override earlier ones
    k = G + 1
    if s >= 0 and d < 0:
        k = s
    if d >= 0 and s < 0:
        k = d
    if d >= 0 and s >= 0:
        k = G
    return k


def layer(h, edges, W, b):
    out = []
    for v in range(N):
        ins = [e for e in edges if e[1] == v]
        neigh = [sum((h[s][j] * ALPHA[k_of(s, d)]) * w for s, d, w in ins) / len(ins) for j in range(len(h[0]))]
        z = [sum(W[o][j] * neigh[j] for j in range(len(neigh))) + b[o] for o in range(len(W))]
        out.append([max(x, Fr(0)) for x in z])
    return out


def main():
    edges = build_edges()
    h1 = layer(FEATS, edges, W1, B1)
    h2 = layer(h1, edges, W2, B2)
    logits = [[sum(WO[o][j] * h2[G + c][j] for j in range(4)) + BO[o] for o in range(2)] for c in range(C)]
    fl = lambda m: [[float(x) for x in r] for r in m]
    st = lambda m: [[str(x) for x in r] for r in m]
    relu_clipped = sum(x == 0 for r in h1 + h2 for x in r)
    assert 0 < relu_clipped < 2 * N * 4, "ReLU must clip some but not all units"
    norm = {f"{'g' if s < G else 'c'}{s if s < G else s - G}->{'g' if d < G else 'c'}{d if d < G else d - G}": str(w)
            for s, d, w in edges if s != d}
    out = dict(genes=G, cells=C, support_mask=SUPPORT,
               expr_rows=[[float(EXPR.get((c, g), 0)) for g in range(G)] for c in range(C)],
               alpha=[float(a) for a in ALPHA], features=fl(FEATS), W1=fl(W1), b1=[float(x) for x in B1],
               W2=fl(W2), b2=[float(x) for x in B2], W_out=fl(WO), b_out=[float(x) for x in BO],
               normalised_weights_exact=norm, h1=fl(h1), h2_cells=fl(h2[G:]), logits=fl(logits),
               logits_exact=st(logits), h1_exact=st(h1), relu_clipped_units=int(relu_clipped))
    path = Path(__file__).resolve().parent / "kat_2layer_predict.json"
    path.write_text(json.dumps(out, indent=1))
    print(path, "logits", fl(logits), "clipped", relu_clipped)


if __name__ == "__main__":
    main()
