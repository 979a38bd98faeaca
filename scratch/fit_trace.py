"""Mini-batch training through DeepSortClassifier.fit on a synthetic tissue (default reference shape: 1 layer,
hidden 200, dense_dim 400 would need >= 400 cells for the PCA; here dense_dim 64) - run under
`rocprofv3 --hip-trace --stats` to count host synchronisations per batch.
usage: python scratch/fit_trace.py [cells] [genes] [batch] [epochs] [graph_steps 0|1] [num_neighbors]"""
import sys, time, tempfile
from pathlib import Path
import numpy as np, pandas as pd, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import scdeepsort_amd as sda

cells, genes_n, batch, epochs, graphed, nnb = (int(x) for x in (sys.argv[1:] + [2000, 3000, 64, 4, 1, 0][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
genes = [f"G{i}" for i in range(genes_n)]
n_types = 6
prog = rng.random((n_types, genes_n)) < 0.08
types = rng.integers(0, n_types, cells)
X = np.zeros((genes_n, cells), np.float32)
for j, t in enumerate(types):
    on = rng.random(genes_n) < (0.02 + 0.5 * prog[t])
    X[on, j] = np.clip(rng.normal(3.0, 0.9, on.sum()), 0.5, 7.0)
tmp = Path(tempfile.mkdtemp())
names = [f"C{j}" for j in range(cells)]
pd.DataFrame(X, index=genes, columns=names).to_csv(tmp / "d.csv")
pd.DataFrame({"Cell": names, "Cell_type": [f"type{t}" for t in types]}).to_csv(tmp / "c.csv")
clf = sda.DeepSortClassifier("mouse", "Synth", dense_dim=64, hidden_dim=200, batch_size=batch, n_epochs=epochs, n_layers=1,
                             random_seed=1, gpu_id=0, dropout=0.1, num_neighbors=nnb or None)
clf.graph_steps = bool(graphed)
torch.cuda.synchronize(); t0 = time.perf_counter()
clf.fit([(tmp / "d.csv", tmp / "c.csv")])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
n_train = cells - int(cells * 0.1)
nb = -(-n_train // batch)
print(f"fit: {cells} cells x {genes_n} genes, batch {batch}: {len(clf.history)} epochs x {nb} batches in {dt:.2f} s "
      f"(incl. ingest); num_neighbors={nnb} graphed={graphed} replays={getattr(clf._step, 'replays', 0)}; "
      f"last epoch: {clf.history[-1]}")
