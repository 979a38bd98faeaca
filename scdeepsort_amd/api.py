"""Thin counterparts of the documented ``deepsort`` package classes (reference ``docs/api.rst:6-130``), whose
source is not in the reference tree; the tree holds the equivalent CLI classes ``Trainer`` (``train.py:16-123``)
and ``Runner`` (``predict.py:17-152``).  Same constructor keywords, same ``fit`` / ``predict`` signatures, same
bundle on disk (``{species}-{tissue}.pt`` = ``{'model', 'optimizer'}`` (train.py:117-123), ``{tissue}_genes.txt`` /
``{tissue}_cell_type.txt`` written with ``\\r\\n`` (preprocess_internal.py:59-67), ``{species}_{tissue}_data.npz``
support matrix (preprocess_internal.py:180)), read and written either as the reference's
``pretrained/{species}/{models,graphs,statistics}/`` tree or as one flat directory (:class:`BundlePaths`).

Only the per-cell hot path (graph normalisation, aggregation, autograd) runs on the GPU through the HIP
kernels; file ingest, vocabularies and PCA are ordinary host code kept deliberately small.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from ._lib import WgnnError
from .gnn import GNN
from .graph import CellGeneGraph
from .ops import cross_entropy_sum


def _device(gpu_id: int) -> torch.device:
    if not torch.cuda.is_available():
        raise WgnnError("the MI355X path needs a GPU (gpu_id=-1 meant CPU in the reference; there is no CPU fallback here)")
    # ``fit`` / ``_predict`` run under ``torch.cuda.device(dev)`` (restored on exit): C-ABI calls are guarded per call by
    # ``_lib.call``, but hipGraph capture (GraphedTrainStep / GraphedForward) opens its stream on the CURRENT device
    return torch.device("cuda", max(gpu_id, 0))


def _read_expression(path, file_type: str) -> pd.DataFrame:
    """(genes x cells) table as produced by pre-process.R:72-74 -> DataFrame (cells x genes)."""
    if file_type == "csv":
        df = pd.read_csv(path, index_col=0)
    elif file_type == "gz":
        df = pd.read_csv(path, compression="gzip", index_col=0)
    else:
        raise ValueError(f"Not supported type {file_type!r}: csv or gz")
    return df.transpose(copy=True)


def _features(expr: sp.csr_matrix, n_support: int, dense_dim: int, seed, device) -> torch.Tensor:
    """gene_feat = PCA(dense_dim) of the support cells' (genes x cells) matrix (host, sklearn, like the reference);
    cell_feat = rownorm(X) . gene_feat on the device through K1 (preprocess_internal.py:183-202, preprocess.py:194-210)."""
    from sklearn.decomposition import PCA
    dense_sup = expr[:n_support].toarray().astype(np.float64)
    k = min(dense_dim, dense_sup.shape[1], n_support)
    gene_feat = PCA(k, random_state=seed).fit_transform(dense_sup.T)
    if k < dense_dim:
        gene_feat = np.pad(gene_feat, ((0, 0), (0, dense_dim - k)))
    gf = torch.from_numpy(gene_feat.astype(np.float32)).to(device)
    cf = CellGeneGraph.cell_features(torch.from_numpy(expr.indptr.astype(np.int64)).to(device),
                                     torch.from_numpy(expr.indices.astype(np.int32)).to(device),
                                     torch.from_numpy(expr.data.astype(np.float32)).to(device), gf)
    return torch.cat([gf, cf])


def read_xlsx_sheet(path, sheet_name: str) -> List[List[Optional[str]]]:
    """Rows of one worksheet of an .xlsx workbook as lists of strings (None = empty cell).  A minimal reader (zip + XML,
    shared and inline strings, plain numbers) for ``map/celltype2subtype.xlsx`` (predict.py:125-128): neither xlrd nor
    openpyxl is a dependency here."""
    import re
    import zipfile
    import xml.etree.ElementTree as ET
    ns = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main",
          "r": "http://schemas.openxmlformats.org/officeDocument/2006/relationships"}
    with zipfile.ZipFile(path) as z:
        wb = ET.fromstring(z.read("xl/workbook.xml"))
        rid = None
        for sh in wb.find("m:sheets", ns):
            if sh.get("name") == sheet_name:
                rid = sh.get(f"{{{ns['r']}}}id")
        if rid is None:
            raise KeyError(f"no sheet {sheet_name!r} in {path}")
        rels = ET.fromstring(z.read("xl/_rels/workbook.xml.rels"))
        target = next(r.get("Target") for r in rels if r.get("Id") == rid)
        target = target.lstrip("/")
        target = target if target.startswith("xl/") else "xl/" + target
        shared: List[str] = []
        if "xl/sharedStrings.xml" in z.namelist():
            for si in ET.fromstring(z.read("xl/sharedStrings.xml")).findall("m:si", ns):
                shared.append("".join(t.text or "" for t in si.iter(f"{{{ns['m']}}}t")))
        rows: List[List[Optional[str]]] = []
        for row in ET.fromstring(z.read(target)).find("m:sheetData", ns).findall("m:row", ns):
            cells: List[Optional[str]] = []
            for c in row.findall("m:c", ns):
                letters = re.match(r"[A-Z]+", c.get("r", "A")).group(0)
                col = 0
                for ch in letters:
                    col = col * 26 + ord(ch) - 64
                while len(cells) < col:
                    cells.append(None)
                v, t = c.find("m:v", ns), c.get("t")
                if t == "s" and v is not None:
                    val = shared[int(v.text)]
                elif t == "inlineStr":
                    val = "".join(x.text or "" for x in c.iter(f"{{{ns['m']}}}t"))
                else:
                    val = v.text if v is not None else None
                cells[col - 1] = val
            rows.append(cells)
        return rows


_PANDAS_NA = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>", "N/A", "NA",
              "NULL", "NaN", "None", "n/a", "nan", "null"}


def load_label_map(path, species: str) -> Tuple[dict, dict]:
    """old cell type -> (new type, new subtype) from the ``species`` sheet of celltype2subtype.xlsx; empty cells become
    'N/A' (predict.py:125-133)."""
    old2new, old2sub = {}, {}
    for row in read_xlsx_sheet(path, species)[1:]:                     # header=0
        row = (row + [None] * 4)[:4]
        # pandas.read_excel parses its default NA strings as NaN, which the reference then fills with 'N/A'
        _, old, new, sub = [("N/A" if x is None or x in _PANDAS_NA else x) for x in row]
        old2new[old], old2sub[old] = new, sub
    return old2new, old2sub


class BundlePaths:
    """Where the four files of a trained bundle live.  Two layouts:

    * ``reference`` - the tree the reference's CLI writes and every released bundle ships as (``root`` = its ``pretrained``
      directory): ``{root}/{species}/models/{species}-{tissue}.pt`` (train.py:20,117-123; predict.py:57),
      ``{root}/{species}/graphs/{species}_{tissue}_data.npz`` (preprocess_internal.py:79,180; preprocess.py:114),
      ``{root}/{species}/statistics/{tissue}_genes.txt`` / ``{tissue}_cell_type.txt`` (preprocess_internal.py:80,59-67;
      preprocess.py:67-78).  ``root`` may also be the species directory itself (``pretrained/mouse``).
    * ``flat`` - all four files in ``root`` (what ``fit(save_path=...)`` of the documented package example produces when it is
      handed a plain directory, docs/api.rst:120-130).

    ``layout="auto"`` is deterministic: a bundle of this (species, tissue) that already exists under ``root``
    decides - its layout is kept, for reading and for re-fitting in place; BOTH layouts present is an error (pass the layout).
    With nothing there yet, writing takes ``reference`` iff ``root`` is named ``pretrained`` (the reference tree's own name),
    else ``flat``; reading falls back to ``flat`` (and the caller reports the missing file)."""

    def __init__(self, root, species: str, tissue: str, layout: str = "auto", for_write: bool = False):
        root = Path(root)
        if layout not in ("auto", "flat", "reference"):
            raise ValueError(f"bundle layout {layout!r}: auto, flat or reference")
        ref_base = root if (root / "models").is_dir() and not (root / species).is_dir() else root / species
        pt_name = f"{species}-{tissue}.pt"
        if layout == "auto":
            has_ref, has_flat = (ref_base / "models" / pt_name).exists(), (root / pt_name).exists()
            if has_ref and has_flat:
                raise ValueError(f"{root} holds {pt_name} in BOTH bundle layouts ({ref_base / 'models' / pt_name} and "
                                 f"{root / pt_name}): pass bundle_layout='reference' or 'flat'")
            if has_ref or has_flat:
                layout = "reference" if has_ref else "flat"
            else:
                layout = "reference" if (for_write and root.name == "pretrained") else "flat"
        self.layout, self.root = layout, root
        if layout == "reference":
            self.model = ref_base / "models" / pt_name
            self.support = ref_base / "graphs" / f"{species}_{tissue}_data.npz"
            self.genes = ref_base / "statistics" / f"{tissue}_genes.txt"
            self.cell_types = ref_base / "statistics" / f"{tissue}_cell_type.txt"
            # the label map sits next to ``pretrained/`` in the reference tree (predict.py:125 reads ./map/ from the cwd)
            self.map_candidates = [root / "celltype2subtype.xlsx", ref_base / "celltype2subtype.xlsx",
                                   root.parent / "map" / "celltype2subtype.xlsx", ref_base.parent.parent / "map" / "celltype2subtype.xlsx",
                                   Path("map") / "celltype2subtype.xlsx"]
        else:
            self.model = root / pt_name
            self.support = root / f"{species}_{tissue}_data.npz"
            self.genes = root / f"{tissue}_genes.txt"
            self.cell_types = root / f"{tissue}_cell_type.txt"
            self.map_candidates = [root / "celltype2subtype.xlsx", Path("map") / "celltype2subtype.xlsx"]

    def mkdirs(self) -> None:
        for f in (self.model, self.support, self.genes, self.cell_types):
            f.parent.mkdir(parents=True, exist_ok=True)

    def label_map(self) -> Optional[Path]:
        return next((f for f in self.map_candidates if f.exists()), None)


def _classify(logits: torch.Tensor, unsure_rate: float) -> Tuple[np.ndarray, np.ndarray]:
    """softmax -> 'unsure' iff max_prob < unsure_rate/num_classes, else argmax (predict.py:78-88)."""
    prob = F.softmax(logits.float(), dim=1)
    mx, arg = prob.max(dim=1)
    unsure = mx < unsure_rate / logits.shape[1]
    return torch.where(unsure, torch.full_like(arg, -1), arg).cpu().numpy(), prob.cpu().numpy()


class DeepSortClassifier:
    def __init__(self, species, tissue, dense_dim=400, hidden_dim=200, batch_size=256, dropout=0.1, gpu_id=-1,
                 file_type='csv', learning_rate=0.001, weight_decay=5e-4, n_epochs=300, n_layers=1, threshold=0,
                 num_neighbors=None, exclude_rate=0.005, random_seed=None, validation_fraction=0.1):
        self.num_neighbors = int(num_neighbors or 0)          # 0 / None = every in-edge (train.py:37-38)
        self.species, self.tissue = species, tissue
        self.dense_dim, self.hidden_dim, self.batch_size, self.dropout = dense_dim, hidden_dim, batch_size, dropout
        self.gpu_id, self.file_type = gpu_id, file_type
        self.learning_rate, self.weight_decay, self.n_epochs, self.n_layers = learning_rate, weight_decay, n_epochs, n_layers
        self.threshold, self.exclude_rate = threshold, exclude_rate
        self.random_seed, self.validation_fraction = random_seed, validation_fraction
        self.graph_steps = True                               # replay full-size mini-batch steps as one hipGraph each
        self.bundle_layout = "auto"                           # "auto" | "flat" | "reference" (see BundlePaths)
        self.model: Optional[GNN] = None
        self.history: List[dict] = []

    # ---------------------------------------------------------------------------------------------
    def fit(self, files: Sequence[Tuple[str, str]], save_path=None):
        """``files`` = list of (data_file, celltype_file) (docs/api.rst:89-95)."""
        # everything below - allocations, Adam state, hipGraph capture streams (``torch.cuda.graph`` opens its capture stream
        # on the CURRENT device) - must see ``gpu_id``'s device as current; the caller's device is restored on exit
        with torch.cuda.device(_device(self.gpu_id)):
            return self._fit(files, save_path)

    def _fit(self, files: Sequence[Tuple[str, str]], save_path=None):
        dev = _device(self.gpu_id)
        if self.random_seed is not None:
            np.random.seed(self.random_seed); torch.manual_seed(self.random_seed)
        tables, types = [], []
        for data_file, type_file in files:
            df = _read_expression(data_file, self.file_type)
            ct = pd.read_csv(type_file, index_col=0)
            ct.columns = ['cell', 'type']                                   # preprocess_internal.py:125-127
            ct['type'] = ct['type'].map(str.strip)
            assert ct['cell'].tolist() == df.index.tolist(), 'cell order of data and celltype files differs'
            tables.append(df); types.append(ct['type'])
        id2gene = sorted(set().union(*[set(map(str, t.columns)) for t in tables]))     # :26-41 union of genes
        all_types = pd.concat(types)
        counts = all_types.value_counts()
        id2label = sorted(t for t, n in counts.items() if n / len(all_types) > self.exclude_rate)   # :91-96
        gene2id = {g: i for i, g in enumerate(id2gene)}
        label2id = {l: i for i, l in enumerate(id2label)}
        mats, labels = [], []
        for df, ty in zip(tables, types):
            keep = ty.isin(label2id).to_numpy()
            arr = df.to_numpy(dtype=np.float32)[keep]
            cols = np.array([gene2id[str(c)] for c in df.columns])
            r, c = np.nonzero(arr > self.threshold)                          # :158
            mats.append(sp.csr_matrix((arr[r, c], (r, cols[c])), shape=(arr.shape[0], len(id2gene))))
            labels += [label2id[t] for t in ty[keep]]
        expr = sp.vstack(mats).tocsr(); expr.sort_indices()
        C, G = expr.shape
        feats = _features(expr, C, self.dense_dim, self.random_seed, dev)
        graph = CellGeneGraph.from_expression(expr, device=dev)
        y = torch.tensor(labels, dtype=torch.long, device=dev)
        perm = np.random.permutation(C)
        n_val = int(C * self.validation_fraction)
        val_ids = torch.from_numpy(perm[:n_val] + G).to(dev); train_ids = torch.from_numpy(perm[n_val:] + G).to(dev)
        model = GNN(self.dense_dim, self.hidden_dim, len(id2label), self.n_layers, G, activation=F.relu,
                    dropout=self.dropout).to(dev)
        best, self.history = -1.0, []
        sample_gen = None
        if self.num_neighbors:                                               # train.py:39-40,71-78
            from .sampler import DeviceSampler
            seed = self.random_seed if self.random_seed is not None else torch.initial_seed() % 2 ** 31
            kk = min(self.num_neighbors, max(graph.cg.max_row_nnz, graph.gc.max_row_nnz) + 1)
            # K5's static NodeFlow draws for ALL cells and genes below the last block (O((C+G) k) per layer, capturable);
            # the closure sampler only touches the nodes a batch reaches (~ B k^(L-1)) but has data-dependent shapes.
            # Static pays off when the closure is a sizeable part of the graph anyway, and always for one layer.
            closure = self.batch_size * kk ** max(0, self.n_layers - 1)
            if kk <= 256 and (self.n_layers == 1 or 4 * closure >= C + G):
                sample_gen = DeviceSampler(seed, dev)                        # K5: static shapes, sync-free, capturable
            else:                                                            # very wide draws / shallow closures: torch-op sampler
                sample_gen = torch.Generator(device=dev)
                sample_gen.manual_seed(seed)
        # static shapes (full neighbourhoods, or NodeFlows drawn by the device sampler): one hipGraph launch per batch
        static_shapes = not self.num_neighbors or not isinstance(sample_gen, torch.Generator)
        will_graph = self.graph_steps and static_shapes and len(train_ids) >= 8 * self.batch_size
        # train.py:34-35.  `capturable` (step counter and bias correction on device tensors) only when a step is actually
        # replayed as a hipGraph; otherwise the reference's plain Adam arithmetic
        # (fused=True: the same Adam arithmetic - L2 weight decay folded into the gradient - in one multi-tensor launch instead
        # of ~10 per step)
        opt = torch.optim.Adam(model.parameters(), lr=self.learning_rate, weight_decay=self.weight_decay, capturable=will_graph,
                               fused=True)
        bundle = BundlePaths(save_path, self.species, self.tissue, self.bundle_layout, for_write=True) if save_path is not None else None
        if bundle is not None:
            bundle.mkdirs()
            bundle.genes.write_bytes("".join(g + "\r\n" for g in id2gene).encode())
            bundle.cell_types.write_bytes("".join(l + "\r\n" for l in id2label).encode())
            sp.save_npz(bundle.support, expr)

        def accuracy(ids):
            if len(ids) == 0:
                return 1.0, 0
            model.eval()
            with torch.no_grad():
                pred, _ = _classify(model(graph, feats, seeds=ids), 2.0)
            truth = y[ids - G].cpu().numpy()
            return float((pred == truth).mean()), int((pred < 0).sum())

        def train_step(batch):                                               # train.py:71-87, no host synchronisation
            logits = model(graph, feats, seeds=batch, num_neighbors=self.num_neighbors, generator=sample_gen)
            loss = cross_entropy_sum(logits, y[batch - G])                   # CrossEntropyLoss(reduction='sum'), train.py:36
            opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
            return loss.detach()

        step = train_step
        if will_graph:
            from .graphed import GraphedTrainStep
            step = GraphedTrainStep(train_step, self.batch_size, dev)
        self._step = step
        for epoch in range(self.n_epochs):
            model.train()
            total = torch.zeros((), device=dev)
            order = train_ids[torch.randperm(len(train_ids), device=dev)]
            for batch in torch.split(order, self.batch_size):
                total += step(batch)                                         # accumulated on the device
            total = float(total)                                             # ONE read-back per epoch
            tr_acc, _ = accuracy(train_ids)
            va_acc, va_unsure = accuracy(val_ids)
            self.history.append(dict(epoch=epoch, loss=total / max(1, len(train_ids)), train_acc=tr_acc, val_acc=va_acc))
            if va_acc >= best:                                                # train.py:52-58
                best = va_acc
                if bundle is not None:
                    torch.save({'model': model.state_dict(), 'optimizer': opt.state_dict()}, bundle.model)
            if tr_acc == 1:                                                   # train.py:62-63
                break
        self.model, self._graph, self._feats, self._id2label, self._id2gene = model, graph, feats, id2label, id2gene
        return self

    # ---------------------------------------------------------------------------------------------
    def predict(self, input_file, model_path, save_path=None, unsure_rate=2., file_type='csv') -> pd.DataFrame:
        return _predict(self.species, self.tissue, input_file, Path(model_path), save_path, unsure_rate, file_type,
                        self.dense_dim, self.hidden_dim, self.gpu_id, self.threshold, self.random_seed)


class DeepSortPredictor:
    def __init__(self, species, tissue, file_type='csv', unsure_rate=2.):
        self.species, self.tissue, self.file_type, self.unsure_rate = species, tissue, file_type, unsure_rate

    def predict(self, input_file, save_path=None, model_path='pretrained') -> pd.DataFrame:
        """docs/api.rst:25-26 (the doc example also passes ``model_path=``, :42).  Uses the reference's fixed
        predict-time sizes dense_dim=400, hidden_dim=200 (predict.py:170-172)."""
        return _predict(self.species, self.tissue, input_file, Path(model_path), save_path, self.unsure_rate,
                        self.file_type, 400, 200, -1, 0, 10086)


def _predict(species, tissue, input_file, model_path: Path, save_path, unsure_rate, file_type, dense_dim, hidden_dim,
             gpu_id, threshold, seed) -> pd.DataFrame:
    with torch.cuda.device(_device(gpu_id)):                 # current device = gpu_id's for the call, restored on exit
        return _predict_on(species, tissue, input_file, model_path, save_path, unsure_rate, file_type, dense_dim, hidden_dim,
                           gpu_id, threshold, seed)


def _predict_logits(species, tissue, input_file, model_path: Path, file_type, gpu_id, threshold, seed):
    """The graph-side half of ``predict``: bundle -> predict graph (support cells + test cells) -> logits of the test cells.
    Returns (logits [n_test, n_classes] on the device, test cell names, id2label, BundlePaths)."""
    dev = _device(gpu_id)
    bundle = BundlePaths(model_path, species, tissue)
    missing = [str(f) for f in (bundle.model, bundle.support, bundle.genes, bundle.cell_types) if not f.exists()]
    if missing:
        raise FileNotFoundError(f"bundle for {species}/{tissue} under {model_path} ({bundle.layout} layout) lacks: " + ", ".join(missing))
    id2gene = [l.strip() for l in bundle.genes.read_text().splitlines() if l.strip()]
    id2label = [l.strip() for l in bundle.cell_types.read_text().splitlines() if l.strip()]
    support = sp.load_npz(bundle.support).tocsr()                                        # preprocess.py:114-117
    state = torch.load(bundle.model, map_location=dev)['model']                          # predict.py:56-59
    n_layers = sum(1 for k in state if k.endswith("fc_neigh.weight"))
    hidden_dim, dense_dim = state["layers.0.fc_neigh.weight"].shape
    G = len(id2gene)
    gene2id = {g: i for i, g in enumerate(id2gene)}
    df = _read_expression(input_file, file_type)
    cols = [c for c in df.columns if str(c) in gene2id]                                  # preprocess.py:160-161
    arr = df[cols].to_numpy(dtype=np.float32)
    cid = np.array([gene2id[str(c)] for c in cols])
    r, c = np.nonzero(arr > threshold)
    test = sp.csr_matrix((arr[r, c], (r, cid[c])), shape=(arr.shape[0], G))
    expr = sp.vstack([support, test]).tocsr(); expr.sort_indices()
    n_sup = support.shape[0]
    mask = np.zeros(expr.shape[0], bool); mask[:n_sup] = True           # test cells: gene->cell edges only (preprocess.py:184-187)
    feats = _features(expr, n_sup, dense_dim, seed, dev)
    graph = CellGeneGraph.from_expression(expr, support_mask=mask, device=dev)
    model = GNN(dense_dim, hidden_dim, len(id2label), n_layers, G, activation=F.relu, dropout=0.1).to(dev)
    model.load_state_dict(state)
    model.eval()
    seeds = range(G + n_sup, G + expr.shape[0])           # the test cells: one contiguous block of node ids (predict.py:64-76)
    with torch.no_grad():
        logits = model(graph, feats, seeds=seeds)
    return logits, df.index, id2label, bundle


def _predict_on(species, tissue, input_file, model_path: Path, save_path, unsure_rate, file_type, dense_dim, hidden_dim,
                gpu_id, threshold, seed) -> pd.DataFrame:
    logits, index, id2label, bundle = _predict_logits(species, tissue, input_file, model_path, file_type, gpu_id, threshold, seed)
    pred, _ = _classify(logits, unsure_rate)
    names = [id2label[p] if p >= 0 else "unsure" for p in pred]
    out = pd.DataFrame({"index": index, "cell_type": names})
    map_file = bundle.label_map()
    if map_file is not None:                                             # predict.py:124-146: new type / subtype names
        old2new, old2sub = load_label_map(map_file, species)
        out = pd.DataFrame({"index": index, "cell_type": [old2new.get(p, p) for p in names],
                            "cell_subtype": [old2sub.get(p, p) for p in names]})
    if save_path is not None:
        Path(save_path).mkdir(parents=True, exist_ok=True)
        out.to_csv(Path(save_path) / f"{species}_{tissue}_{Path(input_file).stem}.csv", index=False)
    return out
