"""DeepSortClassifier / DeepSortPredictor surface (reference docs/api.rst:6-130)."""
import inspect

import numpy as np
import pandas as pd
import pytest
import torch

import scdeepsort_amd as sda


def test_documented_signatures():
    """Keyword names and defaults exactly as documented (docs/api.rst:12-15, 50-66, 25-26, 89-102)."""
    sig = inspect.signature(sda.DeepSortClassifier.__init__)
    want = dict(dense_dim=400, hidden_dim=200, batch_size=256, dropout=0.1, gpu_id=-1, file_type='csv', learning_rate=0.001,
                weight_decay=5e-4, n_epochs=300, n_layers=1, threshold=0, num_neighbors=None, exclude_rate=0.005,
                random_seed=None, validation_fraction=0.1)
    got = {k: v.default for k, v in sig.parameters.items() if k not in ("self", "species", "tissue")}
    assert got == want
    assert list(sig.parameters)[1:3] == ["species", "tissue"]
    p = inspect.signature(sda.DeepSortPredictor.__init__).parameters
    assert p["file_type"].default == 'csv' and p["unsure_rate"].default == 2.
    f = inspect.signature(sda.DeepSortClassifier.fit).parameters
    assert list(f)[1:] == ["files", "save_path"] and f["save_path"].default is None
    q = inspect.signature(sda.DeepSortClassifier.predict).parameters
    assert list(q)[1:] == ["input_file", "model_path", "save_path", "unsure_rate", "file_type"]
    assert q["unsure_rate"].default == 2. and q["file_type"].default == 'csv'
    r = inspect.signature(sda.DeepSortPredictor.predict).parameters
    assert list(r)[1:3] == ["input_file", "save_path"]


def test_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sda.WgnnError):
        sda.DeepSortClassifier("mouse", "Testis").fit([])


def _write_dataset(tmp, name, n_cells, rng, genes, programs, with_types=True):
    """(genes x cells) csv in the format of pre-process.R:72-74 + celltype csv (index, Cell, Cell_type)."""
    types = rng.integers(0, len(programs), n_cells)
    X = np.zeros((len(genes), n_cells), np.float32)
    for j, t in enumerate(types):
        on = rng.random(len(genes)) < (0.05 + 0.6 * programs[t])
        X[on, j] = np.clip(rng.normal(3.0, 0.9, on.sum()), 0.5, 7.0)
    cells = [f"{name}_C{j}" for j in range(n_cells)]
    data = tmp / f"{name}_data.csv"
    pd.DataFrame(X, index=genes, columns=cells).to_csv(data)
    if not with_types:
        return data, types
    ct = tmp / f"{name}_celltype.csv"
    pd.DataFrame({"Cell": cells, "Cell_type": [f"type{t} " for t in types]}).to_csv(ct)     # trailing blank: stripped like :127
    return data, ct, types


@pytest.mark.gpu
def test_fit_predict_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    genes = [f"G{i}" for i in range(120)]
    programs = [np.zeros(120) for _ in range(3)]
    for t in range(3):
        programs[t][t * 40:(t + 1) * 40] = 1.0
    d1, c1, _ = _write_dataset(tmp_path, "mouse_Demo1", 240, rng, genes, programs)
    d2, c2, _ = _write_dataset(tmp_path, "mouse_Demo2", 120, rng, genes, programs)
    clf = sda.DeepSortClassifier("mouse", "Demo", dense_dim=16, hidden_dim=12, batch_size=64, n_epochs=300, n_layers=2, learning_rate=0.005,
                                 random_seed=1, gpu_id=0, dropout=0.1)
    clf.fit([(d1, c1), (d2, c2)], save_path=tmp_path / "bundle")
    assert clf.history[-1]["train_acc"] > 0.95 and max(h["val_acc"] for h in clf.history) > 0.9
    # bundle format (train.py:117-123, preprocess_internal.py:59-67,180)
    state = torch.load(tmp_path / "bundle" / "mouse-Demo.pt", map_location="cpu")
    assert set(state) == {"model", "optimizer"}
    assert set(state["model"]) == {"layers.0.fc_neigh.weight", "layers.0.fc_neigh.bias", "layers.1.fc_neigh.weight",
                                   "layers.1.fc_neigh.bias", "alpha", "linear.weight", "linear.bias"}
    assert state["model"]["alpha"].shape == (122, 1)
    assert (tmp_path / "bundle" / "Demo_genes.txt").read_bytes().count(b"\r\n") == 120
    assert (tmp_path / "bundle" / "Demo_cell_type.txt").read_text().split() == ["type0", "type1", "type2"]
    # predict on unseen cells: gene->cell edges only for them (preprocess.py:184-187)
    dt, truth = _write_dataset(tmp_path, "mouse_Test9", 90, rng, genes, programs, with_types=False)
    out = clf.predict(dt, model_path=tmp_path / "bundle", save_path=tmp_path / "result")
    assert list(out.columns) == ["index", "cell_type"] and len(out) == 90
    acc = np.mean([o == f"type{t}" for o, t in zip(out["cell_type"], truth)])
    assert acc > 0.9
    out2 = sda.DeepSortPredictor("mouse", "Demo").predict(dt, model_path=tmp_path / "bundle")
    assert out2["cell_type"].tolist() == out["cell_type"].tolist()
    assert (tmp_path / "result").exists()
    # with the label map in the bundle the output carries the new type / subtype names (predict.py:124-146)
    _write_xlsx(tmp_path / "bundle" / "celltype2subtype.xlsx",
                {"mouse": [["Species", "Cell type", "Cell-type", "Cell-subtype"], ["Mouse", "type0", "Zero", "zero-a"],
                           ["Mouse", "type1", "One", None]]})
    out3 = clf.predict(dt, model_path=tmp_path / "bundle")
    assert list(out3.columns) == ["index", "cell_type", "cell_subtype"]
    # a save_path named `pretrained` is written as the reference's tree (train.py:20, preprocess_internal.py:79-80) and reloads
    clf2 = sda.DeepSortClassifier("mouse", "Demo", dense_dim=16, hidden_dim=12, batch_size=64, n_epochs=2, n_layers=2, random_seed=1, gpu_id=0)
    clf2.fit([(d1, c1)], save_path=tmp_path / "pretrained")
    for rel in ("models/mouse-Demo.pt", "graphs/mouse_Demo_data.npz", "statistics/Demo_genes.txt", "statistics/Demo_cell_type.txt"):
        assert (tmp_path / "pretrained" / "mouse" / rel).exists(), rel
    assert len(clf2.predict(dt, model_path=tmp_path / "pretrained")) == 90
    want_t = {"type0": "Zero", "type1": "One"}; want_s = {"type0": "zero-a", "type1": "N/A"}
    assert out3["cell_type"].tolist() == [want_t.get(p, p) for p in out["cell_type"]]
    assert out3["cell_subtype"].tolist() == [want_s.get(p, p) for p in out["cell_type"]]


@pytest.mark.gpu
@pytest.mark.parametrize("n_layers", [1, 2])
def test_fit_replays_minibatch_steps_as_hipgraphs(tmp_path, n_layers):
    """Default reference shape (n_layers = 1, train.py:145): full-size mini-batch steps are captured once and replayed
    (graphed.GraphedTrainStep) - same training outcome as the eager loop."""
    rng = np.random.default_rng(5)
    genes = [f"G{i}" for i in range(150)]
    programs = [np.zeros(150) for _ in range(3)]
    for t in range(3):
        programs[t][t * 50:(t + 1) * 50] = 1.0
    d1, c1, _ = _write_dataset(tmp_path, "mouse_Demo1", 700, rng, genes, programs)
    hist = {}
    for graphed in (True, False):
        clf = sda.DeepSortClassifier("mouse", "Demo", dense_dim=16, hidden_dim=12, batch_size=64, n_epochs=12,
                                     n_layers=n_layers, learning_rate=0.01, random_seed=4, gpu_id=0, dropout=0.0)
        clf.graph_steps = graphed
        clf.fit([(d1, c1)])
        hist[graphed] = clf.history
        if graphed:
            assert clf._step.replays == 9 * len(clf.history) - 3     # 9 full batches per epoch (the 10th is a tail), 3 eager warm-ups
    assert max(h["val_acc"] for h in hist[True]) > 0.9
    for a, b in zip(hist[True], hist[False]):                    # dropout 0, same seed -> same trajectory
        assert a["loss"] == pytest.approx(b["loss"], rel=1e-3)


@pytest.mark.gpu
def test_fit_with_neighbour_subsampling(tmp_path):
    """num_neighbors > 0 (docs/api.rst:62, train.py:37-40): training draws <= k in-edges per node per batch,
    evaluation still uses the full neighbourhood (train.py:92-108)."""
    rng = np.random.default_rng(2)
    genes = [f"G{i}" for i in range(90)]
    programs = [np.zeros(90) for _ in range(3)]
    for t in range(3):
        programs[t][t * 30:(t + 1) * 30] = 1.0
    d1, c1, _ = _write_dataset(tmp_path, "mouse_Demo1", 300, rng, genes, programs)
    clf = sda.DeepSortClassifier("mouse", "Demo", dense_dim=16, hidden_dim=12, batch_size=64, n_epochs=60, n_layers=2,
                                 learning_rate=0.01, random_seed=3, gpu_id=0, dropout=0.0, num_neighbors=8)
    clf.fit([(d1, c1)])
    assert clf.num_neighbors == 8
    assert max(h["val_acc"] for h in clf.history) > 0.8


def test_bundle_paths_resolve_both_layouts(tmp_path):
    """f4: the reference's bundle tree (predict.py:57; preprocess.py:67-68,77-78,114; train.py:20; preprocess_internal.py:79-80)
    and the flat directory; auto-detection for reading and for writing."""
    from scdeepsort_amd.api import BundlePaths
    root = tmp_path / "pretrained"
    b = BundlePaths(root, "mouse", "Testis", for_write=True)                 # a directory NAMED pretrained: the reference tree
    assert b.layout == "reference"
    assert b.model == root / "mouse" / "models" / "mouse-Testis.pt"
    assert b.support == root / "mouse" / "graphs" / "mouse_Testis_data.npz"
    assert b.genes == root / "mouse" / "statistics" / "Testis_genes.txt"
    assert b.cell_types == root / "mouse" / "statistics" / "Testis_cell_type.txt"
    b.mkdirs()
    assert (root / "mouse" / "models").is_dir() and (root / "mouse" / "graphs").is_dir() and (root / "mouse" / "statistics").is_dir()
    flat = BundlePaths(tmp_path / "model_save_path", "mouse", "Testis", for_write=True)
    assert flat.layout == "flat" and flat.model == tmp_path / "model_save_path" / "mouse-Testis.pt"
    assert flat.genes == tmp_path / "model_save_path" / "Testis_genes.txt"
    # reading: the layout whose checkpoint exists; the species directory itself is accepted as root
    assert BundlePaths(root, "mouse", "Testis").layout == "flat"             # nothing written yet: falls back to flat
    b.model.write_bytes(b"x")
    assert BundlePaths(root, "mouse", "Testis").layout == "reference"
    inner = BundlePaths(root / "mouse", "mouse", "Testis")
    assert inner.layout == "reference" and inner.model == b.model and inner.genes == b.genes
    assert BundlePaths(root, "mouse", "Testis", layout="flat").model == root / "mouse-Testis.pt"
    # the label map of the reference tree sits next to pretrained/ (predict.py:125: ./map/celltype2subtype.xlsx)
    (tmp_path / "map").mkdir()
    (tmp_path / "map" / "celltype2subtype.xlsx").write_bytes(b"")
    assert BundlePaths(root, "mouse", "Testis").label_map() == tmp_path / "map" / "celltype2subtype.xlsx"
    with pytest.raises(ValueError):
        BundlePaths(root, "mouse", "Testis", layout="tree")
    # the choice is deterministic.  An existing bundle keeps its layout on a re-fit (a flat bundle inside a directory
    # that has grown a models/ subtree stays flat; the reference tree stays the reference tree) ...
    fdir = tmp_path / "model_save_path"
    fdir.mkdir(); (fdir / "mouse-Testis.pt").write_bytes(b"x"); (fdir / "models").mkdir()
    assert BundlePaths(fdir, "mouse", "Testis", for_write=True).layout == "flat"
    assert BundlePaths(root, "mouse", "Testis", for_write=True).layout == "reference"
    # ... a species directory handed over before it holds anything is a plain directory, not half a reference tree ...
    assert BundlePaths(tmp_path / "pretrained2" / "human", "human", "Lung", for_write=True).layout == "flat"
    # ... and a root that holds the checkpoint in BOTH layouts is refused instead of one silently shadowing the other
    (root / "mouse-Testis.pt").write_bytes(b"y")
    with pytest.raises(ValueError, match="BOTH"):
        BundlePaths(root, "mouse", "Testis")
    assert BundlePaths(root, "mouse", "Testis", layout="reference").model == b.model


def _handwritten_bundle(tmp_path, layout):
    """A bundle written BY HAND (no fit): parameters from the reference's own GNN.__init__ (tests/golden/refcode_predict.npz
    `param.*`), the golden predict graph's support cells as graphs/*.npz, its test cells as the input csv."""
    import scipy.sparse as sp
    from pathlib import Path
    from scdeepsort_amd.api import BundlePaths
    z = np.load(Path(__file__).parent / "golden" / "refcode_predict.npz")
    expr, mask = z["expr"].astype(np.float32), z["support_mask"].astype(bool)
    genes = [f"Gene{i}" for i in range(expr.shape[1])]
    labels = [f"type{i}" for i in range(int(z["n_classes"]))]
    root = tmp_path / ("pretrained" if layout == "reference" else "flatdir")
    b = BundlePaths(root, "mouse", "Demo", layout=layout, for_write=True)
    b.mkdirs()
    b.genes.write_bytes("".join(g + "\r\n" for g in genes).encode())
    b.cell_types.write_bytes("".join(l + "\r\n" for l in labels).encode())
    sp.save_npz(b.support, sp.csr_matrix(expr[mask]))
    state = {k[len("param."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    torch.save({"model": state, "optimizer": {}}, b.model)
    test = expr[~mask]
    cells = [f"T{j}" for j in range(test.shape[0])]
    data = tmp_path / f"mouse_Demo_test_{layout}.csv"
    pd.DataFrame(test.T, index=genes, columns=cells).to_csv(data)
    return root, data, cells


@pytest.mark.gpu
def test_reference_layout_bundle_loads_as_shipped(tmp_path):
    """a bundle in the reference's `pretrained/{species}/{models,graphs,statistics}/` layout - written by
    hand, parameters from the reference's own constructor - loads and gives the same logits as the flat bundle."""
    from scdeepsort_amd.api import _predict_logits
    root_r, data_r, cells = _handwritten_bundle(tmp_path, "reference")
    root_f, data_f, _ = _handwritten_bundle(tmp_path, "flat")
    assert (root_r / "mouse" / "models" / "mouse-Demo.pt").exists() and (root_f / "mouse-Demo.pt").exists()
    lr, idx, labels, br = _predict_logits("mouse", "Demo", data_r, root_r, "csv", 0, 0, 10086)
    lf, _, _, bf = _predict_logits("mouse", "Demo", data_f, root_f, "csv", 0, 0, 10086)
    assert br.layout == "reference" and bf.layout == "flat"
    assert lr.shape == (len(cells), 4) and list(idx) == cells and labels == [f"type{i}" for i in range(4)]
    assert torch.equal(lr, lf) and torch.isfinite(lr).all()
    out_r = sda.DeepSortPredictor("mouse", "Demo", unsure_rate=0.).predict(data_r, model_path=root_r)
    out_f = sda.DeepSortPredictor("mouse", "Demo", unsure_rate=0.).predict(data_f, model_path=root_f)
    assert out_r["cell_type"].tolist() == out_f["cell_type"].tolist() == [labels[i] for i in lr.argmax(1).tolist()]
    # the species directory as model_path, and a fit() that is told to write the reference tree
    out_s = sda.DeepSortPredictor("mouse", "Demo", unsure_rate=0.).predict(data_r, model_path=root_r / "mouse")
    assert out_s["cell_type"].tolist() == out_r["cell_type"].tolist()
    with pytest.raises(FileNotFoundError):
        sda.DeepSortPredictor("mouse", "Nope").predict(data_r, model_path=root_r)


def _write_xlsx(path, sheets):
    """Minimal .xlsx (shared strings only) written with zipfile: {sheet name: rows of str | None}."""
    import zipfile
    from xml.sax.saxutils import escape
    strings, index = [], {}
    def sid(s):
        if s not in index:
            index[s] = len(strings); strings.append(s)
        return index[s]
    sheet_xml = {}
    for n, (name, rows) in enumerate(sheets.items(), 1):
        body = []
        for r, row in enumerate(rows, 1):
            cells = "".join(f'<c r="{chr(64 + c)}{r}" t="s"><v>{sid(v)}</v></c>' for c, v in enumerate(row, 1) if v is not None)
            body.append(f'<row r="{r}">{cells}</row>')
        sheet_xml[n] = ('<?xml version="1.0" encoding="UTF-8"?><worksheet xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main">'
                        f'<sheetData>{"".join(body)}</sheetData></worksheet>')
    ns = 'xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main" xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships"'
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("xl/workbook.xml", f'<?xml version="1.0"?><workbook {ns}><sheets>' + "".join(
            f'<sheet name="{name}" sheetId="{n}" r:id="rId{n}"/>' for n, name in enumerate(sheets, 1)) + "</sheets></workbook>")
        z.writestr("xl/_rels/workbook.xml.rels", '<?xml version="1.0"?><Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">' + "".join(
            f'<Relationship Id="rId{n}" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/worksheet" Target="worksheets/sheet{n}.xml"/>'
            for n in sheet_xml) + "</Relationships>")
        z.writestr("xl/sharedStrings.xml", '<?xml version="1.0"?><sst xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main">' +
                   "".join(f"<si><t>{escape(s)}</t></si>" for s in strings) + "</sst>")
        for n, xml in sheet_xml.items():
            z.writestr(f"xl/worksheets/sheet{n}.xml", xml)


def test_label_map_reader(tmp_path):
    """celltype2subtype.xlsx (predict.py:124-133): per-species sheet, old type -> (new type, new subtype), blanks and
    pandas' NA strings -> 'N/A'."""
    from scdeepsort_amd.api import load_label_map, read_xlsx_sheet
    f = tmp_path / "celltype2subtype.xlsx"
    _write_xlsx(f, {"human": [["Species", "Cell type", "Cell-type", "Cell-subtype"], ["Human", "T cell", "T cell", "NA"]],
                    "mouse": [["Species", "Cell type", "Cell-type", "Cell-subtype"],
                              ["Mouse", "type0", "Alpha & beta", "sub<0>"], ["Mouse", "type1", "Beta", None], ["Mouse", "type2", "Gamma", "NA"]]})
    assert read_xlsx_sheet(f, "human")[1] == ["Human", "T cell", "T cell", "NA"]
    new, sub = load_label_map(f, "mouse")
    assert new == {"type0": "Alpha & beta", "type1": "Beta", "type2": "Gamma"}
    assert sub == {"type0": "sub<0>", "type1": "N/A", "type2": "N/A"}
    with pytest.raises(KeyError):
        read_xlsx_sheet(f, "rat")
