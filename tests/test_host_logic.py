"""CPU tests of the host-side mirror of the reference interface (no kernels launched)."""
import io
import pytest

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


def _rank_inclusion(col, cells, genes, rank):
    """Observed inclusion frequency around a popularity rank (mean over a +-10 % rank window of the sorted frequencies)."""
    pop = np.sort(np.bincount(col.numpy(), minlength=genes) / cells)[::-1]
    lo, hi = max(0, int(rank / 1.1) - 1), int(rank * 1.1) + 1
    return float(pop[lo:hi].mean())


def test_synthetic_generator_statistics():
    """SURVEY 8d's generator: numpy default_rng(seed), k_c log-normal (log-std 0.52, mean density*G, >= 16), popularity =
    Testis199's curve (0.98 / 0.67 / 0.32 / 0.09 / 0.035 at rank 1 / 10 / 100 / 1000 / 3000 of 9339), values clip(N(3, .9), .5, 7)."""
    C, G = 3000, S.TESTIS_GENES
    rp, col, val = S.synth_expression(C, G, seed=1)
    assert rp.dtype == torch.int64 and col.dtype == torch.int32 and val.dtype == torch.float32
    k = (rp[1:] - rp[:-1]).float()
    assert abs(k.mean().item() / (0.04 * G) - 1) < 0.05
    assert abs(k.log().std().item() - 0.52) < 0.04
    assert k.min() >= 16
    # sorted, unique columns per row
    for r in (0, 7, C - 1):
        c = col[rp[r]:rp[r + 1]]
        assert torch.all(c[1:] > c[:-1])
    for rank, want in ((1, 0.98), (10, 0.67), (100, 0.32), (1000, 0.09), (3000, 0.035)):
        got = _rank_inclusion(col, C, G, rank)
        assert abs(got / want - 1) < 0.15, (rank, got, want)
    pop = torch.bincount(col.long(), minlength=G).float() / C
    assert 0.005 < pop.median() < 0.025                                 # "the median gene in 1-2 % of cells"
    assert val.min() >= 0.5 and val.max() <= 7.0 and abs(val.mean().item() - 3.0) < 0.05 and abs(val.std().item() - 0.9) < 0.05
    rp2, col2, val2 = S.synth_expression(C, G, seed=1)
    assert torch.equal(col, col2) and torch.equal(val, val2)            # deterministic
    rp3, col3, val3 = S.synth_expression(C, G, seed=1, chunk_cells=257)
    assert torch.equal(rp, rp3)                                         # the k_c law does not depend on the chunking
    # hub genes are scattered over the id range (shuffled), not the first ids
    top = torch.topk(pop, 50).indices
    assert top.float().mean() > 0.25 * G


def test_synthetic_curve_at_other_gene_counts():
    """The curve is applied at RELATIVE rank (it must sum to density*G): same top / median / density at cfg3's 20 000 genes, the
    five anchors at rank*G/9339; the weights solve reproduces the target curve for the k_c law (no sampling involved)."""
    for G in (2_000, 20_000):
        pi = S.inclusion_curve(G, 0.04)
        assert abs(pi.sum() / (0.04 * G) - 1) < 1e-6 and pi.max() <= 0.98 + 1e-12 and np.all(np.diff(pi) <= 1e-15)
        for rank, want in ((1, 0.98), (10, 0.67), (100, 0.32), (1000, 0.09), (3000, 0.035)):
            r = round(rank * G / S.TESTIS_GENES)
            if r >= 1:                                                  # coarser than Testis' ranks: no gene sits at that quantile
                assert abs(pi[r - 1] / want - 1) < 0.15, (G, rank, pi[r - 1])
        assert 0.01 <= pi[G // 2] <= 0.02
        w = S.sampling_weights(G, 0.04)
        ks = S._nnz_quantiles(G, 0.04)
        t = np.ones_like(ks)
        for _ in range(200):
            e = np.exp(-np.outer(t, w)); t = t - ((1 - e).sum(1) - ks) / (e * w).sum(1)
        got = (1 - np.exp(-np.outer(t, w))).mean(0)
        assert np.abs(got / pi - 1).max() < 1e-3
    rp, col, _ = S.synth_expression(1500, 2_000, seed=5)
    assert abs(col.shape[0] / (1500 * 2_000) / 0.04 - 1) < 0.06
    assert abs(_rank_inclusion(col, 1500, 2_000, 21) / 0.32 - 1) < 0.15    # rank 100 of 9339 == rank 21 of 2000


def test_synthetic_dense_head_generator_kept_for_ab():
    rp, col, val = S.synth_expression(2000, 1500, seed=1, popularity="dense_head")
    pop = torch.bincount(col.long(), minlength=1500).float() / 2000
    assert pop.max() > 0.9 and pop.median() < 0.05
    assert abs((rp[1:] - rp[:-1]).float().mean().item() / (0.04 * 1500) - 1) < 0.1
    with pytest.raises(ValueError):
        S.synth_expression(10, 10, popularity="nope")


def test_shard_ranges_cover_cells():
    from scdeepsort_amd.dist import shard_range
    for C, W in ((100_000, 8), (764_741, 8), (10, 4), (7, 8)):
        spans = [shard_range(C, r, W) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == C
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and b >= a
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _cpu_csr(n_rows=60, n_cols=45, seed=0):
    """A destination-major AggCsr on CPU tensors (the sampler is pure index work + the host plan builder)."""
    from scdeepsort_amd.graph import AggCsr, build_plan
    rng = np.random.default_rng(seed)
    dense = rng.random((n_rows, n_cols)) < 0.25
    dense[2] = False                                    # an isolated row: only its self-loop can be drawn
    dense[7] = True                                     # a hub row
    rowptr = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    col = np.nonzero(dense)[1].astype(np.int32)
    val = rng.random(len(col)).astype(np.float32)
    inv = (1.0 / (dense.sum(1) + 1)).astype(np.float32)
    return AggCsr(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), torch.from_numpy(inv),
                  n_rows, n_cols, build_plan(rowptr), rowptr), dense


def test_neighbour_sampler_draw_counts_and_membership():
    """train.py:37-40: at most num_neighbors in-edges per node, uniform without replacement, self-loop included."""
    from scdeepsort_amd.sampler import sample_block
    csr, dense = _cpu_csr()
    rows = torch.tensor([7, 2, 0, 33, 59, 12])
    gen = torch.Generator().manual_seed(3)
    for k in (1, 2, 5, 100):
        blk = sample_block(csr, rows, k, gen)
        rp = blk.csr.rowptr.numpy()
        for j, r in enumerate(rows.tolist()):
            got = blk.csr.col.numpy()[rp[j]:rp[j + 1]]
            full = np.nonzero(dense[r])[0]
            assert len(set(got)) == len(got) and set(got) <= set(full)
            n_drawn = len(got) + int(blk.self_drawn[j])
            assert n_drawn == min(k, len(full) + 1)
            assert abs(float(blk.csr.inv_deg[j]) - 1.0 / n_drawn) < 1e-7
            # weights travel with their edge
            pos = {c: csr.val.numpy()[csr.rowptr_host[r] + i] for i, c in enumerate(full)}
            np.testing.assert_array_equal(blk.csr.val.numpy()[rp[j]:rp[j + 1]], [pos[c] for c in got])
    blk = sample_block(csr, rows, 100, gen)             # k >= every degree: the whole neighbourhood
    assert torch.all(blk.self_drawn == 1)


def test_neighbour_sampler_is_uniform_and_seeded():
    from scdeepsort_amd.sampler import sample_block
    csr, dense = _cpu_csr(seed=1)
    rows = torch.tensor([7])                            # hub row: 45 edges + self-loop = 46 candidates
    gen = torch.Generator().manual_seed(11)
    hits = np.zeros(46)
    n = 4000
    for _ in range(n):
        blk = sample_block(csr, rows, 5, gen)
        hits[blk.csr.col.numpy()] += 1
        hits[45] += float(blk.self_drawn[0])
    p = 5 / 46
    assert np.all(np.abs(hits / n - p) < 5 * np.sqrt(p * (1 - p) / n))
    a = sample_block(csr, rows, 5, torch.Generator().manual_seed(5))
    b = sample_block(csr, rows, 5, torch.Generator().manual_seed(5))
    assert torch.equal(a.csr.col, b.csr.col) and torch.equal(a.self_drawn, b.self_drawn)


def test_generated_flat_pipeline_is_in_sync_and_well_formed(tmp_path):
    """csrc/wgnn_flat_asm.inc is generated (gen_flat_asm.py): the committed file must be the generator's output, every
    step must keep its LDS wait count consistent with the reads issued after the pair it consumes, and every branch
    target must exist."""
    import re
    import runpy
    from pathlib import Path
    csrc = Path(sda.__file__).resolve().parent / "csrc"
    gen = runpy.run_path(str(csrc / "gen_flat_asm.py"))
    out = tmp_path / "flat.inc"
    gen["main"](str(out))
    assert out.read_text() == (csrc / "wgnn_flat_asm.inc").read_text()
    lines = [l.strip().strip('"\\ ').replace("\\n\\t", "") for l in out.read_text().splitlines() if l.strip().startswith('"')]
    labels = {l[:-1] for l in lines if l.endswith(":")}
    for l in lines:
        m = re.match(r"s_c?branch(?:_scc1)? (\S+)", l)
        if m:
            assert m.group(1) in labels, l
    assert sum(1 for l in labels if "_step" in l) == 32 and sum(1 for l in labels if "_pre" in l) == 32
    assert sum(1 for l in labels if "_sstep" in l) == 31
    # per step: reads of pair p+1 (3 LDS ops; 2 in the shared-pair stream) are outstanding when pair p is consumed
    text = "\n".join(lines)
    for stream, n_ops in (("step", 3), ("sstep", 2)):
        for p in range(0 if stream == "step" else 1, 32):
            body = re.split(r"\.Lw4_s?step\d+_%=:|\.Lw4_end_%=:", text.split(f".Lw4_{stream}{p}_%=:")[1])[0]
            n_reads = len(re.findall(r"ds_read_b(128|64)", body))
            wait = int(re.search(r"s_waitcnt lgkmcnt\((\d+)\)", body).group(1))
            assert n_reads == (n_ops if p < 31 else 0) and wait == n_reads, (stream, p, n_reads, wait)
            assert len(re.findall(r"v_pk_fma_f32", body)) == 4
            assert len(re.findall(r"v_readlane_b32", body)) == (0 if p >= 30 else n_ops - 1)
            if stream == "step" and p < 31:          # hand-over to the shared-pair stream after every unshared step
                assert f"s_cmp_eq_u32 %[sw], {p + 1}" in body and f"s_cbranch_scc1 .Lw4_sstep{p + 1}_%=" in body


def test_plan_heuristics():
    from scdeepsort_amd.graph import auto_chunk, auto_tile_geometry
    assert auto_chunk(0) == 256 and auto_chunk(2_000_000) == 256 and auto_chunk(80_000_000) == 2048
    assert all(auto_chunk(n) in (256, 512, 1024, 2048) for n in (1, 3_000_000, 6_000_000, 10_000_000, 10 ** 9))
    # cells side of cfg3: two full rounds over 256 CUs, no column split; few-row operands (the gene side, small graphs):
    # ONE round of <= 256 workgroups, column-split, row tiles slightly shrunk to fill the round
    assert auto_tile_geometry(100_000, 20_000) == (512, 1) and auto_tile_geometry(50_000, 20_000, nnz=40_000_000) == (256, 1)
    rt, cs = auto_tile_geometry(20_600, 100_000, nnz=80_000_000)
    assert (rt, cs) == (85, 3) and rt * 250 >= 20_600 and rt * cs <= 256
    assert auto_tile_geometry(10_000, 10_000, nnz=4_000_000) == (42, 6)
    assert auto_tile_geometry(300, 1_000, nnz=30_000) == (3, 1)            # a small operand is never shredded to fill CUs
    assert auto_tile_geometry(300, 100) == (2, 1)
    # dedicated loader waves: a tile's rows must fit the computing waves' accumulator rows (15 x 16 with one loader wave, 14 x 16
    # with two) - cfg3 keeps its 512 tiles, cfg5's 764,741 rows take 13 / 14 rounds of 230- / 213-row tiles instead of 12 of 249
    assert auto_tile_geometry(100_000, 20_000, rows_cap=224) == (512, 1) and auto_tile_geometry(100_000, 20_000, rows_cap=240) == (512, 1)
    assert auto_tile_geometry(764_741, 20_000) == (3072, 1) and auto_tile_geometry(764_741, 20_000, rows_cap=224) == (3584, 1)
    assert auto_tile_geometry(764_741, 20_000, rows_cap=240) == (3328, 1)


def test_cu_budget_of_the_pass_that_overlaps_a_collective():
    """A rank's cells<-genes pass runs next to the in-flight [G, H] all-reduce: its ONE-round geometry leaves the communicator
    its CUs (dist.COMM_CUS; every tile workgroup needs a whole CU).  Shards of the cfg3 job at N = 2 / 4 / 8 fit 224 CUs; an
    operand that needs several rounds anyway keeps the whole chip."""
    from scdeepsort_amd import dist as D, graph as GR
    from scdeepsort_amd.graph import auto_tile_geometry
    assert D.COMM_CUS == 32
    for rows, nnz in ((12_500, 9_900_000), (25_000, 19_800_000), (50_000, 39_800_000)):
        full = auto_tile_geometry(rows, 20_000, 256, nnz, rows_cap=240)
        lean = auto_tile_geometry(rows, 20_000, 224, nnz, rows_cap=240)
        assert 224 < full[0] * full[1] <= 256 and 200 <= lean[0] * lean[1] <= 224
        assert lean[0] * 240 >= rows                                   # the rows still fit the computing waves' accumulators
    rng = np.random.default_rng(5)
    A = (rng.random((900, 700)) < 0.05) * rng.uniform(0.5, 2.0, (900, 700))
    csr = _cpu_agg_csr(A)
    one = csr.tile_plan(32)
    csr.cu_budget = 3                       # 900 rows need >= 4 tiles: more than one round of the budget -> the full-chip plan
    assert csr.tile_plan(32).n_tiles == one.n_tiles
    csr.cu_budget = 224                     # fits: a plan of its own (cached under its own key), within the budget
    assert csr.tile_plan(32).n_tiles <= 224 and len(csr._tile_plan) == 3
    # the channel cap handed to RCCL is OPT-IN (process-wide, calibrated on a stand-in); when asked for it matches the CUs
    # left free; a value the user exported wins
    import os
    saved = os.environ.pop("NCCL_MAX_NCHANNELS", None)
    try:
        D.reserve_comm_cus()
        assert "NCCL_MAX_NCHANNELS" not in os.environ
        D.reserve_comm_cus(cap_channels=True)
        assert os.environ["NCCL_MAX_NCHANNELS"] == "32"
        os.environ["NCCL_MAX_NCHANNELS"] = "8"
        D.reserve_comm_cus(cap_channels=True)
        assert os.environ["NCCL_MAX_NCHANNELS"] == "8"
    finally:
        os.environ.pop("NCCL_MAX_NCHANNELS", None)
        if saved is not None:
            os.environ["NCCL_MAX_NCHANNELS"] = saved


def _cpu_agg_csr(A):
    """AggCsr over CPU tensors (the tile-plan builder is pure index arithmetic and runs on any device)."""
    import scipy.sparse as sp
    from scdeepsort_amd import graph as GR
    A = sp.csr_matrix(A); A.sort_indices()
    rp = torch.from_numpy(A.indptr.astype(np.int32))
    empty = torch.zeros((0, 4), dtype=torch.int32)
    return GR.AggCsr(rp, torch.from_numpy(A.indices.astype(np.int32)), torch.from_numpy(A.data.astype(np.float32)),
                     torch.ones(A.shape[0]), A.shape[0], A.shape[1], GR.Plan(empty, empty, 0, 256), A.indptr.astype(np.int32))


def test_tile_plan_with_loader_waves_covers_every_edge_once():
    """Dedicated loader waves are a property of the tile plan: the first L waves of every tile own no rows, every other
    wave's slots fill from slot 0 (the kernel reads "slot 0 empty" as "loader wave"), and the re-ordered entries still hold
    every non-zero of the CSR exactly once with its weight - for explicit geometries with and without column splits, for
    the heuristic geometry of a many-row and a few-row operand, and for tiles too full to spare a wave (fallback to 0)."""
    import scipy.sparse as sp
    from scdeepsort_amd import graph as GR
    rng = np.random.default_rng(3)
    X = sp.random(3000, 700, density=0.06, format="csr", random_state=5, dtype=np.float32)
    X.data = np.abs(X.data) + 0.5
    Xd = X.toarray(); Xd[:, 3] = 2.0                                      # a hub gene: one row of the gene side is 3000 long
    X = sp.csr_matrix(Xd)
    for A, geoms in ((X, [(16, 1, 2), (None, None, 1), (None, None, 2), (12, 1, 3)]),
                     (sp.csr_matrix(X.T), [(4, 3, 1), (None, None, 1), (3, 2, 2)])):
        csr = _cpu_agg_csr(A)
        want = sorted(zip(np.repeat(np.arange(A.shape[0]), np.diff(sp.csr_matrix(A).indptr)).tolist(),
                          csr.col.tolist(), csr.val.tolist()))
        for rt, cs, L in geoms:
            saved = dict(GR.LOADER_MIN_ENTRIES)
            GR.LOADER_MIN_ENTRIES.update({1: 0.0, 2: 0.0, 3: 0.0})         # small operand: the density guard would switch loaders off
            try:
                tp = GR.build_tile_plan(csr, rt, cs, block_rows=32, n_loaders=L)
            finally:
                GR.LOADER_MIN_ENTRIES.update(saved)
            slots = tp.items[:, :, 0].reshape(tp.n_tiles, 16, 16)          # [tile, wave, slot] -> row | -1
            rows_in_tile = (slots >= 0).sum((1, 2))
            assert int(rows_in_tile.max()) <= 16 * (16 - tp.n_loaders)
            if tp.n_loaders:
                assert (slots[:, : tp.n_loaders] < 0).all()
            filled = slots >= 0                                            # slots fill from 0: no gap inside a wave
            assert not (filled[:, :, 1:] & ~filled[:, :, :-1]).any()
            # decode the entries back to (row, col, val)
            seg = tp.seg_ptr.long()
            n_seg = seg.shape[0] - 1
            per = seg[1:] - seg[:-1]
            sid = torch.repeat_interleave(torch.arange(n_seg), per)
            wave = sid % 16
            blk = (sid // 16) % tp.nblk_max
            tile = sid // (16 * tp.nblk_max)
            meta, wbits = tp.entries[:, 0].long(), tp.entries[:, 1]
            dslot, src_local = (meta >> 8) & 0xF, meta & 0xFF
            paired, padded = meta < 0, (meta & GR.TILE_PAD_FLAG) != 0
            # shared pairs: the LAST entries of a segment, at even offsets, both members on one source row, the first
            # carrying the second's slot; pads: zero weight, behind an odd unshared run (in front of the pairs, if any)
            off = torch.arange(meta.shape[0]) - seg[sid]
            assert (wbits[padded] == 0).all() and not (paired & padded).any()
            assert int(paired.sum()) % 2 == 0
            first = torch.nonzero(paired & (off % 2 == 0)).squeeze(1)
            assert first.numel() * 2 == int(paired.sum())
            assert paired[first + 1].all() and (sid[first + 1] == sid[first]).all()
            assert (src_local[first + 1] == src_local[first]).all()
            assert (((meta[first] >> 16) & 0xF) == dslot[first + 1]).all()
            n_pair = torch.bincount(sid[paired], minlength=n_seg)
            assert (paired == (off >= (per - n_pair)[sid])).all()                  # pairs close their segment
            assert ((per - n_pair)[n_pair > 0] % 2 == 0).all()
            n_pad = torch.bincount(sid[padded], minlength=n_seg)
            assert (per % 2 == 0).all() and (n_pad == (per - n_pad) % 2).all()     # ABI 0.2.5: an odd segment is padded to even
            assert paired.any()
            keep = ~padded
            row = slots[tile, wave, dslot].long()
            col = tp.hdr[tile, 0].long() + blk * tp.block_rows + src_local
            got = sorted(zip(row[keep].tolist(), col[keep].tolist(), wbits[keep].view(torch.float32).tolist()))
            assert got == want, (rt, cs, L)
    crowded = GR.build_tile_plan(_cpu_agg_csr(X), 12, 1, block_rows=32, n_loaders=2)      # 250 rows per tile > 14 x 16
    assert crowded.n_loaders == 0


def test_tall_tile_plan_covers_every_edge_once():
    """Round 5: the TALL tile geometry (8 waves x 49 rows, `graph.GEOM_TALL`, kernel agg_tiled_tall).  Same plan builder, other
    constants: 392 item slots per tile, 8 segments per (tile, block), 6-bit slot fields, no dedicated loader wave; the
    re-ordered entries hold every non-zero exactly once, shared pairs close their segment.  `graph.TILE_TALL = "auto"` picks it for
    many-row operands (>= 40 000 rows) and sizes ONE round of <= 392-row tiles where that covers the rows."""
    import scipy.sparse as sp
    from scdeepsort_amd import graph as GR, _lib
    X = sp.random(2100, 500, density=0.07, format="csr", random_state=7, dtype=np.float32)
    X.data = np.abs(X.data) + 0.5
    Xd = X.toarray(); Xd[:, 5] = 1.5
    X = sp.csr_matrix(Xd)
    for A, geoms in ((X, [(6, 1), (None, None), (7, 2)]), (sp.csr_matrix(X.T), [(2, 3), (None, None)])):
        csr = _cpu_agg_csr(A)
        want = sorted(zip(np.repeat(np.arange(A.shape[0]), np.diff(sp.csr_matrix(A).indptr)).tolist(),
                          csr.col.tolist(), csr.val.tolist()))
        for rt, cs in geoms:
            tp = GR.build_tile_plan(csr, rt, cs, block_rows=40, n_loaders=1, geom=GR.GEOM_TALL)
            assert tp.geom.tall and tp.n_loaders == 0 and tp.items.shape[1] == 392
            assert tp.block_rows_arg == 40 | _lib.PLAN_TALL
            slots = tp.items[:, :, 0].reshape(tp.n_tiles, 8, 49)
            filled = slots >= 0
            assert not (filled[:, :, 1:] & ~filled[:, :, :-1]).any()
            seg = tp.seg_ptr.long()
            n_seg = seg.shape[0] - 1
            assert n_seg == tp.n_tiles * tp.nblk_max * 8
            per = seg[1:] - seg[:-1]
            sid = torch.repeat_interleave(torch.arange(n_seg), per)
            wave, blk, tile = sid % 8, (sid // 8) % tp.nblk_max, sid // (8 * tp.nblk_max)
            meta, wbits = tp.entries[:, 0].long(), tp.entries[:, 1]
            dslot, src_local = (meta >> 8) & 0x3F, meta & 0xFF
            paired, padded = meta < 0, (meta & GR.TILE_PAD_FLAG) != 0
            off = torch.arange(meta.shape[0]) - seg[sid]
            first = torch.nonzero(paired & (off % 2 == 0)).squeeze(1)
            assert first.numel() * 2 == int(paired.sum()) and paired.any()
            assert (src_local[first + 1] == src_local[first]).all() and (((meta[first] >> 16) & 0x3F) == dslot[first + 1]).all()
            n_pair = torch.bincount(sid[paired], minlength=n_seg)
            assert (paired == (off >= (per - n_pair)[sid])).all() and ((per - n_pair)[n_pair > 0] % 2 == 0).all()
            keep = ~padded
            row = slots[tile, wave, dslot].long()
            col = tp.hdr[tile, 0].long() + blk * tp.block_rows + src_local
            got = sorted(zip(row[keep].tolist(), col[keep].tolist(), wbits[keep].view(torch.float32).tolist()))
            assert got == want, (rt, cs)
    # heuristics: 100 000 rows -> one round of 256 tiles (391 rows each); 764 741 rows -> whole rounds of <= 392-row tiles
    assert GR.auto_tile_geometry(100_000, 20_000, 256, 80_000_000, 392, 392) == (256, 1)
    rt, sp_ = GR.auto_tile_geometry(764_741, 20_000, 256, 612_000_000, 392, 392)
    assert sp_ == 1 and rt % 256 == 0 and -(-764_741 // rt) <= 392
    big = _cpu_agg_csr(sp.csr_matrix((np.ones(1, np.float32), ([0], [0])), shape=(GR.TALL_MIN_ROWS, 8)))
    assert GR.TILE_TALL == "off" and not big.tile_plan(16).geom.tall          # opt-in (measured slower at cfg3, see graph.TILE_TALL)
    saved, GR.TILE_TALL = GR.TILE_TALL, "auto"
    try:
        big._tile_plan = None
        assert big.tile_plan(16).geom.tall and not _cpu_agg_csr(X).tile_plan(16).geom.tall
    finally:
        GR.TILE_TALL = saved


def test_device_built_plan_cuts_only_hub_rows():
    """the static-shape plan of a device-built (transposed / sampled) block gives every row ONE item and cuts only
    the rows longer than the chunk, into <= 8 parts on a fixed number of hub slots (unused slots = items with row -1, which
    the kernels skip) - not S items and S partial rows for every row.  Pure index arithmetic: checked on the CPU."""
    from scdeepsort_amd.graph import device_plan
    torch.manual_seed(0)
    for trial in range(24):
        n = int(torch.randint(1, 60, (1,)))
        ln = torch.randint(0, 50, (n,), dtype=torch.int32)
        if trial % 3 == 0:
            ln[torch.randint(0, n, (1,))] = 5000 + trial            # a hub row
        if trial % 5 == 0:
            ln[:] = 3                                               # no row reaches the chunk
        rp = torch.zeros(n + 1, dtype=torch.int32)
        rp[1:] = torch.cumsum(ln, 0)
        bound = int(ln.max()) + (0 if trial % 2 else 100)           # a BOUND on the row length, not necessarily tight
        p = device_plan(rp, n, max(1, bound), 16, nnz_bound=int(rp[-1]) + (7 if trial % 2 else 0))
        cover = torch.zeros(int(rp[-1]) + 1, dtype=torch.int32)
        partial_slots = set()
        for slot, b, e, ps in p.items.tolist():
            if slot < 0:
                continue
            assert rp[slot] <= b <= e <= rp[slot + 1]
            cover[b:e] += 1
            if ps >= 0:
                assert ps not in partial_slots and ps < p.n_partials
                partial_slots.add(ps)
        assert (cover[:-1] == 1).all()                              # every entry exactly once
        longs = {i for i in range(n) if ln[i] > p.chunk}
        assert {r[0] for r in p.long_rows.tolist() if r[0] >= 0} == longs
        for slot, first, cnt, _ in p.long_rows.tolist():
            if slot >= 0:
                assert cnt <= 8 and all((first + k) in partial_slots for k in range(cnt))
        direct = sorted(it[0] for it in p.items.tolist() if it[0] >= 0 and it[3] < 0)
        assert direct == [i for i in range(n) if ln[i] <= p.chunk]   # short rows: one item, written directly
        assert p.n_partials <= 8 * max(1, (int(rp[-1]) + 7) // p.chunk)
    # the every-row form (no entry bound) is unchanged
    rp = torch.tensor([0, 5, 5, 40], dtype=torch.int32)
    q = device_plan(rp, 3, 40, 16)
    assert q.items.shape[0] == 3 * 3 and q.n_partials == 9


def test_bench_workload_string_does_not_depend_on_the_rank_count():
    """the N = 1 leg of a SCALE run must name the same workload as the BENCH line.  `config.workload` is a
    pure function of the config and the scaling mode (bench.workload_string), the default mode is strong at every N, and the
    default config is BASELINE cfg3."""
    import importlib.util
    import inspect
    from pathlib import Path
    from scdeepsort_amd import synthetic as S
    spec = importlib.util.spec_from_file_location("bench_mod", Path(sda.__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    assert list(inspect.signature(bench.workload_string).parameters) == ["cfg", "total_cells", "mode"]     # no world / rank argument
    cfg = S.CONFIGS["cfg3"]
    w = bench.workload_string(cfg, cfg.cells, "strong")
    assert w == ("cfg3: 100000 cells x 20000 genes (ONE job, cells sharded over the ranks), density 0.04, dense_dim 400, "
                 "hidden 256, 2-layer WGNN forward + 16-class head")
    src = Path(bench.__file__).read_text()
    assert '"workload": workload_string(cfg, total_cells, mode)' in src
    assert 'default=os.environ.get("WGNN_BENCH_CONFIG", "cfg3")' in src and 'default=os.environ.get("WGNN_BENCH_SCALING", "strong")' in src


def test_tracked_gemm_picks_file_is_wellformed():
    """scdeepsort_amd/tuned_gemms_gfx950.csv (PyTorch TunableOp result format): validator lines naming the library stack it was
    tuned against, then one `op,shape,solution,time` line per GEMM shape of the BASELINE workloads."""
    from scdeepsort_amd import tuning
    lines = tuning.TUNED_FILE.read_text().strip().splitlines()
    vals = {l.split(",")[1]: l.split(",", 2)[2] for l in lines if l.startswith("Validator,")}
    assert {"PT_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"} <= set(vals) and "gfx950" in vals["GCN_ARCH_NAME"]
    picks = [l.split(",") for l in lines if not l.startswith("Validator,")]
    assert len(picks) >= 10 and all(len(p) == 4 and p[0].startswith("Gemm") and float(p[3]) > 0 for p in picks)
    shapes = {p[1] for p in picks}
    assert "tn_256_100000_400_ld_400_400_256" in shapes          # cfg3's largest projection: [1e5, 400] x [400, 256]
    assert not tuning.active()                                   # nothing is switched on by importing the package


def test_tuned_gemm_selection_is_a_no_op_without_a_gpu():
    from scdeepsort_amd import tuning
    if not torch.cuda.is_available():
        assert tuning.use_tuned_gemms() is False and not tuning.active()
    assert tuning.use_tuned_gemms("/nonexistent/file.csv") is False


def test_from_device_csr_guards_the_int32_range_like_from_expression():
    """the offsets of the operand are int32; a CSR with >= 2^31 non-zeros must raise, not wrap."""
    from scdeepsort_amd.graph import CellGeneGraph
    col = torch.empty(2 ** 31, dtype=torch.int32, device="meta")
    raw = torch.empty(2 ** 31, dtype=torch.float32, device="meta")
    rowptr = torch.tensor([0, 2 ** 31], dtype=torch.int64)
    with pytest.raises(ValueError, match="nnz >= 2\\^31"):
        CellGeneGraph.from_device_csr(rowptr, col, raw, 10)
    with pytest.raises(ValueError, match="entries"):
        CellGeneGraph.from_device_csr(torch.tensor([0, 2]), torch.zeros(2, dtype=torch.int32), torch.zeros(3), 10)


def test_sorted_columns_sorts_within_rows_and_rejects_repeats():
    """the device plan walk needs ascending columns inside every row; from_device_csr establishes it."""
    from scdeepsort_amd.graph import sorted_columns
    rowptr = torch.tensor([0, 3, 3, 5, 6], dtype=torch.int32)
    col = torch.tensor([4, 1, 2, 0, 3, 2], dtype=torch.int32)          # row 0 unsorted; row boundaries may descend (4 -> 0)
    raw = torch.tensor([40., 10., 20., 1., 31., 22.])
    c, r = sorted_columns(rowptr, col, raw, 5)
    assert c.tolist() == [1, 2, 4, 0, 3, 2] and r.tolist() == [10., 20., 40., 1., 31., 22.]
    c2, r2 = sorted_columns(rowptr, c, r, 5)
    assert c2 is c and r2 is r                                          # sorted input is passed through untouched
    with pytest.raises(ValueError, match="more than once"):
        sorted_columns(rowptr, torch.tensor([4, 1, 4, 0, 3, 2], dtype=torch.int32), raw, 5)
    # descending col across a row boundary is NOT an inversion; an empty operand / single entry is fine
    e = torch.zeros(0, dtype=torch.int32)
    assert sorted_columns(torch.tensor([0, 0], dtype=torch.int32), e, e.float(), 5)[0] is e


def test_synthetic_cell_range_is_a_slice_of_the_whole_matrix(monkeypatch):
    """A rank of a cell-sharded job draws only ITS rows (`cell_range`), bit-identical to slicing the whole matrix - for both
    popularity laws, ranges that cut chunks, empty and one-row ranges; the generator's thread count follows LOCAL_WORLD_SIZE."""
    for pop in ("testis199", "dense_head"):
        rp, col, val = S.synth_expression(1300, 700, seed=4, popularity=pop, chunk_cells=257)
        for lo, hi in ((0, 1300), (0, 1), (250, 900), (1299, 1300), (514, 514), (257, 514)):
            a, b, c = S.synth_expression(1300, 700, seed=4, popularity=pop, chunk_cells=257, cell_range=(lo, hi))
            s, e = int(rp[lo]), int(rp[hi])
            assert torch.equal(a, rp[lo:hi + 1] - rp[lo]) and torch.equal(b, col[s:e]) and torch.equal(c, val[s:e]), (pop, lo, hi)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    import os
    cores = len(os.sched_getaffinity(0))
    assert S._host_threads() == max(1, min(64, cores // 8))


def test_synthetic_features_come_from_numpy_default_rng():
    """node features ~ 0.5 * N(0, 1) from numpy.random.default_rng(seed) - the same rows on every machine, independent
    of the thread count (blocks of 65 536 rows from child streams); fp16 storage for cfg5; the torch device generator is kept for
    the dense_head A/B workload only."""
    a = S.synth_features(70_000, 12, seed=7)
    assert a.dtype == torch.float32 and a.shape == (70_000, 12)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 0.5) < 5e-3
    assert torch.equal(a, S.synth_features(70_000, 12, seed=7)) and not torch.equal(a, S.synth_features(70_000, 12, seed=8))
    assert torch.equal(a[:65_536], S.synth_features(65_536, 12, seed=7))            # a block does not depend on the ones behind it
    want = 0.5 * np.random.default_rng(7).spawn(2)[0].standard_normal(size=(65_536, 12), dtype=np.float32)
    assert np.array_equal(a[:65_536].numpy(), want)
    assert S.synth_features(100, 8, seed=1, dtype=torch.float16).dtype == torch.float16
    t = S.synth_features(100, 8, seed=1, rng="torch")
    assert t.shape == (100, 8) and not torch.equal(t, S.synth_features(100, 8, seed=1))
    with pytest.raises(ValueError):
        S.synth_features(4, 4, rng="nope")
