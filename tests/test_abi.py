"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/wgnn.h declares, validates arguments before touching HIP, and its host-side plan
builder is correct.  No compute is launched here."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import torch

import scdeepsort_amd as sda
from scdeepsort_amd import _lib
from scdeepsort_amd.graph import build_plan

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "wgnn.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wgnn_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    syms = declared_symbols()
    assert {"wgnn_agg_fwd", "wgnn_agg_bwd_src", "wgnn_agg_bwd_alpha", "wgnn_normalize_rows",
            "wgnn_plan_build_host", "wgnn_version", "wgnn_last_error_string"} <= set(syms)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in wgnn.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.py"
    assert lib.wgnn_version() == 206 and lib.wgnn_version() >= _lib.ABI_MIN
    assert b"ok" == lib.wgnn_last_error_string(0)


def test_header_enums_match_python():
    text = (ROOT / "include" / "wgnn.h").read_text()
    def val(name):
        return int(re.search(rf"#define\s+{name}\s+(-?\d+)", text).group(1))
    assert (val("WGNN_SRC_IS_GENE"), val("WGNN_DST_IS_GENE"), val("WGNN_NO_ALPHA")) == (sda.SRC_IS_GENE, sda.DST_IS_GENE, sda.NO_ALPHA)
    assert (val("WGNN_F32"), val("WGNN_F16")) == (_lib.F32, _lib.F16)
    assert (val("WGNN_FLAG_RELU"), val("WGNN_FLAG_NO_MEAN"), val("WGNN_FLAG_NO_SELF"), val("WGNN_FLAG_SELF_COMPACT")) == \
           (_lib.FLAG_RELU, _lib.FLAG_NO_MEAN, _lib.FLAG_NO_SELF, _lib.FLAG_SELF_COMPACT)
    assert (val("WGNN_FLAG_ROWPTR_I64"), val("WGNN_FLAG_SRC_PRESCALED")) == (_lib.FLAG_ROWPTR_I64, _lib.FLAG_SRC_PRESCALED)
    assert val("WGNN_VERSION") // 100 == _lib.ABI_MAJOR


def test_argument_validation_returns_error_codes_without_gpu():
    lib = _lib.lib()
    one = C.c_void_p(16)      # fake, aligned, never dereferenced: validation happens first
    def fwd(D=8, ld=8, mode=0, alpha=one, n_items=0, dtype=0):
        return lib.wgnn_agg_fwd(one, one, one, alpha, mode, 0, one, ld, one, ld, None, None, None, one, ld, None, 4, D,
                                dtype, 0, 0, None, n_items, None, 0, None, 0, None)
    assert fwd(D=6) == -2                       # D % 4
    assert fwd(ld=6) == -2                      # ld % 4
    assert fwd(D=2048, ld=2048) == -3           # too wide
    assert fwd(mode=7) == -1
    assert fwd(alpha=None) == -1                # alpha required unless NO_ALPHA
    assert fwd(n_items=3) == -1                 # items missing
    assert fwd(dtype=5) == -3
    assert lib.wgnn_normalize_rows(None, one, one, None, 4, None) == -1
    assert lib.wgnn_normalize_rows_i64(None, one, one, None, 4, None) == -1
    lin = lambda **k: lib.wgnn_linear_fwd_ex(k.get("x", one), k.get("dt", 0), 8, one, 8, None, k.get("out", one), 8, k.get("rs"),
                                             k.get("o2"), 8, 4, 8, k.get("K", 8), k.get("fl", 0), None)
    assert lin(K=6) == -2                        # K % 4
    assert lin(dt=3) == -1                       # unknown storage type
    assert lin(rs=one) == -1 and lin(o2=one) == -1      # row_scale and out_scaled come together
    assert lin(out=None) == -1                   # no output at all
    assert lin(fl=2) == -1                       # only WGNN_FLAG_RELU
    assert lin(x=C.c_void_p(8), dt=1, K=0) == -1
    assert lib.wgnn_normalize_rows(one, one, one, None, 0, None) == 0      # empty input is a no-op
    # tile-plan construction (ABI 0.2.5): the two plan geometries only, 16 <= block_rows <= 255, range check on the tile
    # count ahead of everything else; seg_pairs is ignored and may be NULL; an empty plan is a no-op
    def count(n_flat=4, waves=16, rpw=16, nblk=8, kb=78, seg_total=one, rowptr=one):
        return lib.wgnn_tile_plan_count(rowptr, one, one, one, one, one, one, one, n_flat, waves, rpw, nblk, kb, seg_total, None, None)
    def fill(n_flat=4, waves=16, rpw=16, nblk=8, kb=78, seg_ptr=one, entries=one, val=one):
        return lib.wgnn_tile_plan_fill(one, one, val, one, one, one, one, one, one, n_flat, waves, rpw, nblk, kb, one, None, seg_ptr, entries, None)
    assert count(waves=4) == -1 and count(waves=16, rpw=64) == -1 and count(waves=8, rpw=16) == -1      # not a plan geometry
    assert count(kb=8) == -1 and count(kb=256) == -1 and count(nblk=0) == -1
    assert count(seg_total=None) == -1 and count(rowptr=None) == -1
    assert count(n_flat=-1) == -1 and count(n_flat=2 ** 31) == -3
    assert count(n_flat=0) == 0 and count(n_flat=0, waves=8, rpw=49) == 0
    assert fill(waves=5) == -1 and fill(kb=300) == -1 and fill(n_flat=2 ** 31) == -3
    assert fill(seg_ptr=None) == -1 and fill(entries=None) == -1 and fill(val=None) == -1
    assert fill(n_flat=0) == 0
    # stable CSR transpose (0.2.6): counters for <= 32768 columns, everything else is the caller's sort
    n_ch, n_b = C.c_int64(), C.c_int64()
    assert lib.wgnn_csr_transpose_workspace(100000, 20000, C.addressof(n_ch), C.addressof(n_b)) == 0
    assert n_ch.value == 512 and n_b.value == 512 * 20000 * 4
    assert lib.wgnn_csr_transpose_workspace(100, 50, C.addressof(n_ch), C.addressof(n_b)) == 0 and n_ch.value == 1
    assert lib.wgnn_csr_transpose_workspace(100, 40000, C.addressof(n_ch), C.addressof(n_b)) == -3
    assert lib.wgnn_csr_transpose_workspace(-1, 5, C.addressof(n_ch), C.addressof(n_b)) == -1
    assert lib.wgnn_csr_transpose_count(None, one, None, 4, 8, 1, one, one, None) == -1
    assert lib.wgnn_csr_transpose_count(one, one, None, 4, 40000, 1, one, one, None) == -3
    assert lib.wgnn_csr_transpose_fill(one, one, None, None, 4, 8, 1, one, one, one, one, None) == -1
    for code in (-1, -2, -3, -4, -5, -6):
        assert len(lib.wgnn_last_error_string(code)) > 3


def test_plan_builder_chunks_long_rows():
    nnz = np.array([0, 5, 2048, 2049, 10000, 1, 0, 4096])
    rowptr = np.concatenate([[0], np.cumsum(nnz)]).astype(np.int32)
    plan = build_plan(rowptr, chunk=2048)
    items, longs = plan.items.numpy(), plan.long_rows.numpy()
    # every row is covered exactly once, in order, by contiguous chunks of <= chunk nnz
    covered = {r: [] for r in range(len(nnz))}
    for r, b, e, slot in items:
        assert 0 <= e - b <= 2048
        covered[r].append((b, e, slot))
    for r, n in enumerate(nnz):
        segs = covered[r]
        assert segs[0][0] == rowptr[r] and segs[-1][1] == rowptr[r + 1]
        for (b0, e0, _), (b1, e1, _) in zip(segs, segs[1:]):
            assert e0 == b1
        if n <= 2048:
            assert len(segs) == 1 and segs[0][2] == -1
        else:
            assert len(segs) == -(-n // 2048) and all(s[2] >= 0 for s in segs)
    assert [int(l[0]) for l in longs] == [3, 4, 7]
    assert [int(l[2]) for l in longs] == [2, 5, 2]
    assert plan.n_partials == 9
    slots = sorted(int(s[2]) for segs in covered.values() for s in segs if s[2] >= 0)
    assert slots == list(range(9))
    for r, first, n, _ in longs:
        assert [s[2] for s in covered[r]] == list(range(first, first + n))


def test_plan_builder_i64_rowptr_and_2g_guard():
    """64-bit row pointers (SURVEY 8b "rowptr i32 or i64"): same plan as the 32-bit builder; a row reaching past 2^31
    non-zeros is refused with WGNN_ERR_UNSUPPORTED instead of being truncated."""
    import ctypes as C
    from scdeepsort_amd import _lib
    lib = _lib.lib()
    nnz = np.array([0, 5, 2048, 2049, 10000, 1])
    rp64 = np.concatenate([[0], np.cumsum(nnz)]).astype(np.int64)
    ni, nl, npart = C.c_int64(), C.c_int64(), C.c_int64()
    assert lib.wgnn_plan_build_host_i64(rp64.ctypes.data, None, len(nnz), 2048, None, None, C.addressof(ni), C.addressof(nl),
                                        C.addressof(npart)) == 0
    items = np.empty((ni.value, 4), np.int32); longs = np.empty((nl.value, 4), np.int32)
    assert lib.wgnn_plan_build_host_i64(rp64.ctypes.data, None, len(nnz), 2048, items.ctypes.data, longs.ctypes.data,
                                        C.addressof(ni), C.addressof(nl), C.addressof(npart)) == 0
    ref = build_plan(rp64.astype(np.int32), chunk=2048)
    np.testing.assert_array_equal(items, ref.items.numpy()); np.testing.assert_array_equal(longs, ref.long_rows.numpy())
    big = np.array([0, 2 ** 31 - 10, 2 ** 31 + 5], dtype=np.int64)
    assert lib.wgnn_plan_build_host_i64(big.ctypes.data, None, 2, 2048, None, None, C.addressof(ni), C.addressof(nl),
                                        C.addressof(npart)) == -3
    assert b"2^31" in lib.wgnn_last_error_string(-3)


def test_plan_builder_row_subset():
    nnz = np.array([3, 0, 7, 5000, 2])
    rowptr = np.concatenate([[0], np.cumsum(nnz)]).astype(np.int32)
    ids = np.array([3, 0, 3], dtype=np.int32)          # repeated seed, arbitrary order
    plan = build_plan(rowptr, chunk=2048, row_ids_host=ids)
    items = plan.items.numpy()
    assert sorted(set(items[:, 0].tolist())) == [0, 1, 2]        # slots, not row ids
    assert plan.n_long == 2 and plan.n_partials == 6
    for slot, b, e, _ in items:
        r = ids[slot]
        assert rowptr[r] <= b <= e <= rowptr[r + 1]


def test_product_path_fails_loudly_without_gpu_tensors():
    """No CPU fallback: CPU tensors (or a missing .so) must raise, never silently compute."""
    rowptr = np.array([0, 1, 2], dtype=np.int32)
    plan = build_plan(rowptr)
    csr = sda.AggCsr(torch.tensor([0, 1, 2], dtype=torch.int32), torch.tensor([0, 1], dtype=torch.int32),
                     torch.ones(2), torch.ones(2), 2, 2, plan, rowptr)
    with pytest.raises(sda.WgnnError):
        sda.agg_fwd(csr, torch.ones(4), sda.SRC_IS_GENE, 3, torch.ones(2, 4), torch.ones(2, 4))
    import scipy.sparse as sp
    with pytest.raises(sda.WgnnError):
        sda.CellGeneGraph.from_expression(sp.csr_matrix(np.eye(3, dtype=np.float32)), device="cpu")


def test_no_product_module_imports_the_oracle():
    for f in (ROOT / "scdeepsort_amd").rglob("*.py"):
        src = f.read_text()
        assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f


def test_workspace_query():
    import ctypes as C
    from scdeepsort_amd import _lib
    lib = _lib.lib()
    pb, sb = C.c_int64(), C.c_int64()
    assert lib.wgnn_agg_workspace_bytes(10, 1000, 256, _lib.SRC_IS_GENE, 1, C.addressof(pb), C.addressof(sb)) == 0
    assert (pb.value, sb.value) == (10 * 256 * 4, 1000 * 256 * 4)
    assert lib.wgnn_agg_workspace_bytes(0, 1000, 64, _lib.DST_IS_GENE, 1, C.addressof(pb), C.addressof(sb)) == 0
    assert (pb.value, sb.value) == (0, 0)
    assert lib.wgnn_agg_workspace_bytes(1, 1, 6, 0, 0, C.addressof(pb), C.addressof(sb)) == -2      # WGNN_ERR_ALIGNMENT
    assert lib.wgnn_agg_workspace_bytes(-1, 1, 8, 0, 0, C.addressof(pb), C.addressof(sb)) == -1     # WGNN_ERR_BAD_ARG


def test_flat4_register_contract_is_enforced_at_build_time():
    """agg_tiled_flat4 splits the register file by hand (compiler v[0:31] / s[0:79]; accumulators, staging and
    chunk registers above).  The build audits every instantiation: `-Rpass-analysis=kernel-resource-usage` says VGPRs 128,
    scratch 0, spills 0; the code-object metadata agrees; and no compiler-emitted instruction (outside the inline-asm
    regions) names v32..v127 - round 3's build parked 3-20 spilled SGPRs in a VGPR of the hand-owned file."""
    from scdeepsort_amd import build as B
    usage = B.flat4_resource_usage()
    # 3 epilogues x {production, ablation}, for each of the two tile geometries (round 5: agg_tiled_tall - 8 waves x 256 VGPRs,
    # the compiler capped at v[0:41], v[42:255] carry state across statements)
    assert len(usage) == 12 and sum("agg_tiled_flat4" in k for k in usage) == 6 and sum("agg_tiled_tall" in k for k in usage) == 6
    for name, rec in usage.items():
        rm, md = rec["remarks"], rec["metadata"]
        tall = "agg_tiled_tall" in name
        assert rm["VGPRs"] == ("256" if tall else "128") and rm["ScratchSize"] == "0", (name, rm)
        assert rm["SGPRs Spill"] == "0" and rm["VGPRs Spill"] == "0" and rm["Occupancy"] == ("2" if tall else "4"), (name, rm)
        assert md["sgpr_spill_count"] == 0 and md["vgpr_spill_count"] == 0 and md["private_segment_fixed_size"] == 0, (name, md)
        assert rec["compiler_touches_hand_registers"] == [], (name, rec["compiler_touches_hand_registers"][:5])
    B.audit_flat4(usage)
    # the audit does reject a violation: a fake record with one spilled SGPR / a compiler write into the accumulators
    import copy
    bad = copy.deepcopy(usage)
    next(iter(bad.values()))["metadata"]["sgpr_spill_count"] = 1
    with pytest.raises(B.RegisterContractError):
        B.audit_flat4(bad)
    bad = copy.deepcopy(usage)
    next(iter(bad.values()))["compiler_touches_hand_registers"] = ["v_writelane_b32 v47, s8, 0"]
    with pytest.raises(B.RegisterContractError):
        B.audit_flat4(bad)


def test_hand_written_statements_name_every_owned_register():
    """The clobber lists of the flat kernel's asm statements spell out v32..v127 / s80..s95 one by one (a clobber list takes
    single registers: `"v48", "v63"` names two registers, not a range)."""
    inc = (ROOT / "scdeepsort_amd" / "csrc" / "wgnn_flat_asm.inc").read_text()
    vl = next(l for l in inc.splitlines() if l.startswith("#define WGNN_HAND_VGPRS"))
    sl = next(l for l in inc.splitlines() if l.startswith("#define WGNN_HAND_SGPRS"))
    assert [f'"v{i}"' for i in range(32, 128)] == [t.strip() for t in vl.split("WGNN_HAND_VGPRS", 1)[1].split(",")]
    assert [f'"s{i}"' for i in range(80, 96)] == [t.strip() for t in sl.split("WGNN_HAND_SGPRS", 1)[1].split(",")]
    src = (ROOT / "scdeepsort_amd" / "csrc" / "wgnn_tiled.hip").read_text()
    assert '#define WGNN_CLOB "m0", "memory", "scc", WGNN_HAND_VGPRS, WGNN_HAND_SGPRS' in src
    tl = next(l for l in inc.splitlines() if l.startswith("#define WGNN_TALL_VGPRS"))
    assert [f'"v{i}"' for i in range(20, 256)] == [t.strip() for t in tl.split("WGNN_TALL_VGPRS", 1)[1].split(",")]
    assert '#define WGNN_TALL_CLOB "m0", "memory", "scc", WGNN_TALL_VGPRS, WGNN_HAND_SGPRS' in src
    assert "__attribute__((amdgpu_num_vgpr(21)))" in src          # the tall kernel: compiler v[0:41]
    assert "__attribute__((amdgpu_num_vgpr(16)))" in src and "amdgpu_num_sgpr" not in src.split("agg_tiled_flat4(const KArgs")[0][-300:]
