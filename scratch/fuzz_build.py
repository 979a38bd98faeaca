"""Randomised checks of the graph-build kernels of round 6 against the framework paths: the stable transpose (with / without a
support mask) bit for bit, and the tile plans of both directions (plan kernels vs the framework builder: same segment boundaries,
same (row, column, weight) multiset per segment, layout invariants).  usage: fuzz_build.py SEED ITERS"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import scdeepsort_amd as sda
from scdeepsort_amd import graph as GR, synthetic as S
dev = "cuda:0"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_plans = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    C = int(rng.integers(1, 6000)); G = int(rng.integers(1, 3000))
    dens = float(rng.choice([rng.uniform(0.002, 0.3), rng.uniform(0.3, 0.98)], p=[0.8, 0.2]))
    m = torch.rand(C, G, device=dev) < dens
    if rng.random() < 0.5: m[:, int(rng.integers(0, G))] = True
    if rng.random() < 0.5: m[int(rng.integers(0, C)), :] = False
    if rng.random() < 0.2: m[int(rng.integers(0, C)), :] = True               # a cell expressing every gene (> 1024 entries when G is large)
    rows, cols = torch.nonzero(m, as_tuple=True)
    if rows.numel() == 0: continue
    rp = torch.zeros(C + 1, dtype=torch.int32, device=dev); rp[1:] = torch.cumsum(torch.bincount(rows, minlength=C), 0).to(torch.int32)
    col = cols.to(torch.int32).contiguous(); val = (torch.rand(col.shape[0], device=dev) * 6.5 + 0.5)
    mask = None if rng.random() < 0.4 else (torch.rand(C, device=dev) < rng.uniform(0.05, 0.95))
    a = GR._transpose_on_device(rp, col, val, C, G, mask); b = GR._transpose_by_sort(rp, col, val, C, G, mask)
    for x, y, name in zip(a, b, ("t_rowptr", "t_col", "t_raw")):
        if not torch.equal(x, y):
            print("TRANSPOSE MISMATCH", it, name, C, G, dens, mask is not None); sys.exit(1)
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G, support_mask=mask)
    for csr in (g.cg, g.gc):
        if csr.nnz == 0: continue
        geom = GR.GEOM_TALL if rng.random() < 0.4 else GR.GEOM_FLAT
        rt = None if rng.random() < 0.3 else int(rng.integers(1, 9)) * max(1, -(-csr.n_rows // geom.rows) // int(rng.integers(1, 4)) or 1)
        rt = None if rt is None else max(rt, -(-csr.n_rows // geom.rows))
        cs = None if rt is None else int(rng.integers(1, 6))
        kb = int(rng.integers(16, 256)); L = 0 if geom.tall else int(rng.choice([0, 1, 2]))
        plans = {}
        try:
            for kern in (True, False):
                GR.TILE_PLAN_KERNEL = kern
                plans[kern] = GR.build_tile_plan(csr, rt, cs, block_rows=kb, n_loaders=L, geom=geom)
        except ValueError as ex:                                            # ("tile overflow": more virtual rows than the requested tiles hold)
            continue
        finally:
            GR.TILE_PLAN_KERNEL = True
        p, q = plans[True], plans[False]
        ok = torch.equal(p.seg_ptr, q.seg_ptr) and torch.equal(p.items, q.items) and p.entries.shape == q.entries.shape
        if ok:
            W, rpw = geom.waves, geom.rpw
            seg = p.seg_ptr.long(); per = seg[1:] - seg[:-1]
            sid = torch.repeat_interleave(torch.arange(per.shape[0], device=dev), per)
            def key(tp):
                meta, wb = tp.entries[:, 0].long(), tp.entries[:, 1].long()
                pad = (meta & GR.TILE_PAD_FLAG) != 0
                k = ((sid * 64 + ((meta >> 8) & 0x3F)) * 256 + (meta & 0xFF)) * (1 << 32) + (wb & 0xFFFFFFFF)
                return torch.sort(k[~pad]).values, int(pad.sum())
            (ka, pa), (kb_, pb) = key(p), key(q)
            ok = torch.equal(ka, kb_) and pa == pb and bool((per % 2 == 0).all())
        if not ok:
            print("PLAN MISMATCH", it, C, G, dens, geom, rt, cs, kb, L); sys.exit(1)
        n_plans += 1
print("build fuzz ok:", n_plans, "plan pairs")
