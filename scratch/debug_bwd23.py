"""Reproduce fuzz_bwd.py seed 3 draw 23 and compare both routes with a dense torch autograd evaluation."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import scipy.sparse as sp
import scdeepsort_amd as sda
from scdeepsort_amd import ops
dev = 'cuda:0'
rng = np.random.default_rng(3)
for it in range(24):
    C = int(rng.integers(20, 2500)); G = int(rng.integers(10, 1200))
    dens = float(rng.choice([rng.uniform(0.005, 0.4), rng.uniform(0.4, 0.98)], p=[0.75, 0.25]))
    D = int(rng.choice([256, 256, 128, 64, 200, 32, 100, 132, 192, 16]))
    m = rng.random((C, G)) < dens
    if rng.random() < 0.5: m[:, rng.integers(0, G)] = True
    if rng.random() < 0.5: m[rng.integers(0, C), :] = False
    x = sp.csr_matrix(np.where(m, rng.uniform(0.5, 7, (C, G)), 0).astype(np.float32))
    relu = bool(rng.random() < 0.6); use_bias = bool(rng.random() < 0.6)
    fused = bool(rng.random() < 0.8)
print(it, C, G, dens, D, relu, use_bias, fused)
g = sda.CellGeneGraph.from_expression(x, device=dev)
ops.FUSED_BWD_GLUE = fused
csr, mode, si, ns, nd = g.cg, sda.SRC_IS_GENE, G + 1, G, C
gen = torch.Generator(device=dev).manual_seed(it)
base = dict(hs=torch.randn(ns, D, device=dev, generator=gen), hd=torch.randn(nd, D, device=dev, generator=gen),
            al=torch.rand(G + 2, 1, device=dev, generator=gen) + 0.5, b=torch.randn(D, device=dev, generator=gen))
r = torch.randn(nd, D, device=dev, generator=gen)
res = {}
for route, thr in (("rowwave", None), ("tiled", 1)):
    ops.TILED_MIN_WORK = thr
    t = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    out = ops.weighted_mean_aggregate(csr, t["al"], mode, si, t["hs"], t["hd"], bias=None, relu=relu)
    (out * r).sum().backward()
    res[route] = (out.detach(), t["hs"].grad, t["hd"].grad, t["al"].grad)
# dense reference in fp64
A = torch.zeros(C, G, dtype=torch.float64, device=dev)
rows = torch.repeat_interleave(torch.arange(C, device=dev), (csr.rowptr[1:] - csr.rowptr[:-1]).long())
A[rows, csr.col.long()] = csr.val.double()
t = {k: v.clone().double().requires_grad_(True) for k, v in base.items()}
a = t["al"].reshape(-1)
z = (A @ (a[:G, None] * t["hs"]) + a[G + 1] * t["hd"]) * csr.inv_deg.double()[:, None]
o = torch.relu(z) if relu else z
(o * r.double()).sum().backward()
ref = (o.detach(), t["hs"].grad, t["hd"].grad, t["al"].grad)
for name, i in (("out", 0), ("dh_src", 1), ("dh_self", 2), ("dalpha", 3)):
    for route in ("rowwave", "tiled"):
        d = (res[route][i].double() - ref[i]).abs()
        print(name, route, "max err", float(d.max()), "scale", float(ref[i].abs().max()), "n_bad", int((d > 1e-3).sum()))
d = (res["tiled"][1].double() - ref[1]).abs().max(1).values
bad = torch.nonzero(d > 1e-3).reshape(-1)
deg = torch.bincount(csr.col.long(), minlength=G)
print("bad gene rows", bad[:30].tolist(), "deg", deg[bad[:30]].tolist())
d2 = (res["rowwave"][1].double() - ref[1]).abs().max(1).values
bad2 = torch.nonzero(d2 > 1e-3).reshape(-1)
print("bad gene rows rowwave", bad2[:30].tolist(), "deg", deg[bad2[:30]].tolist())
tt = csr.transposed()
tp = tt.tile_plan(ops.tiled_block_rows(D))
print("tplan tiles", tp.n_tiles, "row tiles", tp.n_row_tiles, "splits", tp.n_col_splits, "loaders", tp.n_loaders, "partials", tp.n_partials, "long", tp.long_rows.shape, "blk", tp.block_rows, "nblk_max", tp.nblk_max)
