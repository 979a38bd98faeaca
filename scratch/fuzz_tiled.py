import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import scipy.sparse as sp
import scdeepsort_amd as sda
from scdeepsort_amd import ops
from scdeepsort_amd.graph import build_tile_plan
from scdeepsort_amd import graph as GR
dev='cuda:0'
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
ops.TILED_MIN_WORK=None
worst=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 60):
    C=int(rng.integers(20,3000)); G=int(rng.integers(10,1500)); dens=float(rng.choice([rng.uniform(0.005,0.4), rng.uniform(0.4,0.98)], p=[0.75,0.25]))
    D=int(rng.choice([256,256,128,64,200,32,100,132,192,16]))          # all three LDS row strides (round 4: 256 / 512 / 1024 B)
    m=rng.random((C,G))<dens
    if rng.random()<0.5: m[:, rng.integers(0,G)] = True            # a hub gene
    if rng.random()<0.5: m[rng.integers(0,C), :] = False            # an empty cell
    x=sp.csr_matrix(np.where(m, rng.uniform(0.5,7,(C,G)),0).astype(np.float32))
    if x.nnz==0: continue
    g=sda.CellGeneGraph.from_expression(x, device=dev)
    alpha=torch.rand(G+2,device=dev)+0.5
    hg=torch.randn(G,D,device=dev); hc=torch.randn(C,D,device=dev)
    kb_max=ops.tiled_block_rows(D)                                     # packed rows: up to 255 / 156 / 78 source rows per block
    kb=int(rng.integers(16,kb_max+1)) if rng.random()<0.7 else kb_max
    osa=bool(rng.random()<0.3)                                          # WGNN_FLAG_OUT_SCALE_ALPHA on the gene pass
    for csr,mode,si,hs,hself in ((g.cg,sda.SRC_IS_GENE,G+1,hg,hc),(g.gc,sda.DST_IS_GENE,G,hc,hg)):
        rt=int(rng.integers(1,8)); cs=int(rng.integers(1,6))
        GR.TILE_SHARED_PAIRS = bool(rng.random() < 0.8)               # shared pairs (round 3) on most draws
        L = int(rng.choice([0, 0, 1, 2, 3]))                          # dedicated loader waves (falls back to 0 if the tile is too tall)
        tall = bool(rng.random() < 0.5) and GR.TILE_SHARED_PAIRS            # round 5: the tall tile geometry (8 waves x 50 rows)
        geom = GR.GEOM_TALL if tall else GR.GEOM_FLAT
        tp=build_tile_plan(csr, None if rng.random()<0.3 else max(rt,-(-csr.n_rows//geom.rows)), cs, block_rows=kb, n_loaders=L, geom=geom)
        kw=dict(out_scale_alpha=True) if (osa and mode==sda.DST_IS_GENE) else {}
        ref=ops.agg_fwd(csr,alpha,mode,si,hs,hself,**kw)
        out=ops.agg_fwd_tiled(csr,tp,alpha,mode,si,hs,hself,**kw)
        err=float((ref-out).abs().max()); worst=max(worst,err)
        if not err < 1e-4:
            print('MISMATCH',it,C,G,dens,D,kb,rt,cs,L,GR.TILE_SHARED_PAIRS,tall,err); sys.exit(1)
print('fuzz ok, worst abs diff', worst)
