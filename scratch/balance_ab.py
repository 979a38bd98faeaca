import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
from scdeepsort_amd.graph import build_tile_plan
dev='cuda:0'
G,C,H=20000,100000,256
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
def timeit(f,n=20):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for rep in range(2):
  for name,kw in [('unbalanced 80x16',dict(n_row_tiles=80,n_col_splits=16,balance=False)),('balanced auto',dict(n_row_tiles=None,n_col_splits=None)),
                ('balanced 83x15',dict(n_row_tiles=83,n_col_splits=15)),('balanced 85x15',dict(n_row_tiles=85,n_col_splits=15)),('balanced 128x10',dict(n_row_tiles=128,n_col_splits=10)),('balanced 102x12',dict(n_row_tiles=102,n_col_splits=12))]:
    tp=build_tile_plan(g.gc,block_rows=78,**kw)
    t=timeit(lambda: ops.agg_fwd_tiled(g.gc,tp,alpha,sda.DST_IS_GENE,G,hc,hg))
    print(f"{name:20s} tiles {tp.items.shape[0]:5d} partials {tp.n_partials:7d}  {t:.3f} ms", flush=True)
