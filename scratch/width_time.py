import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev='cuda:0'
G,C=20000,100000
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
def timeit(f,n=10):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for H in (64,128,192,200,256):
    hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
    kb=ops.tiled_block_rows(H)
    tpc=g.cg.tile_plan(kb); tpg=g.gc.tile_plan(kb)
    tc=timeit(lambda: ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    tg=timeit(lambda: ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg))
    print(f"D={H}: tiled cells {tc:.3f} ms genes {tg:.3f} ms (kb {kb})", flush=True)
