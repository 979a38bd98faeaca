import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
from scdeepsort_amd.graph import build_tile_plan
dev='cuda:0'
cfg=S.CONFIGS['cfg3']; G,C,H=cfg.genes,cfg.cells,cfg.hidden
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
def timeit(f,n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
tpc=build_tile_plan(g.cg,512,1); tpg=build_tile_plan(g.gc,80,16)
for name,fl in [('full',0),('nofill',1<<16),('nocompute',1<<17),('nocompute+nofill',(1<<16)|(1<<17)),('nofill+nobarrier',(1<<16)|(1<<18)),('nobarrier',(1<<18))]:
    ops.DEBUG_FLAGS=fl
    tc=timeit(lambda: ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    tg=timeit(lambda: ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg))
    print(f'{name:20s} cells {tc:.3f} ms   genes {tg:.3f} ms')
