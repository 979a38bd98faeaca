import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev='cuda:0'
cfg=S.CONFIGS['cfg3']; G,C=cfg.genes,cfg.cells; H=256
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
def timeit(f,n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
kb=78
tpc=g.cg.tile_plan(kb); tpg=g.gc.tile_plan(kb)
B=lambda *bits: sum(1<<b for b in bits)
for nm,fl in [('full',0),('nofill',B(16)),('nofill+nobar',B(16,18)),('noentry',B(21)),('noentry+nofill',B(21,16)),('noentry+nofill+nobar',B(21,16,18)),('noentry+nobar',B(21,18)),('nocompute',B(17)),('nocompute+noentry',B(17,21)),('nocompute+noentry+nofill',B(17,21,16))]:
    ops.DEBUG_FLAGS=fl
    tc=timeit(lambda: ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    tg=timeit(lambda: ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg))
    print(f'{nm:28s} cells {tc:.3f}  genes {tg:.3f}', flush=True)
