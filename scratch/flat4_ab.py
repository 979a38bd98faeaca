import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
from scdeepsort_amd.graph import build_tile_plan
dev='cuda:0'
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
cfg=S.CONFIGS[name]; G,C=cfg.genes,cfg.cells; H=256
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
def timeit(f,n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
kb=int(sys.argv[2]) if len(sys.argv) > 2 else 78
tpc=g.cg.tile_plan(kb); tpg=g.gc.tile_plan(kb)
res={}
for nm,fl in [('generic',1<<19),('flat4',0)]:
    ops.DEBUG_FLAGS=fl
    zc=ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc)
    zg=ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg)
    torch.cuda.synchronize()
    res[nm]=(zc,zg)
    tc=timeit(lambda: ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    tg=timeit(lambda: ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg))
    print(f'{nm:8s} cells {tc:.3f} ms   genes {tg:.3f} ms', flush=True)
print('max abs diff cells', float((res['generic'][0]-res['flat4'][0]).abs().max()), 'genes', float((res['generic'][1]-res['flat4'][1]).abs().max()))
print('bit-identical', torch.equal(res['generic'][0],res['flat4'][0]), torch.equal(res['generic'][1],res['flat4'][1]))
for nm,fl in [('flat4 nofill',1<<16),('flat4 nofill+nobarrier',(1<<16)|(1<<18)),('flat4 nocompute',1<<17)]:
    ops.DEBUG_FLAGS=fl
    tc=timeit(lambda: ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    tg=timeit(lambda: ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg))
    print(f'{nm:24s} cells {tc:.3f} ms   genes {tg:.3f} ms', flush=True)
