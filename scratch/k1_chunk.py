import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev='cuda:0'
name=sys.argv[1] if len(sys.argv)>1 else 'cfg2'
cfg=S.CONFIGS[name]; G,C,H=cfg.genes,cfg.cells,cfg.hidden
rp,col,val=S.synth_expression(C,G,device=dev)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
ops.TILED_MIN_WORK=None
def timeit(f,n=20):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for chunk in (2048,1024,512,256,128):
    g=sda.CellGeneGraph.from_device_csr(rp,col,val,G,chunk)
    tc=timeit(lambda: ops.agg_fwd(g.cg,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    tg=timeit(lambda: ops.agg_fwd(g.gc,alpha,sda.DST_IS_GENE,G,hc,hg))
    print(f"{name} chunk {chunk}: K1 cells {tc*1e3:.1f} us  genes {tg*1e3:.1f} us  items {g.cg.plan.n_items}/{g.gc.plan.n_items}", flush=True)
