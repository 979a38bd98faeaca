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
def timeit(f,n=10):
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
kb=78
tpc=g.cg.tile_plan(kb); tpg=g.gc.tile_plan(kb)
B=lambda *bits: sum(1<<b for b in bits)
for rep in range(1):
  for nm,fl in [('full',0),('nofill',B(16)),('nobarrier(racy)',B(18)),('nofill+nobar',B(16,18)),('nocompute',B(17)),('nocompute+nobar',B(17,18)),('nocompute+nofill',B(17,16)),('fill->vgpr (no LDS writes)',B(20)),('fill->vgpr + nobar',B(20,18)),('fill->vgpr + nocompute',B(20,17))]:
    ops.DEBUG_FLAGS=fl
    tc=timeit(lambda: ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    tg=timeit(lambda: ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg))
    print(f'{nm:28s} cells {tc:.3f}  genes {tg:.3f}', flush=True)
# per-(wave, block) load statistics of the cells plan: mean and max over the 16 waves of a block
seg=tpc.seg_ptr.long()
cnt=(seg[1:]-seg[:-1]).reshape(-1,16).float()
print("entries per (wave,block): mean %.1f  mean of max-over-16-waves %.1f  ratio %.3f" % (cnt.mean().item(), cnt.max(1).values.mean().item(), (cnt.max(1).values.mean()/cnt.mean()).item()))
