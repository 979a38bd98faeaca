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
ops.TILED_MIN_WORK=None
ref_c=sda.agg_fwd(g.cg,alpha,sda.SRC_IS_GENE,G+1,hg,hc); ref_g=sda.agg_fwd(g.gc,alpha,sda.DST_IS_GENE,G,hc,hg)
def timeit(f,n=6):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for kb in (48,64,72,80):
    tpc=build_tile_plan(g.cg,512,1,block_rows=kb)
    o=ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc)
    tc=timeit(lambda: ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    for geom in ((80,16),(80,12),(160,8)):
        tpg=build_tile_plan(g.gc,*geom,block_rows=kb)
        og=ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg)
        tg=timeit(lambda: ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg))
        print(f'kb={kb} cells {tc:.3f} ms err {(o-ref_c).abs().max().item():.1e} | genes{geom} {tg:.3f} ms err {(og-ref_g).abs().max().item():.1e}', flush=True)
    del tpc
