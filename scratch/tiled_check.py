import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
from scdeepsort_amd.graph import build_tile_plan
dev='cuda:0'
name = sys.argv[1] if len(sys.argv)>1 else 'cfg2'
cfg=S.CONFIGS[name]; G,C,H=cfg.genes,cfg.cells,cfg.hidden
H = int(sys.argv[2]) if len(sys.argv)>2 else H
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
bias=torch.randn(H,device=dev)
def timeit(f,n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
ops.TILED_MIN_WORK=None
ref_c=sda.agg_fwd(g.cg,alpha,sda.SRC_IS_GENE,G+1,hg,hc,bias=bias,relu=True)
ref_g=sda.agg_fwd(g.gc,alpha,sda.DST_IS_GENE,G,hc,hg,bias=bias,relu=True)
print('v1 cells ms',timeit(lambda: sda.agg_fwd(g.cg,alpha,sda.SRC_IS_GENE,G+1,hg,hc,bias=bias,relu=True)),
      'genes ms',timeit(lambda: sda.agg_fwd(g.gc,alpha,sda.DST_IS_GENE,G,hc,hg,bias=bias,relu=True)))
for nrt,ncs in [(None,1),(512,1),(1024,1)]:
    tp=build_tile_plan(g.cg,nrt,ncs)
    o=ops.agg_fwd_tiled(g.cg,tp,alpha,sda.SRC_IS_GENE,G+1,hg,hc,bias=bias,relu=True)
    print('cells tiles',tp.n_tiles,'err',(o-ref_c).abs().max().item(),'ms',timeit(lambda: ops.agg_fwd_tiled(g.cg,tp,alpha,sda.SRC_IS_GENE,G+1,hg,hc,bias=bias,relu=True)))
for nrt,ncs in [(None,1),(128,2),(128,4),(128,8),(256,4),(80,16)]:
    tp=build_tile_plan(g.gc,nrt,ncs)
    o=ops.agg_fwd_tiled(g.gc,tp,alpha,sda.DST_IS_GENE,G,hc,hg,bias=bias,relu=True)
    print('genes tiles',tp.n_tiles,(tp.n_row_tiles,ncs),'err',(o-ref_g).abs().max().item(),'ms',timeit(lambda: ops.agg_fwd_tiled(g.gc,tp,alpha,sda.DST_IS_GENE,G,hc,hg,bias=bias,relu=True)))
