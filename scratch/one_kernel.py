import os, sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev='cuda:0'
cfg=S.CONFIGS['cfg3']; G,C,H=cfg.genes,cfg.cells,cfg.hidden
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
kb=ops.tiled_block_rows(H)
tpc=g.cg.tile_plan(kb); tpg=g.gc.tile_plan(kb)
ops.DEBUG_FLAGS=int(os.environ.get('WGNN_DBG','0'))
for _ in range(3):
    ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc)
torch.cuda.synchronize()
if not os.environ.get('WGNN_ONE_PASS'):
    for _ in range(3):
        ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg)
    torch.cuda.synchronize()
