import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
from scdeepsort_amd.graph import build_tile_plan
dev='cuda:0'
G,C,H=300,700,256
rp,col,val=S.synth_expression(C,G,0.2,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.ones(G+2,device=dev)
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
ops.TILED_MIN_WORK=None
ref=sda.agg_fwd(g.cg,alpha,sda.SRC_IS_GENE,G+1,hg,hc)
tp=build_tile_plan(g.cg,3,1)
o=ops.agg_fwd_tiled(g.cg,tp,alpha,sda.SRC_IS_GENE,G+1,hg,hc)
d=(o-ref).abs()
print('max err',d.max().item(),'rows wrong',(d.max(1).values>1e-4).sum().item(),'of',C)
print('col-block errs',[round(d[:,i*64:(i+1)*64].max().item(),4) for i in range(4)])
r=int(d.max(1).values.argmax()); print('worst row',r,'nnz',int(rp[r+1]-rp[r]),'ratio o/ref',(o[r,:4]/ref[r,:4]).tolist())
# per-row: fraction of correct sum recovered
inv=g.cg.inv_deg
raw_ref=ref/inv[:,None]-hc; raw_o=o/inv[:,None]-hc
ratio=(raw_o*raw_ref).sum(1)/(raw_ref*raw_ref).sum(1)
print('projection ratio quantiles',np.quantile(ratio.cpu().numpy(),[0,0.1,0.5,0.9,1]).round(3))
