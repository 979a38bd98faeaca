import sys, time, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev='cuda:0'
cfg=S.CONFIGS['cfg3']; G,C=cfg.genes,cfg.cells
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
m=sda.GNN(cfg.dense_dim,cfg.hidden,cfg.n_classes,2,G,activation=F.relu,dropout=0.1).to(dev)
feats=S.synth_features(G+C,cfg.dense_dim,device=dev); y=torch.arange(C,device=dev)%cfg.n_classes
opt=torch.optim.Adam(m.parameters(),lr=1e-3,weight_decay=5e-4)
def step():
    loss=F.cross_entropy(m(g,feats),y,reduction='sum'); opt.zero_grad(); loss.backward(); opt.step(); return loss
for save in (True, False, True, False):
    ops.SAVE_NEIGH_SUM = save
    step(); step(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): loss=step()
    torch.cuda.synchronize(); print(f"plain GNN full-graph training step, SAVE_NEIGH_SUM={save}: {(time.perf_counter()-t)/10*1e3:.2f} ms  loss {float(loss):.1f}")
