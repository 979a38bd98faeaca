import sys, time, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S
from scdeepsort_amd.graphed import GraphedForward
dev='cuda:0'
for name in ('tiny','cfg2'):
    cfg=S.CONFIGS[name]; G,C=cfg.genes,cfg.cells
    rp,col,val=S.synth_expression(C,G,device=dev)
    g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
    m=sda.GNN(cfg.dense_dim,cfg.hidden,cfg.n_classes,2,G,activation=F.relu).to(dev).eval()
    x=S.synth_features(G+C,cfg.dense_dim,device=dev)
    def timeit(f,n=200):
        f(); torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
    with torch.no_grad():
        te=timeit(lambda: m(g,x))
    gf=GraphedForward(m,g,x)
    tg=timeit(lambda: gf())
    print(f"{name}: eager {te:.1f} us   hipGraph replay {tg:.1f} us")
