import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S
dev='cuda:0'
G,C,H=20000,100000,256
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
A=torch.sparse_csr_tensor(g.cg.rowptr.long(), g.cg.col.long(), g.cg.val, size=(C,G))
B=torch.sparse_csr_tensor(g.gc.rowptr.long(), g.gc.col.long(), g.gc.val, size=(G,C))
def timeit(f,n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print('torch.sparse.mm (hipSPARSE) cells<-genes', timeit(lambda: torch.sparse.mm(A,hg)), 'ms; genes<-cells', timeit(lambda: torch.sparse.mm(B,hc)), 'ms')
