import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev='cuda:0'
H=256
def timeit(f,n=10):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for C,G in ((5000,10000),(10000,10000),(20000,15000),(30000,20000),(50000,20000)):
    rp,col,val=S.synth_expression(C,G,device=dev)
    g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
    alpha=torch.rand(G+2,device=dev)+0.5
    hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
    res=[]
    for thr in (None,1):
        ops.TILED_MIN_WORK=thr
        tc=timeit(lambda: ops.agg_fwd(g.cg,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
        tg=timeit(lambda: ops.agg_fwd(g.gc,alpha,sda.DST_IS_GENE,G,hc,hg))
        res.append((tc,tg))
    print(f"C={C} G={G} nnz={g.cg.nnz/1e6:.1f}M nnz*D={g.cg.nnz*H/1e9:.2f}e9 | K1 cells {res[0][0]:.3f} genes {res[0][1]:.3f} | tiled cells {res[1][0]:.3f} genes {res[1][1]:.3f}", flush=True)
