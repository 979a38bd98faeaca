import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev='cuda:0'
G,C,H=20000,100000,256
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
def timeit(f,n=20):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
B=lambda *bits: sum(1<<b for b in bits)
for kb in (78, 64, 48, 32):
    tpc=g.cg.tile_plan(kb)
    out=[]
    for nm,fl in [('full',0),('nocompute',B(17)),('nocompute+nobar',B(17,18)),('nofill',B(16))]:
        ops.DEBUG_FLAGS=fl
        out.append(f"{nm} {timeit(lambda: ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc)):.3f}")
    print(f"kb {kb}: "+' | '.join(out), flush=True)
