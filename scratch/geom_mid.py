import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
from scdeepsort_amd.graph import build_tile_plan
dev='cuda:0'
H=256
def timeit(f,n=10):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for C,G in ((10000,10000),(20000,15000),(50000,20000)):
    rp,col,val=S.synth_expression(C,G,device=dev)
    g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
    alpha=torch.rand(G+2,device=dev)+0.5
    hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
    for name,csr,mode,si,hs,hself in (('cells',g.cg,sda.SRC_IS_GENE,G+1,hg,hc),('genes',g.gc,sda.DST_IS_GENE,G,hc,hg)):
        out=[]
        R=csr.n_rows
        base=-(-R//250)
        for rt,cs in [(None,None),(base,4),(base,8),(base,32),(2*base,4),(2*base,8),(4*base,2),(4*base,4),(256,1),(256,2),(512,1),(256,4)]:
            try:
                tp=build_tile_plan(csr,rt,cs,block_rows=78)
            except Exception as e:
                continue
            t=timeit(lambda: ops.agg_fwd_tiled(csr,tp,alpha,mode,si,hs,hself))
            out.append(f"{tp.n_row_tiles}x{tp.n_col_splits}:{t:.3f}")
        print(f"C={C} G={G} {name}: "+'  '.join(out), flush=True)
