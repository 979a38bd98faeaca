import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
from scdeepsort_amd.graph import build_tile_plan
dev='cuda:0'
G,C,H=20000,100000,256
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
def timeit(f,n=10):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
ref=None
for rt,cs in [(80,16),(96,8),(86,6),(85,3),(128,8),(80,8),(80,12),(103,5),(171,3),(256,2),(79,13)]:
    tp=build_tile_plan(g.gc,rt,cs,block_rows=78)
    out=ops.agg_fwd_tiled(g.gc,tp,alpha,sda.DST_IS_GENE,G,hc,hg)
    if ref is None: ref=out
    t=timeit(lambda: ops.agg_fwd_tiled(g.gc,tp,alpha,sda.DST_IS_GENE,G,hc,hg))
    print(f"genes {rt:4d} x {cs:2d} = {tp.items.shape[0]:5d} tiles  {t:.3f} ms  err {float((out-ref).abs().max()):.1e}", flush=True)
ref=None
for rt,cs in [(512,1),(768,1),(1024,1),(400,1),(256,2),(512,2)]:
    tp=build_tile_plan(g.cg,rt,cs,block_rows=78)
    out=ops.agg_fwd_tiled(g.cg,tp,alpha,sda.SRC_IS_GENE,G+1,hg,hc)
    if ref is None: ref=out
    t=timeit(lambda: ops.agg_fwd_tiled(g.cg,tp,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    print(f"cells {rt:4d} x {cs:2d} = {tp.items.shape[0]:5d} tiles  {t:.3f} ms  err {float((out-ref).abs().max()):.1e}", flush=True)
