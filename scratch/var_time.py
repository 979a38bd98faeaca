import sys, time, torch
sys.path.insert(0, '/root/repo')
from pathlib import Path
from scdeepsort_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = Path(sys.argv[1]).resolve()
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev='cuda:0'
cfg=S.CONFIGS['cfg3']; G,C=cfg.genes,cfg.cells; H=256
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
alpha=torch.rand(G+2,device=dev)+0.5
hg=S.synth_features(G,H,device=dev); hc=S.synth_features(C,H,seed=3,device=dev)
def timeit(f,n=20):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
kb=int(sys.argv[2]) if len(sys.argv) > 2 else 80
tpc=g.cg.tile_plan(kb); tpg=g.gc.tile_plan(kb)
out=[]
ops.DEBUG_FLAGS=0
ref=ops.agg_fwd(g.cg,alpha,sda.SRC_IS_GENE,G+1,hg,hc) if hasattr(ops,'agg_fwd') else None
import scdeepsort_amd.ops as _o
_tm=_o.TILED_MIN_WORK; _o.TILED_MIN_WORK=1e30
ref=ops.agg_fwd(g.cg,alpha,sda.SRC_IS_GENE,G+1,hg,hc); _o.TILED_MIN_WORK=_tm
got=ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc)
err=float((ref-got).abs().max())
for nm,fl in [('full',0),('nofill',1<<16),('nofill+nobar',(1<<16)|(1<<18)),('nobar',1<<18)]:
    ops.DEBUG_FLAGS=fl
    tc=timeit(lambda: ops.agg_fwd_tiled(g.cg,tpc,alpha,sda.SRC_IS_GENE,G+1,hg,hc))
    tg=timeit(lambda: ops.agg_fwd_tiled(g.gc,tpg,alpha,sda.DST_IS_GENE,G,hc,hg))
    out.append(f'{nm} {tc:.3f}/{tg:.3f}')
print(Path(sys.argv[1]).stem if len(sys.argv)>1 else 'default', f'err {err:.1e}', ' | '.join(out), flush=True)
