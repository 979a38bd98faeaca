import sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S
dev='cuda:0'
G,C=20000,100000
rp,col,val=S.synth_expression(C,G,device=dev)
g=sda.CellGeneGraph.from_device_csr(rp,col,val,G)
for name,csr in (('cells',g.cg),('genes',g.gc)):
    tp=csr.tile_plan(78)
    n_tiles=tp.items.shape[0]
    seg=tp.seg_ptr.long()
    cnt=(seg[1:]-seg[:-1]).reshape(n_tiles,tp.nblk_max,16).float()
    blk_tot=cnt.sum(2)
    live=blk_tot>0
    mx=cnt.max(2).values[live]; mean=(blk_tot/16)[live]
    print(f"{name}: tiles {n_tiles} blocks/tile {tp.nblk_max}; per block: mean wave load {mean.mean():.1f}, mean of max {mx.mean():.1f}, ratio {(mx.sum()/mean.sum()):.3f}; "
          f"tile totals: min {blk_tot.sum(1).min():.0f} max {blk_tot.sum(1).max():.0f}; wave totals max/mean {(cnt.sum(1).max(1).values/cnt.sum(1).mean(1)).mean():.3f}; chunks>64: {(cnt>64).float().mean():.4f}")
