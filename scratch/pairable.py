"""How many entries of a (wave, LDS block) chunk share their source row with another entry of the same chunk (cfg3 plans)?"""
import sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, graph as GR
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C = cfg.genes, cfg.cells
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
for name, csr in (("cells<-genes", g.cg), ("genes<-cells", g.gc)):
    for L in (0, 1):
        tp = GR.build_tile_plan(csr, None, None, block_rows=78, n_loaders=L)
        seg = tp.seg_ptr.long()
        per = seg[1:] - seg[:-1]
        sid = torch.repeat_interleave(torch.arange(per.shape[0], device=dev), per)
        src = tp.entries[:, 0].long() & 0xFF
        key = sid * 256 + src                                   # (segment, source row)
        uniq, cnt = torch.unique(key, return_counts=True)
        n = key.shape[0]
        pairs = (cnt // 2).sum().item()                         # disjoint pairs that can share one LDS read
        multi = cnt[cnt >= 2].sum().item()
        hist = torch.bincount(cnt.clamp(max=8))[1:].tolist()
        print(f"{name} L={tp.n_loaders}: entries {n/1e6:.1f} M, in shared pairs {2*pairs/n:.3f}, entries whose row occurs >= 2x {multi/n:.3f}, "
              f"distinct rows / entries {uniq.shape[0]/n:.3f}, multiplicity histogram (1..8+) {hist}", flush=True)
