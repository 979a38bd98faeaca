"""Kernel-time table of one tile-plan build (cfg3 cells side, then gene side)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
from torch.profiler import profile, ProfilerActivity
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']
rp, col, val = S.synth_expression(cfg.cells, cfg.genes, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, cfg.genes)
GR.build_tile_plan(g.cg, None, None, block_rows=78, n_loaders=1); torch.cuda.synchronize()
for name, csr in (('cells<-genes', g.cg), ('genes<-cells', g.gc)):
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        GR.build_tile_plan(csr, None, None, block_rows=78, n_loaders=1); torch.cuda.synchronize()
    print('====', name)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
