"""Where the one-shot path spends its time: wall + kernel-time tables of the graph build and of the tile-plan builds (cfg3)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
from torch.profiler import profile, ProfilerActivity
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']
rp, col, val = S.synth_expression(cfg.cells, cfg.genes, device=dev)
def wall(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3, r
for rep in range(3):
    tg, g = wall(lambda: sda.CellGeneGraph.from_device_csr(rp, col, val, cfg.genes))
    tc, _ = wall(lambda: GR.build_tile_plan(g.cg, None, None, block_rows=78, n_loaders=1))
    tgn, _ = wall(lambda: GR.build_tile_plan(g.gc, None, None, block_rows=78, n_loaders=1))
    print(f'rep {rep}: graph build {tg:.1f} ms, plan cells side {tc:.1f} ms, plan gene side {tgn:.1f} ms', flush=True)
for name, fn in (('graph build', lambda: sda.CellGeneGraph.from_device_csr(rp, col, val, cfg.genes)),
                 ('plan cells<-genes', lambda: GR.build_tile_plan(g.cg, None, None, block_rows=78, n_loaders=1)),
                 ('plan genes<-cells', lambda: GR.build_tile_plan(g.gc, None, None, block_rows=78, n_loaders=1))):
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        fn(); torch.cuda.synchronize()
    print('====', name)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=56))
