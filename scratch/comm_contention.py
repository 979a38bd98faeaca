"""Round 4: what a communicator kernel next to the cells<-genes pass costs.  Every tile workgroup needs a whole CU; the shard
geometries of the strong-scaling job use (nearly) all 256 CUs in ONE round (N = 8: 64 x 4 = 256 tiles), so k CUs held by another
kernel (RCCL: one 256-thread workgroup per channel) push k tiles into a second round.  A spin kernel stands in for the
communicator (scratch/hog.hip); the pass is timed alone and next to k = 8 / 16 / 32 held CUs, for the default geometry and for
geometries that leave CUs free."""
import sys, json, ctypes, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops, graph as GR
dev = 'cuda:0'
hog = ctypes.CDLL('/root/repo/scratch/variants/libhog.so')
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
cfg = S.CONFIGS["cfg3"]; G = cfg.genes; D = 256
out = {}
side = torch.cuda.Stream(device=dev)
for world in (8, 4):
    cells = cfg.cells // world
    rp, col, val = S.synth_expression(cells, G, device=dev)
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
    alpha = torch.rand(G + 2, device=dev) + 0.5
    hg = S.synth_features(G, D, device=dev); hc = S.synth_features(cells, D, seed=3, device=dev)
    kb = ops.tiled_block_rows(D)
    default = g.cg.tile_plan(kb)
    plans = {f"default {default.n_row_tiles}x{default.n_col_splits}": default}
    for rt in ((60, 56, 48) if world == 8 else (112, 104)):
        tp = GR.build_tile_plan(g.cg, rt, default.n_col_splits, block_rows=kb, n_loaders=GR.TILE_LOADER_WAVES)
        plans[f"{tp.n_row_tiles}x{tp.n_col_splits}"] = tp
    rec = {}
    for name, tp in plans.items():
        def run(): return ops.agg_fwd_tiled(g.cg, tp, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
        run(); torch.cuda.synchronize()
        row = {}
        for k in (0, 8, 16, 32):
            ts = []
            for _ in range(6):
                cur = torch.cuda.current_stream(dev)
                if k:
                    side.wait_stream(cur)
                    hog.hog_launch(k, 100_000, side.cuda_stream)          # ~1 ms: outlasts the pass
                    torch.cuda._sleep(20_000)                              # let the spin kernel take its CUs first
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); torch.cuda.synchronize()
                ts.append(round(e0.elapsed_time(e1) * 1e3, 1))
            row[f"{k} CUs held"] = sorted(ts)[len(ts) // 2]
        rec[name] = row
        print(world, name, row, flush=True)
    out[f"N={world} shard ({cells} cells), cells<-genes pass incl. agg_finalize, us"] = rec
    del g
json.dump(out, open('/root/repo/gpurun_out/comm_contention.json', 'w'), indent=1)
