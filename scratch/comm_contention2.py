"""Round 4, second part of scratch/comm_contention.py: both passes of a rank's shard at N = 2 / 4 / 8, heuristic geometry for
a budget of 256 CUs (today) vs 224 (32 left to a communicator kernel), timed alone and next to 16 / 32 held CUs."""
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
for world in (8, 4, 2):
    cells = cfg.cells // world
    rp, col, val = S.synth_expression(cells, G, device=dev)
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
    alpha = torch.rand(G + 2, device=dev) + 0.5
    hg = S.synth_features(G, D, device=dev); hc = S.synth_features(cells, D, seed=3, device=dev)
    kb = ops.tiled_block_rows(D)
    for pname, csr, mode, si, hs, hself in (("cells<-genes", g.cg, sda.SRC_IS_GENE, G + 1, hg, hc), ("genes<-cells", g.gc, sda.DST_IS_GENE, G, hc, hg)):
        rec = {}
        for budget in (256, 240, 224):
            tp = GR.build_tile_plan(csr, None, None, n_cus=budget, block_rows=kb, n_loaders=GR.TILE_LOADER_WAVES)
            def run(): return ops.agg_fwd_tiled(csr, tp, alpha, mode, si, hs, hself)
            run(); torch.cuda.synchronize()
            row = {"tiles": f"{tp.n_row_tiles}x{tp.n_col_splits} L{tp.n_loaders}"}
            for k in (0, 16, 32):
                ts = []
                for _ in range(7):
                    cur = torch.cuda.current_stream(dev)
                    if k:
                        side.wait_stream(cur)
                        hog.hog_launch(k, 200_000, side.cuda_stream)
                        torch.cuda._sleep(20_000)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(); e1.record(); torch.cuda.synchronize()
                    ts.append(round(e0.elapsed_time(e1) * 1e3, 1))
                row[f"{k} held"] = sorted(ts)[len(ts) // 2]
            rec[f"budget {budget}"] = row
            print(world, pname, budget, row, flush=True)
        out[f"N={world} ({cells} cells) {pname}, us incl. agg_finalize"] = rec
    del g
json.dump(out, open('/root/repo/gpurun_out/comm_contention2.json', 'w'), indent=1)
