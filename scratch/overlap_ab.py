"""Round 4: do the two independent layer-1 tile passes (cells<-genes, genes<-cells) gain from running on two streams?
Each launch fills the chip with one workgroup per CU (the kernel takes a CU's whole register file), so a second kernel can
only start on CUs the first has released: what two streams can recover is the TAIL of the first pass (CUs idle while the
slowest tiles finish) and the launch gap.  Same-run A/B, interleaved repetitions; cfg3 and the N = 8 shard of cfg3."""
import sys, json, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, ops
dev = 'cuda:0'
cfg = S.CONFIGS["cfg3"]; G, C = cfg.genes, cfg.cells
out = {}
for cells in (C, C // 8):
    rp, col, val = S.synth_expression(cells, G, device=dev)
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
    alpha = torch.rand(G + 2, device=dev) + 0.5
    D = 256
    hg = S.synth_features(G, D, device=dev); hc = S.synth_features(cells, D, seed=3, device=dev)
    side = torch.cuda.Stream(device=dev)
    def seq():
        a = ops.agg_fwd(g.cg, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
        b = ops.agg_fwd(g.gc, alpha, sda.DST_IS_GENE, G, hc, hg)
        return a, b
    def par():
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            b = ops.agg_fwd(g.gc, alpha, sda.DST_IS_GENE, G, hc, hg)
        a = ops.agg_fwd(g.cg, alpha, sda.SRC_IS_GENE, G + 1, hg, hc)
        cur.wait_stream(side)
        return a, b
    def timeit(f, n=20):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / n * 1e3, 1)
    a0, b0 = seq(); a1, b1 = par(); torch.cuda.synchronize()
    rec = {"equal": bool(torch.equal(a0, a1) and torch.equal(b0, b1)), "seq_us": [], "two_streams_us": []}
    for _ in range(4):
        rec["seq_us"].append(timeit(seq)); rec["two_streams_us"].append(timeit(par))
    rec["cells_only_us"] = timeit(lambda: ops.agg_fwd(g.cg, alpha, sda.SRC_IS_GENE, G + 1, hg, hc))
    rec["genes_only_us"] = timeit(lambda: ops.agg_fwd(g.gc, alpha, sda.DST_IS_GENE, G, hc, hg))
    out[f"{cells} cells"] = rec
    print(cells, rec, flush=True)
    del g
json.dump(out, open('/root/repo/gpurun_out/overlap_ab.json', 'w'), indent=1)
