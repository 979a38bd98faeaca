import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch, torch.nn.functional as F
import scdeepsort_amd as sda
DEV = "cuda:0"
def test_full_size_cfg4_eight_virtual_ranks_training_gradients_match_unsharded():
    """BASELINE cfg4 at FULL size ("Same 100k x 20k graph, training loop, cells sharded 8-way"): eight `ShardedWgnn` engines of
    12 500 cells each run the production `dist.sharded_forward` training branch in eight threads of ONE process on one GPU;
    the differentiable [G, H] all-reduce is replaced by an in-process rendezvous that sums the eight partial tensors (forward:
    the sum; backward: autograd adds the eight consumers' gradients - what the all-reduce of dH1_g does), the parameter-gradient
    all-reduce by autograd accumulating into the one shared parameter set.  Loss and every gradient (alpha on all 20 002
    entries) against the unsharded full-batch step of the same model on the whole graph."""
    import threading
    from scdeepsort_amd import dist as D, ops, synthetic as S
    from scdeepsort_amd.sharded import ShardedWgnn
    cfg = S.CONFIGS["cfg3"]
    G, C, N = cfg.genes, cfg.cells, 8
    rp, col, val = S.synth_expression(C, G, cfg.density, device=DEV)
    torch.manual_seed(3)
    m = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, 2, G, activation=F.relu).to(DEV)
    with torch.no_grad():
        m.alpha.uniform_(0.5, 1.5)
    feats = S.synth_features(G + C, cfg.dense_dim, device=DEV)
    labels = (torch.arange(C, device=DEV) * 2654435761 % cfg.n_classes).long()
    g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
    ref_loss = sda.cross_entropy_sum(m(g, feats), labels)
    ref_loss.backward()
    want = {k: q.grad.detach().clone() for k, q in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    del g
    stats = ShardedWgnn.gene_stats(col, val, G)
    engines = []
    for r in range(N):
        lo, hi = D.shard_range(C, r, N)
        b, e = int(rp[lo]), int(rp[hi])
        engines.append((lo, hi, ShardedWgnn.build(m, (rp[lo:hi + 1] - rp[lo]).clone(), col[b:e].clone(), val[b:e].clone(), G,
                                                  global_stats=stats)))
    bar, parts, box, tl = threading.Barrier(N), [None] * N, {}, threading.local()

    calls = []
    def rendezvous_sum(x):
        calls.append(1)                                  # stands in for dist.all_reduce_sum (differentiable SUM all-reduce)
        parts[tl.rank] = x
        if bar.wait() == 0:
            tot = parts[0]
            for q in parts[1:]:
                tot = tot + q
            box["total"] = tot
        bar.wait()
        return box["total"]

    losses, errors = [None] * N, []

    def rank_main(r):
        try:
            tl.rank = r
            lo, hi, eng = engines[r]
            with torch.cuda.device(DEV), torch.enable_grad():
                logits = D.sharded_forward(eng._weights(), None, feats[:G], feats[G + lo:G + hi], eng._ops(), 2, gather_logits=False,
                                           linear=ops.linear)
                losses[r] = sda.cross_entropy_sum(logits, labels[lo:hi])
        except BaseException as ex:                          # a failing rank must not leave the others at the barrier
            errors.append(ex)
            bar.abort()

    saved, D.all_reduce_sum = D.all_reduce_sum, rendezvous_sum
    try:
        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(N)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        D.all_reduce_sum = saved
    assert not errors, errors
    loss = losses[0]
    for q in losses[1:]:
        loss = loss + q
    loss.backward()
    lv, rv = float(loss.detach()), float(ref_loss.detach())
    print("loss", lv, rv, "rendezvous calls", len(calls), "popularity", os.environ.get("WGNN_SYNTH_POPULARITY"))
    assert abs(lv - rv) < 1e-5 * abs(rv), (lv, rv)
    for k, q in m.named_parameters():
        scale = float(want[k].abs().max())
        d = (q.grad - want[k]).abs().reshape(-1)
        err = float(d.max())
        print(k, "err", err, "scale", scale, "n_bad", int((d > 1e-3 * scale + 1e-6).sum()))
        if k == "alpha":
            bad = torch.nonzero(d > 1e-3 * scale + 1e-6).reshape(-1)
            print("bad idx", bad[:40].tolist())
            deg = torch.bincount(col.long(), minlength=G)
            print("their degrees", deg[bad[:40].clamp(max=G-1)].tolist())
            print("got", q.grad.reshape(-1)[bad[:10]].tolist(), "want", want[k].reshape(-1)[bad[:10]].tolist())
            srt = torch.sort(deg, descending=True)
            print("top degrees", srt.values[:10].tolist(), srt.indices[:10].tolist())



test_full_size_cfg4_eight_virtual_ranks_training_gradients_match_unsharded()
