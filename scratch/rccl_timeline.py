"""Round 4: what RCCL launches for the data-path collectives of ONE rank's forward (N = 8-size shard, a ONE-rank `nccl` group with
dist.FORCE_COLLECTIVES - the only communicator the 1-GPU lease allows), to be run under `rocprofv3 --kernel-trace`; the companion
scratch/rccl_timeline_summary.py turns the trace into the ordered kernel list of one forward (name, grid, start, end)."""
import os, sys, json, socket, torch, torch.distributed as dist, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, dist as D, tuning
from scdeepsort_amd.sharded import ShardedWgnn
tuning.use_tuned_gemms()
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
D.reserve_comm_cus()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
D.FORCE_COLLECTIVES = True
cfg = S.CONFIGS['cfg3']; G = cfg.genes; N = 8
rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, seed=S.REFERENCE_SEED, device=dev)
lo, hi = D.shard_range(cfg.cells, 0, N)
b, e = int(rp[lo]), int(rp[hi])
gdeg, gsum = ShardedWgnn.gene_stats(col, val, G)
feats_g = S.synth_features(G, cfg.dense_dim, seed=7, device=dev)
fc = S.synth_features(cfg.cells, cfg.dense_dim, seed=100, device=dev)[lo:hi].contiguous()
torch.manual_seed(1234)
model = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, 2, G, activation=F.relu).to(dev).eval()
eng = ShardedWgnn.build(model, (rp[lo:hi + 1] - rp[lo]).clone(), col[b:e].clone(), val[b:e].clone(), G, global_stats=(gdeg, gsum))
with torch.no_grad():
    for _ in range(5):
        out = eng.forward(feats_g, fc, async_gather=True)
    eng.wait_gather(); torch.cuda.synchronize()
    torch.cuda._sleep(2_000_000)                 # a visible gap in the trace: the forwards after it are the ones summarised
    for _ in range(3):
        out = eng.forward(feats_g, fc, async_gather=True)
    eng.wait_gather(); torch.cuda.synchronize()
print(json.dumps({"NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"), "logits": list(out.shape)}))
dist.destroy_process_group()
