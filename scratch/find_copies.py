"""Which torch ops launch the two `direct_copy` kernels per cfg3 forward (rocprofv3: 2 x 24.5 us per step)?"""
import sys, torch
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S
from scdeepsort_amd.sharded import ShardedWgnn
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
cfg = S.CONFIGS['cfg3']; G = cfg.genes
rp, col, val = S.synth_expression(cfg.cells, G, cfg.density, device=dev)
fg = S.synth_features(G, cfg.dense_dim, seed=7, device=dev); fc = S.synth_features(cfg.cells, cfg.dense_dim, seed=100, device=dev)
import torch.nn.functional as F
torch.manual_seed(0)
model = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, cfg.n_layers, G, activation=F.relu).to(dev).eval()
eng = ShardedWgnn.build(model, rp, col, val, G)
with torch.no_grad():
    for _ in range(3): eng.forward(fg, fc)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        eng.forward(fg, fc); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
for e in prof.events():
    if e.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::to", "aten::_to_copy") and e.device_time_total > 5:
        print(e.name, e.input_shapes, e.device_time_total, [str(s) for s in (e.stack or [])[:6]])
