"""Round 4: the full-batch cfg3 training step (BASELINE cfg4 at N = 1: forward + CE-sum + backward through K2t / K3t + Adam,
dropout 0.1) - 3 warm-up + N timed steps; run under rocprofv3 --kernel-trace --stats for the per-kernel table."""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S
dev = 'cuda:0'
cfg = S.CONFIGS['cfg3']; G, C = cfg.genes, cfg.cells
rp, col, val = S.synth_expression(C, G, device=dev)
g = sda.CellGeneGraph.from_device_csr(rp, col, val, G)
torch.manual_seed(1)
m = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, 2, G, activation=F.relu, dropout=float(os.environ.get('DROPOUT', '0.1'))).to(dev)
feats = S.synth_features(G + C, cfg.dense_dim, device=dev); y = torch.arange(C, device=dev) % cfg.n_classes
from scdeepsort_amd import ops
FUSED = os.environ.get('FUSED', '1') == '1'           # round-4 glue: fused CE-sum, fused backward glue, fused Adam
ops.FUSED_BWD_GLUE = FUSED
opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-4, fused=FUSED)
def step():
    logits = m(g, feats)
    loss = sda.cross_entropy_sum(logits, y) if FUSED else F.cross_entropy(logits, y, reduction='sum')
    opt.zero_grad(); loss.backward(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
n = int(os.environ.get('STEPS', '10'))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): loss = step()
e1.record(); torch.cuda.synchronize()
print(f"FUSED={int(FUSED)} full-batch cfg3 training step: {e0.elapsed_time(e1) / n:.3f} ms  (loss {float(loss):.1f})")
