"""Predictor-shaped one-shot path (10k support + 100k test cells): wall vs device time, top kernels."""
import sys, os, time, torch, torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import scdeepsort_amd as sda
from scdeepsort_amd import synthetic as S, tuning
from scdeepsort_amd.graph import CellGeneGraph
from torch.profiler import profile, ProfilerActivity
try:
    torch.backends.cuda.preferred_blas_library("hipblaslt")
except Exception:
    pass
tuning.use_tuned_gemms()
dev = torch.device("cuda:0")
cfg = S.CONFIGS["cfg3"]; G = cfg.genes; n_sup, n_test = 10000, 100000
rp, col, val = S.synth_expression(n_sup + n_test, G, cfg.density, seed=S.REFERENCE_SEED + 29, device=dev)
feats = S.synth_features(G + n_sup + n_test, cfg.dense_dim, seed=31, device=dev)
mask = torch.zeros(n_sup + n_test, dtype=torch.bool, device=dev); mask[:n_sup] = True
seeds = range(G + n_sup, G + n_sup + n_test)
torch.manual_seed(1234)
model = sda.GNN(cfg.dense_dim, cfg.hidden, cfg.n_classes, 2, G, activation=F.relu).to(dev).eval()
def predict():
    gp = CellGeneGraph.from_device_csr(rp, col, val, G, support_mask=mask)
    with torch.no_grad():
        return model(gp, feats, seeds=seeds)
for _ in range(3): predict()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): predict()
torch.cuda.synchronize(); print("wall per call: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    predict(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
