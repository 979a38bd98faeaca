import sys, torch
sys.path.insert(0, '/root/repo')
from scdeepsort_amd import ops
dev = 'cuda:0'
x = torch.randn(100000, 400, device=dev); w = torch.randn(256, 400, device=dev)
for _ in range(5): ops.linear_fwd(x, w)
torch.cuda.synchronize()
