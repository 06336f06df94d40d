"""single-query searches on one rank's share of the index (1.25M x 512), for a per-kernel profile"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.index.flat_ip import FlatIPIndex
N, d, k, n = 1_250_000, 512, 10, 40
g = torch.Generator(device="cuda").manual_seed(3)
X = torch.nn.functional.normalize(torch.randn(N, d, device="cuda", generator=g), dim=1)
Q = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
idx = FlatIPIndex(d, shadow=True).adopt(X)
for i in range(n):
    idx.search_device(Q[i:i + 1], k)
torch.cuda.synchronize()
print(idx.shadow_counts())
