"""batched searches (nq = 256) for a per-kernel profile: python tools/search_batched_kernels.py [k] [n]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.index.flat_ip import FlatIPIndex
k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N, d = 10_000_000, 512
g = torch.Generator(device="cuda").manual_seed(3)
X = torch.empty(N, d, device="cuda")
for s in range(0, N, 1 << 20):
    e = min(N, s + (1 << 20))
    X[s:e] = torch.nn.functional.normalize(torch.randn(e - s, d, device="cuda", generator=g), dim=1)
Q = torch.nn.functional.normalize(torch.randn(256, d, device="cuda", generator=g), dim=1)
idx = FlatIPIndex(d, shadow=True).adopt(X)
for i in range(n):
    idx.search_device(Q, k)
torch.cuda.synchronize()
print(idx.shadow_counts())
