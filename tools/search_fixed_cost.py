"""per-kernel times of the single-query two-stage search (run under rocprofv3 --kernel-trace --stats):
python tools/search_fixed_cost.py [rows] [k] [queries]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.index.flat_ip import FlatIPIndex

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nqs = int(sys.argv[3]) if len(sys.argv) > 3 else 200
d = 512
g = torch.Generator(device="cuda").manual_seed(5)
X = torch.empty(N, d, device="cuda")
for s in range(0, N, 1_000_000):
    e = min(N, s + 1_000_000)
    X[s:e] = torch.nn.functional.normalize(torch.randn(e - s, d, device="cuda", generator=g), dim=1)
Q = torch.nn.functional.normalize(torch.randn(nqs, d, device="cuda", generator=g), dim=1)
idx = FlatIPIndex(d).adopt(X)
for i in range(5):
    idx.search_device(Q[i:i + 1], k)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(nqs):
    idx.search_device(Q[i:i + 1], k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / nqs
print(f"N={N} k={k}: {dt * 1e3:.4f} ms/query, {1 / dt:.1f} q/s, shadow counts {idx.shadow_counts()}")
