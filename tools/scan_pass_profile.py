"""one batched (64-query) two-stage search pass over 10M x 512 under rocprofv3 --kernel-trace --stats: where the pass goes"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.index.flat_ip import FlatIPIndex
N, d = 10_000_000, 512
g = torch.Generator(device="cuda").manual_seed(5)
X = torch.empty(N, d, device="cuda")
for s in range(0, N, 1_000_000):
    X[s:s + 1_000_000] = torch.nn.functional.normalize(torch.randn(1_000_000, d, device="cuda", generator=g), dim=1)
Q = torch.nn.functional.normalize(torch.randn(256, d, device="cuda", generator=g), dim=1)
idx = FlatIPIndex(d, shadow=True).adopt(X)
for nq in (64, 1):
    for _ in range(3): idx.search_device(Q[:nq], 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): idx.search_device(Q[:nq], 10)
    torch.cuda.synchronize(); print(nq, "queries:", (time.perf_counter() - t0) / 20 * 1e3, "ms per call", idx.shadow_counts(), flush=True)
