"""a few batched calls (nq = 128 and 64) of the two-stage search over 10M x 512 — the command a per-kernel profile is taken of"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.index.flat_ip import FlatIPIndex
N, d = 10_000_000, 512
X = torch.empty(N, d, device="cuda")
g = torch.Generator(device="cuda").manual_seed(3)
for s in range(0, N, 1_000_000):
    X[s:s + 1_000_000] = torch.nn.functional.normalize(torch.randn(1_000_000, d, device="cuda", generator=g), dim=1)
Q = torch.nn.functional.normalize(torch.randn(256, d, device="cuda", generator=g), dim=1)
idx = FlatIPIndex(d, shadow=True).adopt(X)
for nq in (128, 64):
    for _ in range(6): idx.search_device(Q[:nq], 10)
torch.cuda.synchronize()
