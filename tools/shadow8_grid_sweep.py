"""single-query int8 search time against the collect grid (blocks per CU): python tools/shadow8_grid_sweep.py"""
import os
import sys
import time
from pathlib import Path

import torch

os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd import _lib  # noqa: E402
from wise_amd.index.flat_ip import FlatIPIndex  # noqa: E402

lib = _lib.lib()
N, d, k = 10_000_000, 512, 10
g = torch.Generator(device="cuda").manual_seed(3)
X = torch.empty(N, d, device="cuda")
for s in range(0, N, 1 << 20):
    e = min(N, s + (1 << 20))
    X[s:e] = torch.nn.functional.normalize(torch.randn(e - s, d, device="cuda", generator=g), dim=1)
Q = torch.nn.functional.normalize(torch.randn(64, d, device="cuda", generator=g), dim=1)
idx = FlatIPIndex(d, shadow="int8").adopt(X)
for bpc in (0, 2, 3, 4, 6, 8, 2):
    lib.wise_debug_set_scan(4, bpc)
    for i in range(4):
        idx.search_device(Q[i:i + 1], k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        idx.search_device(Q[i:i + 1], k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"blocks per CU {bpc}: {dt * 1e3:.3f} ms/query  {1 / dt:.0f} q/s", flush=True)
lib.wise_debug_set_scan(4, 0)
