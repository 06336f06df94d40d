"""batched two-stage search, int8 against bf16 shadow (10M x 512): python tools/shadow8_batched_bench.py [N] [d]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.index.flat_ip import FlatIPIndex  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = torch.Generator(device="cuda").manual_seed(3)
X = torch.empty(N, d, device="cuda")
for s in range(0, N, 1 << 20):
    e = min(N, s + (1 << 20))
    X[s:e] = torch.nn.functional.normalize(torch.randn(e - s, d, device="cuda", generator=g), dim=1)
Q = torch.nn.functional.normalize(torch.randn(256, d, device="cuda", generator=g), dim=1)
ref = FlatIPIndex(d, shadow=False).adopt(X)
for kind in ("int8", "bf16"):
    idx = FlatIPIndex(d, shadow=kind).adopt(X)
    for k in ((10,) if kind == "int8" and len(sys.argv) > 3 else (10, 100)):
        for nq in (4, 32, 128, 256):
            for _ in range(2):
                D, I = idx.search_device(Q[:nq], k)
            torch.cuda.synchronize()
            n = 4
            t0 = time.perf_counter()
            for _ in range(n):
                D, I = idx.search_device(Q[:nq], k)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            msg = f"{kind} k={k:3d} nq={nq:3d}: {dt * 1e3:8.3f} ms/call {nq / dt:9.0f} q/s  counts {idx.shadow_counts()}"
            if nq == 32:
                Dr, Ir = ref.search_device(Q[:nq], k)
                msg += f"  equal to the f32 scan: ids {bool(torch.equal(I, Ir))} scores {bool(torch.equal(D, Dr))}"
            print(msg, flush=True)
    del idx
    torch.cuda.empty_cache()
