"""HTSAT clips/s against the number of batches in flight: python tools/htsat_inflight_sweep.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict  # noqa: E402

B, N = 128, 480000
w = 0.1 * torch.randn(B, N, device="cuda")
sd = random_htsat_state_dict(0)
for nf in (2, 3, 4, 2, 3):
    eng = HtsatEngine(sd, max_batch=B, max_samples=N)
    eng.batches_in_flight = nf
    hs = [eng.forward_pipelined(w) for _ in range(6)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    hs = [eng.forward_pipelined(w) for _ in range(n)]
    hs[-1].result()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{nf} batches in flight: {dt * 1e3:.3f} ms/step  {B / dt:.0f} clips/s", flush=True)
    del eng
