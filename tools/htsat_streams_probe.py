"""Does the gain of two HTSAT batches in flight depend on how many torch streams the process took before the engine took its two?
(torch hands out pooled streams round-robin; HIP maps streams onto a few hardware queues):  python tools/htsat_streams_probe.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict  # noqa: E402

wav = 0.1 * torch.randn(128, 480000, generator=torch.Generator(device="cuda").manual_seed(4), device="cuda")
sd = random_htsat_state_dict(0)
hold = {}
keep = []
for before in (0, 1, 2, 3, 4, 5, 6, 7):
    while len(keep) < before:
        keep.append(torch.cuda.Stream())
    eng = HtsatEngine(sd, max_batch=128, max_samples=480000)
    for i in range(3):
        hold["p"] = eng.forward_pipelined(wav)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(12):
        hold["p"] = eng.forward_pipelined(wav)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 12 * 1e3
    print(f"{before} streams taken before (this engine's are new each time): two in flight {dt:.3f} ms", flush=True)
    del eng
