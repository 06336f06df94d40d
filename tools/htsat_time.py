"""HTSAT forward time (128 clips x 10 s), one batch at a time and two in flight"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict
eng = HtsatEngine(random_htsat_state_dict(0), max_batch=128, max_samples=480000)
w = 0.1 * torch.randn(128, 480000, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
for _ in range(3): eng.forward(w)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): eng.forward(w)
torch.cuda.synchronize(); a = (time.perf_counter() - t0) / 10
for _ in range(2): eng.forward_pipelined(w)
torch.cuda.synchronize(); t0 = time.perf_counter()
hs = [eng.forward_pipelined(w) for _ in range(10)]
hs[-1].result(); torch.cuda.synchronize(); b = (time.perf_counter() - t0) / 10
print(f"HTSAT 128 clips: {a * 1e3:.3f} ms one at a time, {b * 1e3:.3f} ms two in flight ({128 / b:.0f} clips/s)", flush=True)
