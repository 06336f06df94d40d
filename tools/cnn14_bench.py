"""clips/s of the Cnn14 forward (MS-CLAP 2022 audio encoder): python tools/cnn14_bench.py [batch] [samples] [--serial-only]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.cnn14 import Cnn14Engine, flops_per_clip, random_cnn14_state_dict  # noqa: E402

import os
if any(a.startswith("--ablate=") for a in sys.argv):
    os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")
SERIAL_ONLY = "--serial-only" in sys.argv        # (profiling: per-kernel durations without a second batch beside them)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if len(args) > 0 else 64
N = int(args[1]) if len(args) > 1 else 480000

eng = Cnn14Engine(random_cnn14_state_dict(0), max_batch=B, max_samples=N)
w = 0.1 * torch.randn(B, N, device="cuda")
fl = flops_per_clip(N)
for _ in range(2):
    o = eng.forward(w)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        o = eng.forward(w)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"Cnn14 B={B} N={N}: {dt*1e3:.3f} ms/step  {B/dt:.1f} clips/s  {B/dt*fl/1e12:.1f} TFLOP/s ({fl/1e9:.1f} GFLOP per clip)", flush=True)
ref = o.clone()
ABL = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--ablate=")]   # timing-only instantiations of block 1 (debug library)
for ab in ABL:
    from wise_amd import _lib
    _lib.lib().wise_debug_set_cnn14(ab)
    for _ in range(2):
        eng.forward(w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.forward(w)
    torch.cuda.synchronize()
    print(f"block-1 ablation {ab}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/step", flush=True)
if ABL:
    sys.exit(0)
if SERIAL_ONLY:
    sys.exit(0)
for rep in range(2):
    hs = [eng.forward_pipelined(w) for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hs = [eng.forward_pipelined(w) for _ in range(6)]
    o2 = hs[-1].result()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    print(f"Cnn14 B={B} N={N} two batches in flight: {dt*1e3:.3f} ms/step  {B/dt:.1f} clips/s  {B/dt*fl/1e12:.1f} TFLOP/s  same={torch.equal(ref, o2)}", flush=True)
