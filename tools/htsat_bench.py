"""clips/s of the HTSAT forward: python tools/htsat_bench.py [batch] [samples]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 480000
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib  # noqa: E402

eng = HtsatEngine(random_htsat_state_dict(0), max_batch=B, max_samples=N)
w = 0.1 * torch.randn(B, N, device="cuda")
lib = _lib.lib()
for flags in (0, 32, 0, 32, 16):   # bit 0: no LayerNorm fusion at all, bit 1: no fused MLP, bit 2: no fused attention half (stage 1)
    lib.wise_debug_set_htsat(flags)
    for _ in range(3):
        o = eng.forward(w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        o = eng.forward(w)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"HTSAT B={B} N={N} flags={flags}: {dt*1e3:.3f} ms/step  {B/dt:.1f} clips/s  {B/dt*11.82e9/1e12:.1f} TFLOP/s",
          flush=True)
lib.wise_debug_set_htsat(0)
o = eng.forward(w).clone()
torch.cuda.synchronize()
for rep in range(2):
    hs = [eng.forward_pipelined(w) for _ in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hs = [eng.forward_pipelined(w) for _ in range(10)]
    o2 = hs[-1].result()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"HTSAT B={B} N={N} two batches in flight: {dt*1e3:.3f} ms/step  {B/dt:.1f} clips/s  same={torch.equal(o, o2)}", flush=True)
