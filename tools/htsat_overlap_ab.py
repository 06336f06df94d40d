"""HTSAT (128 clips x 10 s) with two batches in flight under the overlap tile policies of the debug library"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")
from wise_amd import _lib
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict
lib = _lib.lib()
eng = HtsatEngine(random_htsat_state_dict(0), max_batch=128, max_samples=480000)
w = 0.1 * torch.randn(128, 480000, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
for rep in range(2):
    for name, pol in (("policy 0 (lone-stream tiles)", 0), ("policy 1 (128x128 only)", 1), ("hint ignored", 3)):
        lib.wise_debug_set_gemm_flags(pol << 4)
        for _ in range(3): eng.forward_pipelined(w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hs = [eng.forward_pipelined(w) for _ in range(10)]
        hs[-1].result(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{name:32s}: {dt * 1e3:.3f} ms/step  {128 / dt:.0f} clips/s", flush=True)
lib.wise_debug_set_gemm_flags(0)
