"""HTSAT (128 clips x 10 s): the weight-resident stage-1 MLP kernel against the staged one (debug library), one batch at a time"""
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
outs = {}
for rep in range(2):
    for name, fl in (("weight-resident MLP", 0), ("staged MLP", 1 << 7)):
        lib.wise_debug_set_gemm_flags(fl)
        for _ in range(3): o = eng.forward(w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): o = eng.forward(w)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        outs[name] = o.clone()
        print(f"{name:22s}: {dt * 1e3:.3f} ms/forward  {128 / dt:.0f} clips/s", flush=True)
a, b = outs["weight-resident MLP"], outs["staged MLP"]
print("cosine between the two:", float(torch.nn.functional.cosine_similarity(a, b).min()), " max |diff|", float((a - b).abs().max()))
lib.wise_debug_set_gemm_flags(0)
