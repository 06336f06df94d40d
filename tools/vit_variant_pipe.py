"""ViT-B/32 bs=256 with two batches in flight under forced GEMM variants (debug library): does a two-blocks-per-CU tiling,
slower on its own, win once the other stream's kernels can share the compute units?"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")
from wise_amd import _lib
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for
lib = _lib.lib()
spec = spec_for("ViT-B-32", "openai")
B = 256
eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=B)
x = torch.randn(B, 3, 224, 224, device="cuda")
for rep in range(2):
    for name, v, pol in (("overlap policy 0 (lone-stream tiles minus 320-row)", 0, 0), ("policy 7: QKV and fc1 128x128", 0, 7),
                         ("policy 5: fc1 128x128", 0, 5), ("policy 6: QKV 128x128", 0, 6), ("hint ignored (round-1 behaviour)", 0, 3)):
        lib.wise_debug_set_gemm_variant(v)
        lib.wise_debug_set_gemm_flags(pol << 4)
        for _ in range(4): eng.forward_pipelined(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hs = [eng.forward_pipelined(x) for _ in range(20)]
        hs[-1].result(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{name:36s}: {dt * 1e3:.3f} ms/step  {B / dt:.0f} frames/s", flush=True)
lib.wise_debug_set_gemm_variant(0)
lib.wise_debug_set_gemm_flags(0)
