"""A/B the ViT forward (frames/s) under library knobs: python tools/vit_bench.py [model] [batch]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib  # noqa: E402
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for  # noqa: E402


ATTN_QT = 0  # 4 forces the per-wave attention kernel with 64 queries per wave (A/B against the shared-K/V kernel)


def main():
    global ATTN_QT
    if "--qt4" in sys.argv:
        sys.argv.remove("--qt4")
        ATTN_QT = 4
    model = sys.argv[1] if len(sys.argv) > 1 else "ViT-B-32"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    lib = _lib.lib()
    spec = spec_for(model, "openai")
    eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=B)
    x = torch.randn(B, 3, 224, 224, device="cuda")
    ref = None
    # (streams, flags): bit 0 = no 320-row tiling; timing-only ablations: bit 1 = no LayerNorm, bit 2 = no attention
    configs = [(2, 0), (2, 2), (2, 4), (2, 6), (1, 0), (1, 6)] * 2
    for streams, flags in configs:
        lib.wise_debug_set_vit_streams(streams | (ATTN_QT << 8))
        lib.wise_debug_set_gemm_flags(flags)
        for _ in range(5):
            o = eng.forward(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20 if model == "ViT-B-32" else 5
        if "--pipe" in sys.argv:   # two whole batches in flight (VitEngine.forward_pipelined)
            hs = [eng.forward_pipelined(x) for _ in range(n)]
            o = hs[-1].result()
        else:
            for _ in range(n):
                o = eng.forward(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        if ref is None:
            ref = o.clone()
        same = torch.equal(o, ref)
        print(f"{model} B={B} streams={streams} flags={flags}: {dt*1e3:8.3f} ms/step  {B/dt:10.1f} frames/s  "
              f"{B/dt*spec.flops_per_frame()/1e12:7.1f} TFLOP/s  bit-identical-to-first={same}", flush=True)


if __name__ == "__main__":
    main()
