"""ViT-B/32 bs=256 forward_pipelined with 2, 3 and 4 batches in flight (VitEngine.pipeline_depth): python tools/vit_depth.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for  # noqa: E402

spec = spec_for("ViT-B-32", "openai")
sd = random_state_dict(spec, 0)
g = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.randn(256, 3, 224, 224, generator=g, device="cuda") for _ in range(4)]
for rnd in range(2):
    for depth in (2, 3, 4):
        eng = VitEngine(spec, sd, max_batch=256)
        eng.pipeline_depth = depth
        for i in range(8):
            h = eng.forward_pipelined(xs[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(60):
            h = eng.forward_pipelined(xs[i % 4])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 60
        print(f"round {rnd} depth {depth}: {dt * 1e3:.3f} ms per step ({256 / dt / 1e3:.1f} k frames/s)", flush=True)
        del eng
