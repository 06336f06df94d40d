"""one batch at a time: two half batches on two streams (wise_vit_forward) against the whole batch on one stream
(wise_vit_forward_single), per model and batch size"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for

for name, tag, batches in (("ViT-B-32", "openai", (32, 64, 128, 256)), ("ViT-B-16", "openai", (64, 256)),
                           ("ViT-L-14", "openai", (32, 64, 256))):
    spec = spec_for(name, tag)
    eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=max(batches))
    for b in batches:
        x = torch.randn(b, 3, spec.image_size, spec.image_size, device="cuda")
        res = {}
        for mode, kw in (("two streams", {}), ("one stream", {"single_stream": True})):
            for _ in range(3):
                eng.forward(x, **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 20 if b * spec.tokens < 30000 else 6
            for _ in range(n):
                eng.forward(x, **kw)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / n * 1e3
        print(f"{name:9s} bs={b:4d} rows={b * spec.tokens:6d}: two streams {res['two streams']:8.3f} ms   one stream {res['one stream']:8.3f} ms", flush=True)
    del eng
    torch.cuda.empty_cache()
