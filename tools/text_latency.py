"""single-query latency of the CLIP text tower (ViT-B/32 text), and a batch of 8 ViT-B/32 frames"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.feature.text import TextEngine, random_text_state_dict, text_spec_for
from wise_amd.feature.vit import VitEngine, random_state_dict, spec_for
tspec = text_spec_for("ViT-B-32", "openai")
teng = TextEngine(tspec, random_text_state_dict(tspec, 0), max_batch=8)
toks = torch.zeros(8, tspec.context, dtype=torch.int32, device="cuda"); toks[:, 0] = 49406; toks[:, 1:6] = 1234; toks[:, 6] = 49407
spec = spec_for("ViT-B-32", "openai")
eng = VitEngine(spec, random_state_dict(spec, 0), max_batch=8)
x = torch.randn(8, 3, 224, 224, device="cuda")
for name, fn in (("text tower, 1 query", lambda: teng.forward(toks[:1])), ("text tower, 8 queries", lambda: teng.forward(toks)),
                 ("ViT-B/32, 8 frames (the reference's chunk)", lambda: eng.forward(x))):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(f"{name:45s} {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms", flush=True)

# XLM-RoBERTa-large text tower of the reference's default model pair
from wise_amd.feature.xlmr_text import XLMR_SPECS, XlmrTextEngine, random_xlmr_state_dict
xs = XLMR_SPECS["xlm-roberta-large-ViT-H-14"]
xeng = XlmrTextEngine(xs, random_xlmr_state_dict(xs, 0), max_batch=256)
xt = torch.full((256, xs.context), xs.pad_id, dtype=torch.int32, device="cuda"); xt[:, 0] = 0; xt[:, 1:9] = 4321; xt[:, 9] = 2
for name, fn, n in (("XLM-R large text tower, 1 query (graph)", lambda: xeng.forward(xt[:1]), 1),
                    ("XLM-R large text tower, 256 queries", lambda: xeng.forward(xt), 256)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name:45s} {dt * 1e3:.4f} ms  {n / dt:.0f} queries/s  {n / dt * xs.flops_per_query() / 1e12:.1f} TFLOP/s", flush=True)
xeng.graph_max_batch = 0
for _ in range(3): xeng.forward(xt[:1])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): xeng.forward(xt[:1])
torch.cuda.synchronize(); print(f"{'XLM-R large text tower, 1 query (no graph)':45s} {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms", flush=True)
