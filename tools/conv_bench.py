"""A/B the 3x3-convolution tiles on the Cnn14 layer shapes (one process, interleaved rounds, random data):
python tools/conv_bench.py [batch] [samples]  -> TFLOP/s per (layer, variant: 1 = 128-row tile, 2 = ping-pong, 0 = product)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib  # noqa: E402
from wise_amd.feature.cnn14 import CHANNELS, HOP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 480000
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
zeros = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
T, F, cin = N // HOP + 1, 64, 1
for i, cout in enumerate(CHANNELS):
    for j, ci in enumerate((cin, cout)):
        POOL = 1 if (j == 1 and i < len(CHANNELS) - 1) else 0   # the block's second convolution carries the pooling
        if ci >= 64:
            M = B * T * F
            rows = (M + 255) // 256 * 256
            x = torch.randn(rows, ci, generator=g, device="cuda").to(torch.bfloat16)
            w = (torch.randn(cout, 9 * ci, generator=g, device="cuda") * (2.0 / (9 * ci)) ** 0.5).to(torch.bfloat16)
            bias = 0.1 * torch.randn(cout, generator=g, device="cuda")
            out = torch.empty(rows, cout, dtype=torch.bfloat16, device="cuda")
            res, outs = {}, {}
            for v in (1, 2, 0):
                lib.wise_debug_set_gemm_flags(v << 8)
                _lib.check(lib.wise_conv3x3_relu_bf16(x.data_ptr(), w.data_ptr(), bias.data_ptr(), zeros.data_ptr(), B, T, F, ci,
                                                      cout, POOL, out.data_ptr(), _lib.stream_ptr()), "conv")
                torch.cuda.synchronize()
                outs[v] = out[:min(1 << 14, (B * (T // 2) * (F // 2)) if POOL else M)].float().clone()
                res[v] = []
            for rnd in range(3):
                for v in (1, 2, 0):
                    lib.wise_debug_set_gemm_flags(v << 8)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        lib.wise_conv3x3_relu_bf16(x.data_ptr(), w.data_ptr(), bias.data_ptr(), zeros.data_ptr(), B, T, F, ci, cout,
                                                   POOL, out.data_ptr(), _lib.stream_ptr())
                    e1.record()
                    torch.cuda.synchronize()
                    res[v].append(e0.elapsed_time(e1) / 5 * 1e-3)
            fl = 2.0 * (B * (T // 2) * (F // 2) * 4 if POOL else M) * cout * 9 * ci
            line = f"block {i + 1} conv {j + 1}: M={M:9d} T={T:4d} F={F:2d} {ci:4d}->{cout:4d}"
            for v in (1, 2, 0):
                t = sorted(res[v])[1]
                line += f" | v{v}: {fl / t / 1e12:6.1f} TF {t * 1e6:7.1f} us"
            line += f" | max|v2-v1| {(outs[2] - outs[1]).abs().max().item():.3g}"
            print(line, flush=True)
            del x, w, out
    if i < len(CHANNELS) - 1:
        T, F = T // 2, F // 2
    cin = cout
lib.wise_debug_set_gemm_flags(0)
