"""attention kernel alone: time per launch for the towers' shapes under the queries-per-wave knob of the debug library"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")
from wise_amd import _lib
lib = _lib.lib()
for B, T, H in ((256, 257, 16), (256, 197, 12), (256, 576, 16), (256, 50, 12), (128, 257, 16)):
    qkv = (torch.randn(B * T, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
    o = torch.empty(B * T, H * 64, dtype=torch.bfloat16, device="cuda")
    for qt in (0, 4, 2, 3, 5):     # 4 / 2: 64 / 32 queries per wave; 3 / 5: 48 / 32 queries with the next key block prefetched
        lib.wise_debug_set_vit_streams(2 | ((qt if qt else 255) << 8))   # 255: back to the heuristic
        for _ in range(3): lib.wise_attention_bf16(qkv.data_ptr(), B, T, H, o.data_ptr(), _lib.stream_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): lib.wise_attention_bf16(qkv.data_ptr(), B, T, H, o.data_ptr(), _lib.stream_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        fl = 4.0 * B * H * T * T * 64
        print(f"B={B} T={T} H={H} qt={qt}: {dt * 1e6:8.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s", flush=True)
lib.wise_debug_set_vit_streams(2)

# head width 80 (ViT-H/14): 48 queries per wave (the product) against 32 queries with hoisted loads (debug variant 6)
B, T, H = 256, 257, 16
qkv = (torch.randn(B * T, 3 * H * 80, device="cuda") * 1.5).to(torch.bfloat16)
o = torch.empty(B * T, H * 80, dtype=torch.bfloat16, device="cuda")
for rep in range(2):
    for qt in (255, 6):
        lib.wise_debug_set_vit_streams(2 | (qt << 8))
        for _ in range(3): lib.wise_attention_dh_bf16(qkv.data_ptr(), B, T, H, 80, o.data_ptr(), _lib.stream_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): lib.wise_attention_dh_bf16(qkv.data_ptr(), B, T, H, 80, o.data_ptr(), _lib.stream_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"dh=80 B={B} T={T} H={H} variant={qt}: {dt * 1e6:8.1f} us  {4.0 * B * H * T * T * 80 / dt / 1e12:6.1f} TFLOP/s", flush=True)
lib.wise_debug_set_vit_streams(2 | (255 << 8))
