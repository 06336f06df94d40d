"""Where a K-tile of the ping-pong GEMM spends its cycles: s_memtime stamps of one block (debug library only).
slots per K-tile: 0 loop top | 1 after vmcnt wait | 2 after barrier B1 | 3 DMA issued + fragment reads issued |
4 after 2nd vmcnt wait | 5 after lgkmcnt(0) | 6 after barrier B2 | 7 after the 32-MFMA cluster"""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")
from wise_amd import _lib  # noqa: E402

lib = _lib.lib()
lib.wise_debug_set_gemm_variant.argtypes = [C.c_int]
lib.wise_debug_set_gemm_stamps.argtypes = [C.c_void_p, C.c_int]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, mode, variant in (("qkv", 12800, 2304, 768, 0, 40), ("fc2", 12800, 768, 3072, 3, 40),
                                     ("sq4096", 4096, 4096, 4096, 0, 40)):
    A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device="cuda")
    out = torch.zeros(M, N, dtype=torch.float32 if mode == 3 else torch.bfloat16, device="cuda")
    lib.wise_debug_set_gemm_variant(variant)
    for blk in (0, 130, 300):
        buf = torch.zeros(8192, dtype=torch.int64, device="cuda")
        for _ in range(5):   # warm: clocks, caches
            lib.wise_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), M, N, K, mode, out.data_ptr(), _lib.stream_ptr())
        lib.wise_debug_set_gemm_stamps(buf.data_ptr(), blk)
        lib.wise_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), M, N, K, mode, out.data_ptr(), _lib.stream_ptr())
        torch.cuda.synchronize()
        lib.wise_debug_set_gemm_stamps(None, 0)
        b = buf.cpu().view(2, 512, 8)
        nk = K // 32
        for grp, gname in ((0, "early"), (1, "late ")):
            t = b[grp, :nk].double()
            d = {"vmcnt1": t[:, 1] - t[:, 0], "B1": t[:, 2] - t[:, 1], "issue": t[:, 3] - t[:, 2], "vmcnt2": t[:, 4] - t[:, 3],
                 "lgkm": t[:, 5] - t[:, 4], "B2": t[:, 6] - t[:, 5], "mfma": t[:, 7] - t[:, 6]}
            per = (t[-1, 7] - t[0, 0]) / nk
            mid = slice(4, nk - 4)
            print(f"{name} block {blk} {gname}: {per:7.0f} cyc/K-tile | " +
                  "  ".join(f"{k} {v[mid].mean():6.0f}" for k, v in d.items()), flush=True)
lib.wise_debug_set_gemm_variant(0)
