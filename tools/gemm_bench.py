"""A/B the GEMM variants on the ViT shapes (one process, interleaved rounds, random data).
python tools/gemm_bench.py [variants...]   -> prints TFLOP/s per (shape, variant), and max |err|."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib  # noqa: E402

SHAPES = [  # (name, M, N, K, mode)
    ("qkv   12800x2304x768", 12800, 2304, 768, 0),
    ("out   12800x768x768", 12800, 768, 768, 3),
    ("fc    12800x3072x768", 12800, 3072, 768, 1),
    ("proj  12800x768x3072", 12800, 768, 3072, 3),
    ("h-qkv  6400x2304x768", 6400, 2304, 768, 0),
    ("h-out  6400x768x768", 6400, 768, 768, 3),
    ("h-fc   6400x3072x768", 6400, 3072, 768, 1),
    ("h-proj 6400x768x3072", 6400, 768, 3072, 3),
    ("L14qkv 65792x3072x1024", 65792, 3072, 1024, 0),
    ("L14proj 65792x1024x4096", 65792, 1024, 4096, 3),
]


SQUARE = [  # calibration against the guide's 256x256 8-phase figures (1320-1340 TF @4096^3, ~1470 @8192^3)
    ("sq4096", 4096, 4096, 4096, 0),
    ("sq8192", 8192, 8192, 8192, 0),
]


HTSAT = [  # the Swin GEMMs of cfg-5 (128 clips): stage 2 (C=192), 3 (C=384), 4 (C=768); mode 2 = erf GELU
    ("s2 qkv 131072x576x192", 131072, 576, 192, 0), ("s2 proj 131072x192x192", 131072, 192, 192, 3),
    ("s2 fc1 131072x768x192", 131072, 768, 192, 2), ("s2 fc2 131072x192x768", 131072, 192, 768, 3),
    ("s3 qkv 32768x1152x384", 32768, 1152, 384, 0), ("s3 proj 32768x384x384", 32768, 384, 384, 3),
    ("s3 fc1 32768x1536x384", 32768, 1536, 384, 2), ("s3 fc2 32768x384x1536", 32768, 384, 1536, 3),
    ("s4 qkv 8192x2304x768", 8192, 2304, 768, 0), ("s4 proj 8192x768x768", 8192, 768, 768, 3),
    ("s4 fc1 8192x3072x768", 8192, 3072, 768, 2), ("s4 fc2 8192x768x3072", 8192, 768, 3072, 3),
    ("merge1 131072x192x384", 131072, 192, 384, 4), ("merge2 32768x384x768", 32768, 384, 768, 4),
]


def main():
    args = sys.argv[1:]
    if "--htsat" in args:
        args.remove("--htsat")
        SHAPES[:] = HTSAT
    if "--square" in args:
        args.remove("--square")
        SHAPES[:] = SQUARE
    library = "--library" in args   # calibration only: time torch.mm (hipBLASLt / rocBLAS) on the same operands
    if library:
        args.remove("--library")
    variants = [int(v) for v in args] or [0, 1, 2, 3]
    lib = _lib.lib()
    lib.wise_debug_set_gemm_variant.argtypes = [C.c_int]
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, M, N, K, mode in SHAPES:
        A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device="cuda")
        ref = None
        if M <= 12800:
            ref = A.float() @ W.float().t() + bias   # (mode 3 adds into a zeroed output)
        res = {}
        outs = {}
        for v in variants:
            lib.wise_debug_set_gemm_variant(v)
            out = torch.zeros(M, N, dtype=torch.float32 if mode in (3, 4) else torch.bfloat16, device="cuda")
            rc = lib.wise_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), M, N, K, mode, out.data_ptr(),
                                    _lib.stream_ptr())
            _lib.check(rc, "gemm")
            torch.cuda.synchronize()
            if ref is not None:
                r = ref if mode == 3 else (ref * torch.sigmoid(1.702 * ref) if mode == 1 else
                                           (torch.nn.functional.gelu(ref) if mode == 2 else ref))
                err = (out.float() - r).abs().max().item()
            else:
                err = float("nan")
            outs[v] = err
            res[v] = []
        for rnd in range(5):
            for v in variants:
                lib.wise_debug_set_gemm_variant(v)
                out = torch.zeros(M, N, dtype=torch.float32 if mode in (3, 4) else torch.bfloat16, device="cuda")
                for _ in range(3):
                    lib.wise_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), M, N, K, mode, out.data_ptr(),
                                       _lib.stream_ptr())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                iters = 20
                for _ in range(iters):
                    lib.wise_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), M, N, K, mode, out.data_ptr(),
                                       _lib.stream_ptr())
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / iters * 1e-3)
        line = f"{name:26s}"
        if library:   # plain bf16 product, no bias / activation / residual: an upper bound for the library's epilogues
            Wt = W.t()
            for backend in ("cublaslt", "cublas"):
                torch.backends.cuda.preferred_blas_library(backend)
                for _ in range(5):
                    o = torch.mm(A, Wt)
                ts = []
                for rnd in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        o = torch.mm(A, Wt)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 20 * 1e-3)
                ts.sort()
                line += f" | {backend}: {2.0 * M * N * K / ts[2] / 1e12:7.1f} TF {ts[2]*1e6:7.1f}us"
        for v in variants:
            ts = sorted(res[v])
            tf = 2.0 * M * N * K / ts[len(ts) // 2] / 1e12
            tfb = 2.0 * M * N * K / ts[0] / 1e12
            line += f" | v{v}: {tf:7.1f} TF (best {tfb:7.1f}) {ts[len(ts)//2]*1e6:7.1f}us err {outs[v]:.3g}"
        print(line, flush=True)
    lib.wise_debug_set_gemm_variant(0)


if __name__ == "__main__":
    main()
