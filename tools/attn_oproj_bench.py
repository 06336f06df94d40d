"""The attention half of a ViT-B/32 block on the hi + lo stream: wise_attention_bf16 + wise_gemm_fold_resid against
wise_attention_oproj_fold (one kernel), per call, for a few batch sizes:  python tools/attn_oproj_bench.py [reps=200] [dbg=0]
With WISE_AMD_DEBUG_LIB=1 (the debug twin) dbg skips phases of the one kernel (bit 0 attention, 1 projection, 2 residual
epilogue) and bit 3 prints s_memtime stamps of workgroup 0 (shader cycles from kernel entry: attention done, barrier passed,
projection done, epilogue done, second barrier passed, end)."""
import sys

import torch

sys.path.insert(0, ".")
from wise_amd import _lib  # noqa: E402
from wise_amd.feature.vit import tile_out_proj  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    lib = _lib.lib()
    if len(sys.argv) > 2:
        lib.wise_debug_ao_set(int(sys.argv[2]))
    T, H, W = 50, 12, 768
    for B in (256, 128, 64, 8):
        M = B * T
        Mp = (M + 255) // 256 * 256
        g = torch.Generator(device="cuda").manual_seed(B)
        qkv = torch.randn(Mp, 3 * W, generator=g, device="cuda").to(torch.bfloat16)
        Wt = (torch.randn(W, W, generator=g, device="cuda") * W ** -0.5).to(torch.bfloat16)
        Wtiled = tile_out_proj(Wt)
        bias = torch.randn(W, generator=g, device="cuda")
        xs = torch.randn(2, Mp, W, generator=g, device="cuda").to(torch.bfloat16)
        ao = torch.zeros(Mp, W, dtype=torch.bfloat16, device="cuda")
        stats = torch.zeros(lib.wise_gemm_fold_stats_bytes(Mp, W) // 4, device="cuda")
        st = _lib.stream_ptr()

        def two():
            lib.wise_attention_bf16(qkv.data_ptr(), B, T, H, ao.data_ptr(), st)
            lib.wise_gemm_fold_resid(ao.data_ptr(), Wt.data_ptr(), bias.data_ptr(), Mp, W, W, xs.data_ptr(), Mp * W, stats.data_ptr(), 1e-5, 0, st)

        def one():
            lib.wise_attention_oproj_fold(qkv.data_ptr(), B, T, H, Wtiled.data_ptr(), bias.data_ptr(), xs.data_ptr(), Mp * W, stats.data_ptr(), 1e-5, st)

        row = []
        for fn in (two, one):
            for _ in range(10):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / reps * 1e3)
        print(f"B={B:3d}: attention + residual GEMM {row[0]:.1f} us, one kernel {row[1]:.1f} us", flush=True)
        if hasattr(lib, "wise_debug_ao_stamps") and B in (256, 8):
            import ctypes as C
            buf = (C.c_uint64 * 96)()
            lib.wise_debug_ao_stamps(buf)
            for w in (0, 5, 11):
                t = [buf[w * 8 + k] for k in range(7)]
                print(f"   wave {w:2d} stamps (shader cycles from entry): " + " ".join(str(int(t[k] - t[0])) for k in range(1, 7)))


if __name__ == "__main__":
    main()
