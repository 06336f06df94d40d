"""The MLP of an HTSAT Swin block (stage 2: M = 131072, C = 192; stage 3: M = 32768, C = 384; 128 clips) as the two GEMM
calls and as wise_mlp_stream (one kernel), per call:  python tools/mlp_stream_bench.py [reps=50]"""
import sys

import torch

sys.path.insert(0, ".")
from wise_amd import _lib  # noqa: E402
from wise_amd.feature.htsat import mlp_stream_weights  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    lib = _lib.lib()
    st = _lib.stream_ptr()
    for M, C in ((131072, 192), (32768, 384)):
        g = torch.Generator(device="cuda").manual_seed(C)
        h = torch.randn(M, C, generator=g, device="cuda").to(torch.bfloat16)
        w1 = (torch.randn(4 * C, C, generator=g, device="cuda") * C ** -0.5).to(torch.bfloat16)
        w2 = (torch.randn(C, 4 * C, generator=g, device="cuda") * (4 * C) ** -0.5).to(torch.bfloat16)
        ws = mlp_stream_weights(w1, w2)
        b1, b2 = torch.randn(4 * C, generator=g, device="cuda"), torch.randn(C, generator=g, device="cuda")
        x = torch.randn(M, C, generator=g, device="cuda")
        a = torch.empty(M, 4 * C, dtype=torch.bfloat16, device="cuda")

        def two():
            lib.wise_gemm_bf16(h.data_ptr(), w1.data_ptr(), b1.data_ptr(), M, 4 * C, C, 2, a.data_ptr(), st)
            lib.wise_gemm_bf16(a.data_ptr(), w2.data_ptr(), b2.data_ptr(), M, C, 4 * C, 3, x.data_ptr(), st)

        def one():
            lib.wise_mlp_stream(h.data_ptr(), ws.data_ptr(), b1.data_ptr(), b2.data_ptr(), x.data_ptr(), M, C, st)

        row = []
        for fn in (two, one):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / reps * 1e3)
        fl = 2.0 * M * C * 4 * C * 2
        print(f"M={M} C={C}: two GEMMs {row[0]:.1f} us ({fl / row[0] / 1e6:.0f} TFLOP/s), one kernel {row[1]:.1f} us ({fl / row[1] / 1e6:.0f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
