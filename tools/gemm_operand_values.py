"""does the value distribution of the A operand move the GEMM's speed (clock under load)?  same kernel, same shape"""
import ctypes as C, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd import _lib
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 12800, 2304, 768
W = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, generator=g, device="cuda")
out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
base = torch.randn(M, K, generator=g, device="cuda")
cases = {"N(0,1)": base, "N(0,1)*8": base * 8, "N(3,1)": base + 3, "N(0,1)*exp(N(0,1)) per column": base * torch.exp(torch.randn(K, device="cuda", generator=g)),
         "zeros": base * 0}
res = {k: [] for k in cases}
for rnd in range(4):
    for name, a in cases.items():
        A = a.to(torch.bfloat16).contiguous()
        for _ in range(3):
            lib.wise_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), M, N, K, 0, out.data_ptr(), _lib.stream_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.wise_gemm_bf16(A.data_ptr(), W.data_ptr(), bias.data_ptr(), M, N, K, 0, out.data_ptr(), _lib.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for name, ts in res.items():
    print(f"{name:34s} {sorted(ts)[len(ts)//2]:7.1f} us  (min {min(ts):.1f})", flush=True)
