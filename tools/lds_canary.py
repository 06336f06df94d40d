"""run the LDS canary beside single kernels on another stream"""
import ctypes, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib
lib = _lib.lib()
lib.wise_debug_lds_canary.restype = ctypes.c_int
lib.wise_debug_lds_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sts = [torch.cuda.Stream() for _ in range(2)]
M = 131072
x = torch.randn(M, 96, device="cuda"); lnw = torch.ones(96, device="cuda"); lnb = torch.zeros(96, device="cuda")
Wq = (0.05 * torch.randn(288, 96, device="cuda")).bfloat16(); bq = torch.zeros(288, device="cuda")
W1 = (0.05 * torch.randn(384, 96, device="cuda")).bfloat16(); b1 = torch.zeros(384, device="cuda")
W2 = (0.05 * torch.randn(96, 384, device="cuda")).bfloat16(); b2 = torch.zeros(96, device="cuda")
qkv = torch.empty(M, 288, device="cuda", dtype=torch.bfloat16)
hb = torch.randn(M, 96, device="cuda").bfloat16()
a4 = torch.empty(M, 384, device="cuda", dtype=torch.bfloat16)
s1 = sts[1].cuda_stream
P = lambda t: t.data_ptr()
neigh = {
    "nothing": lambda: 0,
    "gemm_ln qkv": lambda: lib.wise_gemm_ln_bf16(P(x), P(lnw), P(lnb), P(Wq), P(bq), M, 288, 96, 1e-5, 0, P(qkv), s1),
    "mlp96_fused": lambda: lib.wise_mlp96_fused(P(x), P(lnw), P(lnb), P(W1), P(b1), P(W2), P(b2), M, 1e-5, s1),
    "gemm fc1": lambda: lib.wise_gemm_bf16(P(hb), P(W1), P(b1), M, 384, 96, 2, P(a4), s1),
    "layernorm": lambda: lib.wise_layernorm_f32_bf16(P(x), P(lnw), P(lnb), M, 96, 1e-5, P(hb), s1),
}
lib.wise_debug_vgpr_canary.restype = ctypes.c_int
lib.wise_debug_vgpr_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for name, fn in neigh.items():
    rep = torch.zeros(1024, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6): _lib.check(fn(), name)
    _lib.check(lib.wise_debug_vgpr_canary(512, 34816, 200, 4, P(rep), sts[0].cuda_stream), "canary")
    for _ in range(6): _lib.check(fn(), name)
    torch.cuda.synchronize()
    r = [int(v) & 0xffffffff for v in rep.tolist()]
    print(f"VGPR canary beside {name}: {r[0]} bad registers", flush=True)
    for e in range(min(r[0], 24)):
        q = r[16 + e * 8: 24 + e * 8]
        print(f"   block {q[0]} it {q[1]} slot {q[2]} thread {q[3]} (lane {q[3] & 63}) got {q[4]:#010x} exp {q[5]:#010x} hwid {q[6]:#x}")
for lds_bytes in ():
    for name, fn in neigh.items():
        rep = torch.zeros(8, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(6): _lib.check(fn(), name)
        _lib.check(lib.wise_debug_lds_canary(512, lds_bytes, 200, 4, P(rep), sts[0].cuda_stream), "canary")
        for _ in range(6): _lib.check(fn(), name)
        torch.cuda.synchronize()
        r = [int(v) & 0xffffffff for v in rep.tolist()]
        print(f"canary {lds_bytes} B beside {name}: {r[0]} bad words" + (f"  first: block {r[1]} it {r[2]} word {r[3]} got {r[4]:#x} exp {r[5]:#x} hwid {r[6]:#x}" if r[0] else ""), flush=True)
