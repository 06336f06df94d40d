"""which single kernel, running on another stream, disturbs the HTSAT front end"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib
from wise_amd.feature.htsat import HtsatEngine, random_htsat_state_dict
B, N, Fc = 32, 480000, 1024
eng = HtsatEngine(random_htsat_state_dict(0), max_batch=B, max_samples=N)
lib = _lib.lib()
w = 0.1 * torch.randn(B, N, device="cuda")
need = lib.wise_htsat_workspace_bytes(B, N)
K = 16
wsk = [torch.empty(need, dtype=torch.uint8, device="cuda") for _ in range(K)]
sts = [torch.cuda.Stream() for _ in range(2)]
out = torch.empty(B, 1024, device="cuda")
base = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def mel_of(ws):
    mel = torch.empty(B * Fc, 64, device="cuda")
    _lib.check(lib.wise_htsat_tap(0, ws.data_ptr(), B, N, mel.data_ptr(), mel.numel(), _lib.stream_ptr()), "tap")
    torch.cuda.synchronize()
    return mel
def front(ws):
    lib.wise_debug_set_htsat(base | 8)
    _lib.check(lib.wise_htsat_forward(eng.wb.data_ptr(), eng.pf.data_ptr(), w.data_ptr(), B, N, out.data_ptr(),
                                      ws.data_ptr(), ws.numel(), sts[0].cuda_stream), "fwd")
front(wsk[0]); torch.cuda.synchronize(); mel0 = mel_of(wsk[0])
M = 131072
x = torch.randn(M, 96, device="cuda"); lnw = torch.ones(96, device="cuda"); lnb = torch.zeros(96, device="cuda")
Wq = (0.05 * torch.randn(288, 96, device="cuda")).bfloat16(); bq = torch.zeros(288, device="cuda")
W1 = (0.05 * torch.randn(384, 96, device="cuda")).bfloat16(); b1 = torch.zeros(384, device="cuda")
W2 = (0.05 * torch.randn(96, 384, device="cuda")).bfloat16(); b2 = torch.zeros(96, device="cuda")
Wp = (0.05 * torch.randn(96, 96, device="cuda")).bfloat16()
qkv = torch.empty(M, 288, device="cuda", dtype=torch.bfloat16)
hb = torch.randn(M, 96, device="cuda").bfloat16()
a4 = torch.empty(M, 384, device="cuda", dtype=torch.bfloat16)
s1 = sts[1].cuda_stream
P = lambda t: t.data_ptr()
neigh = {
    "gemm_ln qkv": lambda: lib.wise_gemm_ln_bf16(P(x), P(lnw), P(lnb), P(Wq), P(bq), M, 288, 96, 1e-5, 0, P(qkv), s1),
    "mlp96_fused": lambda: lib.wise_mlp96_fused(P(x), P(lnw), P(lnb), P(W1), P(b1), P(W2), P(b2), M, 1e-5, s1),
    "gemm proj (mode 3, 96x96)": lambda: lib.wise_gemm_bf16(P(hb), P(Wp), P(b2), M, 96, 96, 3, P(x), s1),
    "gemm fc1 (mode 2, 384x96)": lambda: lib.wise_gemm_bf16(P(hb), P(W1), P(b1), M, 384, 96, 2, P(a4), s1),
    "gemm fc2 (mode 3, 96x384)": lambda: lib.wise_gemm_bf16(P(a4), P(W2), P(b2), M, 96, 384, 3, P(x), s1),
    "layernorm": lambda: lib.wise_layernorm_f32_bf16(P(x), P(lnw), P(lnb), M, 96, 1e-5, P(hb), s1),
}
import ctypes
lib.wise_debug_neighbour.restype = ctypes.c_int
lib.wise_debug_neighbour.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
srcbuf = torch.randint(0, 2**31 - 1, (4096 * 1024 + 1024,), dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
names = ["LDS-DMA x4", "LDS-DMA x1", "ds_bpermute", "MFMA", "plain LDS traffic", "global loads"]
for what, nm in enumerate(names):
    neigh["synthetic: " + nm] = (lambda w_: (lambda: lib.wise_debug_neighbour(w_, 2048, 61440, 64, P(srcbuf), P(sink), s1)))(what)

for name, fn in neigh.items():
    for t in wsk: t[:B * Fc * 256].zero_()
    torch.cuda.synchronize()
    for i, t in enumerate(wsk):
        for _ in range(3): _lib.check(fn(), name)
        front(t)
    for _ in range(3): _lib.check(fn(), name)
    torch.cuda.synchronize()
    tot = sum(0 if torch.equal(mel_of(t), mel0) else 1 for t in wsk)
    print(f"{name}: {tot} of {K} front ends wrong", flush=True)
