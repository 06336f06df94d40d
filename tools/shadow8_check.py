"""int8-shadow search against the f32 scan: python tools/shadow8_check.py [N] [d]   (ids and scores must be equal; timing)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.empty(N, d, device="cuda")
for s in range(0, N, 1 << 20):
    e = min(N, s + (1 << 20))
    X[s:e] = torch.nn.functional.normalize(torch.randn(e - s, d, device="cuda", generator=g), dim=1)
Xq = torch.empty(N, d, dtype=torch.int8, device="cuda")
scales = torch.empty(N, device="cuda")
norms = torch.zeros(4, device="cuda")
_lib.check(lib.wise_ip_shadow_i8(X.data_ptr(), N, d, Xq.data_ptr(), scales.data_ptr(), norms.data_ptr(), _lib.stream_ptr()), "shadow_i8")
torch.cuda.synchronize()
print("norms", norms.tolist(), flush=True)
# the shadow against its definition on a slice
sl = X[:4096]
sc = sl.abs().amax(1) / 127
ref = torch.clamp(torch.round(sl / sc[:, None]), -127, 127).to(torch.int8)
print("rows equal:", bool((ref == Xq[:4096]).all()), "scales equal:", bool(torch.equal(sc, scales[:4096])))
counters = torch.zeros(2, dtype=torch.int32, device="cuda")
for k in (10, 20, 100, 1000):
    need = max(lib.wise_ip_topk_shadow_workspace_bytes(N, d, 1, k), lib.wise_ip_topk_workspace_bytes(N, d, 1, k))
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    bad = 0
    for qi in range(6):
        q = torch.nn.functional.normalize(torch.randn(1, d, device="cuda", generator=g), dim=1)
        if qi % 2:
            q = torch.nn.functional.normalize(X[(qi * 7919) % N][None] + 0.05 * q, dim=1)
        D8 = torch.empty(1, k, device="cuda"); I8 = torch.empty(1, k, dtype=torch.int64, device="cuda")
        Df = torch.empty(1, k, device="cuda"); If = torch.empty(1, k, dtype=torch.int64, device="cuda")
        _lib.check(lib.wise_ip_topk_shadow8_f32(X.data_ptr(), Xq.data_ptr(), scales.data_ptr(), norms.data_ptr(), N, d, q.data_ptr(),
                                                 1, k, None, 0, D8.data_ptr(), I8.data_ptr(), counters.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), _lib.stream_ptr()), "shadow8")
        _lib.check(lib.wise_ip_topk_f32(X.data_ptr(), N, d, q.data_ptr(), 1, k, None, 0, Df.data_ptr(), If.data_ptr(), ws.data_ptr(),
                                         ws.numel(), _lib.stream_ptr()), "f32")
        torch.cuda.synchronize()
        if not (torch.equal(D8, Df) and torch.equal(I8, If)):
            bad += 1
            print("  MISMATCH k", k, "query", qi, (I8 != If).sum().item(), (D8 - Df).abs().max().item())
    q = torch.nn.functional.normalize(torch.randn(1, d, device="cuda", generator=g), dim=1)
    for _ in range(3):
        lib.wise_ip_topk_shadow8_f32(X.data_ptr(), Xq.data_ptr(), scales.data_ptr(), norms.data_ptr(), N, d, q.data_ptr(), 1, k, None, 0,
                                     D8.data_ptr(), I8.data_ptr(), counters.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        lib.wise_ip_topk_shadow8_f32(X.data_ptr(), Xq.data_ptr(), scales.data_ptr(), norms.data_ptr(), N, d, q.data_ptr(), 1, k, None, 0,
                                     D8.data_ptr(), I8.data_ptr(), counters.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"k={k}: mismatches {bad}/6; {dt * 1e3:.3f} ms/query = {1 / dt:.0f} q/s; counters {counters.tolist()}", flush=True)
