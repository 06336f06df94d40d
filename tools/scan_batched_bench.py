"""batched flat scan (10M x 512, top-10): the variants behind wise_debug_set_scan, timed and cross-checked"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib
from wise_amd.index.flat_ip import FlatIPIndex
lib = _lib.lib()
N, d = 10_000_000, 512
X = torch.nn.functional.normalize(torch.randn(N, d, device="cuda"), dim=1)
idx = FlatIPIndex(d, shadow=False).adopt(X)
idx_shadow = FlatIPIndex(d, shadow=True).adopt(X)
Q = torch.nn.functional.normalize(torch.randn(256, d, device="cuda"), dim=1)
variants = [("f32 MFMA, DMA ring", 1 << 11), ("split regs x4, 32/pass", (4 << 12) | (1 << 25)), ("split regs x4, 64/pass", 4 << 12),
            ("two-stage, bf16 shadow 64/pass", -1)]
ref = None
for name, flags in variants * 2:
    lib.wise_debug_set_scan(4 | max(flags, 0), 0)
    use = idx_shadow if flags < 0 else idx
    for nq in (32, 64, 256):
        for _ in range(2): D, I = use.search_device(Q[:nq], 10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 6 if nq == 32 else 2
        for _ in range(n): D, I = use.search_device(Q[:nq], 10)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        msg = f"{name:28s} nq={nq:3d}: {dt * 1e3:.3f} ms/call  {nq / dt:.0f} q/s"
        if nq == 256:
            if ref is None: ref = (D.clone(), I.clone())
            msg += f"  ids equal to first: {bool(torch.equal(I, ref[1]))}  max |dscore| {float((D - ref[0]).abs().max()):.2e}"
        print(msg, flush=True)
lib.wise_debug_set_scan(4, 0)
