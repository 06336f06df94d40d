"""A/B the IP scan under knobs: python tools/scan_bench.py [N] [d]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")  # tuning switches live only in libwise_hip_debug.so
from wise_amd import _lib  # noqa: E402
from wise_amd.index.flat_ip import FlatIPIndex  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = _lib.lib()
X = torch.randn(N, d, device="cuda")
X /= X.norm(dim=1, keepdim=True)
idx = FlatIPIndex(d).adopt(X, None, id_base=1)
Q = torch.randn(96, d, device="cuda")
Q /= Q.norm(dim=1, keepdim=True)
ref = None
for rows, bpc in ((4, 0), (4 + 512, 0), (4 + 1024, 0)):  # batched kernel: full, no-DMA ablation, no-MFMA ablation
    lib.wise_debug_set_scan(rows, bpc)
    idx._ws = None  # workspace size depends on the grid
    for nq in (1, 32):
        for _ in range(3):
            D, I = idx.search_device(Q[:nq], 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            D, I = idx.search_device(Q[(i % 2):(i % 2) + nq] if nq > 32 else Q[i:i + nq], 10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        if nq == 1:
            D0, I0 = idx.search_device(Q[:1], 10)
            if ref is None:
                ref = I0.clone()
            ok = torch.equal(I0, ref)
        print(f"rows={rows} blocks/CU={bpc or 'auto'} nq={nq}: {dt*1e3:7.3f} ms  {nq/dt:8.1f} q/s  "
              f"{N*d*4/dt/1e9:7.1f} GB/s  ids-same={ok}", flush=True)
