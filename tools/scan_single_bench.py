"""single-query flat search (10M x 512, top-10): fp32 scan against the two-stage search over the bf16 shadow"""
import ctypes, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd import _lib
from wise_amd.index.flat_ip import FlatIPIndex
lib = _lib.lib()
lib.wise_ip_shadow_stats.restype = ctypes.c_int
lib.wise_ip_shadow_stats.argtypes = [ctypes.c_void_p]
N, d = 10_000_000, 512
X = torch.nn.functional.normalize(torch.randn(N, d, device="cuda"), dim=1)
Q = torch.nn.functional.normalize(torch.randn(200, d, device="cuda"), dim=1)
res = {}
for name, shadow in (("fp32 scan", False), ("two-stage (bf16 shadow)", True), ("fp32 scan", False), ("two-stage (bf16 shadow)", True)):
    idx = FlatIPIndex(d, shadow=shadow).adopt(X)
    for i in range(3): idx.search_device(Q[i:i + 1], 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = [idx.search_device(Q[i:i + 1], 10) for i in range(100)]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    st = (ctypes.c_int * 2)(); lib.wise_ip_shadow_stats(st)
    res[name] = outs
    print(f"{name:26s}: {dt * 1e3:.3f} ms/query  {1 / dt:.1f} q/s   certified/fallback since last: {st[0]}/{st[1]}", flush=True)
    del idx
a, b = res["fp32 scan"], res["two-stage (bf16 shadow)"]
print("ids identical:", all(torch.equal(x[1], y[1]) for x, y in zip(a, b)),
      " max |score diff|:", max(float((x[0] - y[0]).abs().max()) for x, y in zip(a, b)))
