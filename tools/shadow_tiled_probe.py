"""timing experiment: the batched shadow scan with the addressing a TILED shadow layout would have (results meaningless,
same bytes) against the row-major addressing, and both without the matrix products"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")
from wise_amd import _lib
from wise_amd.index.flat_ip import FlatIPIndex
lib = _lib.lib()
N, d = 10_000_000, 512
X = torch.empty(N, d, device="cuda")
g = torch.Generator(device="cuda").manual_seed(3)
for s in range(0, N, 1_000_000):
    X[s:s + 1_000_000] = torch.nn.functional.normalize(torch.randn(1_000_000, d, device="cuda", generator=g), dim=1)
Q = torch.nn.functional.normalize(torch.randn(256, d, device="cuda", generator=g), dim=1)
idx = FlatIPIndex(d, shadow=True).adopt(X)
for rep in range(2):
    for name, flags in (("row-major", 0), ("tiled addressing", 1 << 27), ("row-major, no lists", 1 << 9),
                        ("row-major, no MFMA no lists", 3 << 9), ("tiled addressing, no MFMA no lists", (1 << 27) | (3 << 9))):
        lib.wise_debug_set_scan(4 | flags, 0)
        for nq in (64, 128):
            for _ in range(2): idx.search_device(Q[:nq], 10)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): idx.search_device(Q[:nq], 10)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
            print(f"{name:28s} nq={nq:3d}: {dt * 1e3:.3f} ms/pass  {N * d * 2 / dt / 1e12:.2f} TB/s", flush=True)
lib.wise_debug_set_scan(4, 0)
