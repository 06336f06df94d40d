"""batched two-stage search: one-piece against two-piece bf16 queries in the shadow scan (10M x 512 and 6.25M x 768, top-10),
same process, interleaved; ids and scores against the f32 scan"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("WISE_AMD_DEBUG_LIB", "1")
from wise_amd import _lib
from wise_amd.index.flat_ip import FlatIPIndex
lib = _lib.lib()
for N, d in ((10_000_000, 512), (6_250_000, 768)):
    X = torch.empty(N, d, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    for s in range(0, N, 1_000_000):
        n = min(1_000_000, N - s)
        X[s:s + n] = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(256, d, device="cuda", generator=g), dim=1)
    ref = FlatIPIndex(d, shadow=False).adopt(X)
    D0, I0 = ref.search_device(Q[:64], 10)
    del ref
    idx = FlatIPIndex(d, shadow=True).adopt(X)
    for rep in range(2):
        for name, flags in (("two pieces", 1 << 26), ("one piece", 0)):
            lib.wise_debug_set_scan(4 | flags, 0)
            for nq in (64, 256):
                for _ in range(2): D, I = idx.search_device(Q[:nq], 10)
                torch.cuda.synchronize(); c0 = idx.shadow_counts(); t0 = time.perf_counter()
                for _ in range(3): D, I = idx.search_device(Q[:nq], 10)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
                c1 = idx.shadow_counts()
                print(f"N={N} d={d} {name:10s} nq={nq:3d}: {dt * 1e3:.3f} ms/call  {nq / dt:.0f} q/s  shadow/fp32 {c1[0]-c0[0]}/{c1[1]-c0[1]}"
                      f"  ids==f32: {bool(torch.equal(I[:64], I0))}  max|dscore| {float((D[:64] - D0).abs().max()):.1e}", flush=True)
    lib.wise_debug_set_scan(4, 0)
    del idx, X
    torch.cuda.empty_cache()
