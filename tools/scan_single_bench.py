"""single-query flat search (10M x 512, top-10): fp32 scan against the two-stage (threshold form) search over the bf16
shadow, on iid rows and on clustered rows (runs of 20 near-duplicates, cosine >= 0.999)"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from wise_amd.index.flat_ip import FlatIPIndex

N, d = 10_000_000, 512
g = torch.Generator(device="cuda").manual_seed(5)


def iid():
    X = torch.empty(N, d, device="cuda")
    for s in range(0, N, 1_000_000):
        X[s:s + 1_000_000] = torch.nn.functional.normalize(torch.randn(1_000_000, d, device="cuda", generator=g), dim=1)
    return X


def clustered(per=20, spread=0.03):
    X = torch.empty(N, d, device="cuda")
    items = N // per
    for s in range(0, items, 50_000):
        n = min(50_000, items - s)
        base = torch.nn.functional.normalize(torch.randn(n, 1, d, device="cuda", generator=g), dim=2)
        blk = base + (spread / d ** 0.5) * torch.randn(n, per, d, device="cuda", generator=g)
        X[s * per:(s + n) * per] = torch.nn.functional.normalize(blk, dim=2).reshape(-1, d)
    return X


for data_name, make in (("iid", iid), ("clustered x20", clustered)):
    X = make()
    Qr = torch.nn.functional.normalize(torch.randn(100, d, device="cuda", generator=g), dim=1)
    Qn = torch.nn.functional.normalize(X[torch.randint(0, N, (100,), device="cuda", generator=g)] +
                                       0.02 * Qr, dim=1)       # queries that have true neighbours
    for qname, Q in (("random queries", Qr), ("neighbour queries", Qn)):
        res = {}
        for name, shadow in (("fp32 scan", False), ("two-stage", True)):
            idx = FlatIPIndex(d, shadow=shadow).adopt(X)
            for i in range(3): idx.search_device(Q[i:i + 1], 10)
            torch.cuda.synchronize(); c0 = idx.shadow_counts(); t0 = time.perf_counter()
            outs = [idx.search_device(Q[i:i + 1], 10) for i in range(100)]
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
            c1 = idx.shadow_counts()
            res[name] = outs
            print(f"{data_name:14s} {qname:18s} {name:10s}: {dt * 1e3:.3f} ms/query  {1 / dt:7.1f} q/s   "
                  f"from shadow / handed to fp32: {c1[0] - c0[0]}/{c1[1] - c0[1]}", flush=True)
            del idx
        a, b = res["fp32 scan"], res["two-stage"]
        print("   ids identical:", all(torch.equal(x[1], y[1]) for x, y in zip(a, b)),
              " max |score diff|:", max(float((x[0] - y[0]).abs().max()) for x, y in zip(a, b)), flush=True)
    del X
    torch.cuda.empty_cache()
