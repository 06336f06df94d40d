"""Where does the one-off stall in the first timed region of bench.py's IVF leg come from?  (9.36 ms/query in
r03_c and 3.56 in r03_d against 0.187 in r03_e — one stall of 0.2-0.5 s after three warm-up calls.)

Replays the leg's sequence — a 20 GB tensor dropped + empty_cache() as the legs before it leave things, the synthesised
lists, three warm-up pairs — then times EVERY call of the first region on the host with a sync behind each one, and prints
the slowest calls with their index, the allocator's counters before / after and the number of hipMalloc calls it made.

    python tools/ivf_stall.py [rows=10000000] [dim=512] [--no-prefill]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from wise_amd.index.ivf_flat import IVFFlatIPIndex  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 512
    if "--no-prefill" not in sys.argv:
        big = torch.empty(rows, dim, device="cuda")          # what the flat-search legs leave behind
        big.normal_()
        sh = torch.empty(rows, dim, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        del big, sh
        torch.cuda.empty_cache()
    nlist, per = 31620, max(1, rows // 31620)
    n = nlist * per
    gen = torch.Generator(device="cuda").manual_seed(200)
    dirs = torch.randn(nlist, dim, generator=gen, device="cuda")
    Xi = torch.empty(n, dim, dtype=torch.float32, device="cuda")
    for s0 in range(0, nlist, 1024):
        e0 = min(nlist, s0 + 1024)
        blk = dirs[s0:e0, None, :] + 0.7 * torch.randn(e0 - s0, per, dim, generator=gen, device="cuda")
        Xi[s0 * per:e0 * per] = (blk / blk.norm(dim=2, keepdim=True)).reshape(-1, dim)
    cent = Xi.view(nlist, per, dim).mean(dim=1)
    cent = cent / cent.norm(dim=1, keepdim=True)
    ivf = IVFFlatIPIndex(dim, nlist)
    ivf.set_centroids(cent)
    ivf.adopt_lists(Xi, torch.arange(n, device="cuda", dtype=torch.int64) + 1,
                    torch.arange(nlist + 1, device="cuda", dtype=torch.int64) * per)
    Qi = Xi[torch.randint(0, n, (1000,), generator=gen, device="cuda")] + 0.05 * torch.randn(1000, dim, generator=gen,
                                                                                              device="cuda")
    ivf.nprobe = 32
    hold = {}
    for i in range(3):
        hold["a"] = ivf.search_device(Qi[i:i + 1], 10)
        hold["a"] = ivf.search_device(Qi[:256], 10)
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    for region in range(3):
        per_call = []
        t_region = time.perf_counter()
        for i in range(50):
            t0 = time.perf_counter()
            hold["a"] = ivf.search_device(Qi[i:i + 1], 10)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            per_call.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3, i))
        total = (time.perf_counter() - t_region) * 1e3
        worst = sorted(per_call, reverse=True)[:3]
        print(f"region {region}: {total:.2f} ms for 50 calls; slowest (total ms, host-enqueue ms, call): "
              + ", ".join(f"({a:.2f}, {b:.2f}, #{c})" for a, b, c in worst), flush=True)
    st1 = torch.cuda.memory_stats()
    for key in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.current"):
        print(key, st0.get(key), "->", st1.get(key))
    # the same 3 x 50 calls the way bench.py times them: no sync inside the region
    for region in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(50):
            hold["a"] = ivf.search_device(Qi[i:i + 1], 10)
        torch.cuda.synchronize()
        print(f"unsynced region {region}: {(time.perf_counter() - t0) * 1e3 / 50:.4f} ms per query", flush=True)


if __name__ == "__main__":
    main()
