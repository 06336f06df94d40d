"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md's HBM
section prescribes) into profiles/pmc_traffic.json: HBM bytes per launch for the dominant kernels.

Corrections applied exactly as that guide states for gfx950:
  FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming reads -> doubled;
  WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Both counters are in KiB.
usage: python tools/pmc_traffic.py <fetch csv>[,<fetch csv>...] <write csv>[,<write csv>...] [<htsat fetch csv> <htsat write csv> <forwards>]
(comma-separated lists merge several profiled commands, e.g. bench.py --no-extra and tools/preproc_bench.py; the
optional HTSAT pair comes from tools/htsat_pmc.py: ALL kernels of `forwards` whole forwards, summed -> "htsat_forward")
"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


BY_GRID = {}   # counter -> {(kernel, grid size): (mean, launches)}: launches of one kernel on different problem sizes apart


def per_kernel(paths, counter):
    acc = defaultdict(lambda: [0.0, 0])
    accg = defaultdict(lambda: [0.0, 0])
    seen = set()
    for path in paths.split(","):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            key = (path, r["Dispatch_Id"], r["Kernel_Name"])
            acc[r["Kernel_Name"]][0] += float(r["Counter_Value"])
            accg[(r["Kernel_Name"], int(r["Grid_Size"]))][0] += float(r["Counter_Value"])
            if key not in seen:
                seen.add(key)
                acc[r["Kernel_Name"]][1] += 1
                accg[(r["Kernel_Name"], int(r["Grid_Size"]))][1] += 1
    BY_GRID[counter] = {k: (v[0] / max(v[1], 1), v[1]) for k, v in accg.items()}
    return {k: (v[0] / max(v[1], 1), v[1]) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write)):
        short = name.split("(")[0].replace("void ", "").replace("wise::", "")
        f, nf = fetch.get(name, (0.0, 0))
        w, nw = write.get(name, (0.0, 0))
        out[short] = {"launches": max(nf, nw), "fetch_size_kib_raw": round(f, 1), "write_size_kib": round(w, 1),
                      "hbm_read_bytes_per_launch": round(f * 1024 * 2), "hbm_write_bytes_per_launch": round(w * 1024),
                      "hbm_bytes_per_launch": round(f * 1024 * 2 + w * 1024)}
    # the same per (kernel, grid size): one kernel launched on several problem sizes (GEMM tiles of different shapes: the grid
    # tells them apart; the search kernels run a fixed grid — their passes are kept like-for-like by `bench.py --pmc-legs`,
    # which launches the headline shapes only)
    for (name, grid), (f, nf) in sorted(BY_GRID.get("FETCH_SIZE", {}).items()):
        short = name.split("(")[0].replace("void ", "").replace("wise::", "")
        w, nw = BY_GRID.get("WRITE_SIZE", {}).get((name, grid), (0.0, 0))
        out[short].setdefault("by_grid", {})[str(grid)] = {"launches": max(nf, nw), "hbm_read_bytes_per_launch": round(f * 2048),
                                                           "hbm_write_bytes_per_launch": round(w * 1024)}
    # aggregate keys bench.py looks up
    def agg(prefix, per=1):
        # "ip_scan_kernel" = the flat scan only; its inverted-list instantiation (last template argument true) is
        # reported as "ivf_scan_kernel"
        if prefix == "ivf_scan_kernel":
            ks = [k for k in out if k.startswith("ip_scan_kernel") and k.rstrip().endswith("true>")]
        elif prefix == "ip_scan_kernel":
            ks = [k for k in out if k.startswith("ip_scan_kernel") and not k.rstrip().endswith("true>")]
        else:
            ks = [k for k in out if k.startswith(prefix)]
        n = sum(out[k]["launches"] for k in ks)
        if not n:
            return None
        # per = 2 / 3: a pass is that many launches of the kernel (the sample pass, then the collect pass in one or two row ranges)
        return {"launches": n // per, "hbm_bytes_per_launch": round(sum(out[k]["hbm_bytes_per_launch"] * out[k]["launches"]
                                                                      for k in ks) / (n // per)), "kernels": ks}
    res = {"_note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request); KiB units",
           "per_kernel": out}
    for key, prefix, per in (("gemm_bf16_kernel", "gemm_", 1), ("ip_scan_kernel", "ip_scan_kernel", 1),
                             ("ip_scan_mfma_kernel", "ip_scan_mfma_kernel", 1),
                             ("ip_collect_bf16_kernel", "ip_collect_bf16_kernel", 1),
                             ("ip_collect_i8_kernel", "ip_collect_i8_kernel", 1),
                             ("ip_scan_split_direct_kernel", "ip_scan_split_direct_kernel", 2),
                             ("ip_scan_split64_kernel", "ip_scan_split64_kernel", 2),
                             ("ip_scan_shadow64_kernel", "ip_scan_shadow64_kernel", 3),
                             ("clip_resize_kernel", "clip_resize_kernel", 1), ("ivf_scan_kernel", "ivf_scan_kernel", 1),
                             ("attention_kernel", "attention_kernel", 1), ("layernorm_kernel", "layernorm_kernel", 1)):
        a = agg(prefix, per)
        if a:
            res[key] = a
    if len(sys.argv) >= 6:
        n_fwd = int(sys.argv[5])
        tot_f = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[3])) if r["Counter_Name"] == "FETCH_SIZE"
                    and "wise::" in r["Kernel_Name"])
        tot_w = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[4])) if r["Counter_Name"] == "WRITE_SIZE"
                    and "wise::" in r["Kernel_Name"])
        res["htsat_forward"] = {"forwards": n_fwd, "hbm_read_bytes_per_launch": round(tot_f * 1024 * 2 / n_fwd),
                                "hbm_write_bytes_per_launch": round(tot_w * 1024 / n_fwd),
                                "hbm_bytes_per_launch": round((tot_f * 2 + tot_w) * 1024 / n_fwd),
                                "note": "every kernel of a 128-clip forward summed (library kernels only)"}
        print("htsat_forward", res["htsat_forward"]["hbm_bytes_per_launch"] / 1e9, "GB per forward")
    dst = Path(__file__).resolve().parent.parent / "profiles" / "pmc_traffic.json"
    dst.write_text(json.dumps(res, indent=1))
    for k in ("gemm_bf16_kernel", "ip_scan_kernel", "ip_collect_bf16_kernel", "ip_collect_i8_kernel", "ip_scan_shadow64_kernel", "ip_scan_split_direct_kernel", "ip_scan_split64_kernel", "clip_resize_kernel"):
        if k in res:
            print(k, res[k]["hbm_bytes_per_launch"] / 1e6, "MB/launch over", res[k]["launches"], "launches")
    for k, v in out.items():
        if v["launches"] and ("gemm" in k or "scan" in k or "attention" in k or "layernorm" in k):
            print(f"  {k[:60]:60s} n={v['launches']:4d} read {v['hbm_read_bytes_per_launch']/1e6:9.1f} MB  write {v['hbm_write_bytes_per_launch']/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
