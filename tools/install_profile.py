"""Copy one tools/final_profile.sh sequence from gpurun_out/ (scratch) into profiles/ (tracked) and regenerate
profiles/pmc_traffic.json from its PMC passes:   python tools/install_profile.py <tag> [<tag to remove> ...]"""
import gzip
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G, P = ROOT / "gpurun_out", ROOT / "profiles"
tag, old = sys.argv[1], sys.argv[2:]
for o in old:
    for f in P.glob(f"{o}_*"):
        f.unlink()
for name in ("bench.json", "kernel_stats.csv", "kernel_stats.txt", "roofline_kernel_stats.csv", "roofline_kernel_stats.txt",
             "roofline_only.json", "cnn14_kernel_stats.csv", "cnn14_kernel_stats.txt"):
    shutil.copy(G / f"{tag}_{name}", P / f"{tag}_{name}")
for name in ("pmc_fetch", "pmc_write", "pmc_fetch_pre", "pmc_write_pre", "pmc_fetch_htsat", "pmc_write_htsat", "pmc_mfma"):
    src = G / f"{tag}_{name}_counter_collection.csv"
    with open(src, "rb") as fi, gzip.open(P / f"{tag}_{name}_counter_collection.csv.gz", "wb") as fo:
        shutil.copyfileobj(fi, fo)
c = lambda n: str(G / f"{tag}_{n}_counter_collection.csv")
subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_traffic.py"), c("pmc_fetch") + "," + c("pmc_fetch_pre"),
                c("pmc_write") + "," + c("pmc_write_pre"), c("pmc_fetch_htsat"), c("pmc_write_htsat"), "3"], check=True,
               stdout=subprocess.DEVNULL)
print("installed", tag, "removed", old)
