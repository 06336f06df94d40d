"""gcc build of the oracle's C restatement (oracle/ip_topk_ref.c -> oracle/_build/libwise_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__ and bench.py's cpu_baseline leg build or load this library; nothing
under wise_amd/ imports this module (tests/test_host_api.py::test_product_path_does_not_import_oracle).
`python -m oracle.build` or `oracle.build.build_oracle()`.
"""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

ODIR = Path(__file__).resolve().parent
SRC = ODIR / "ip_topk_ref.c"
OUT = ODIR / "_build" / "libwise_oracle.so"


def build_oracle(force: bool = False) -> Path | None:
    if not SRC.exists():
        return None
    if not force and OUT.exists() and OUT.stat().st_mtime >= SRC.stat().st_mtime:
        return OUT
    OUT.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", str(OUT), str(SRC), "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"oracle build failed:\n{r.stdout}")
    return OUT


if __name__ == "__main__":
    print("built", build_oracle(force="--force" in sys.argv))
