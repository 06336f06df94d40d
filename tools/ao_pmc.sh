#!/bin/bash
# PMC pass over tools/attn_oproj_bench.py: bash tools/ao_pmc.sh <counter> [<counter> ...]  -> gpurun_out/ao_pmc_<counter>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for C in "$@"; do
  D=gpurun_out/ao_pmc_$C
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 tools/attn_oproj_bench.py 5 > $D.log 2>&1 || { echo "$C: rocprofv3 failed"; tail -3 $D.log; continue; }
  f=$(find $D -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] || { echo "$C: no csv"; continue; }
  python3 - "$f" "$C" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"][:40], r.get("Grid_Size", "")
    acc[k].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "attn" in k[0] or "gemm_w4" in k[0] or "attention" in k[0]:
        print(sys.argv[2], k, "launches", len(v), "mean", sum(v) / len(v))
PY
done
