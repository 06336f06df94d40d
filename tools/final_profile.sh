#!/bin/bash
# Round-end evidence run on the GPU box: bench line, rocprofv3 per-kernel statistics of the same command, and the
# two PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run with kernel-trace only).  Outputs under gpurun_out/.
#   usage (from the repo root on the box):  bash tools/final_profile.sh <tag>
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
python3 bench.py --steps 20 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o stats -- python3 bench.py --steps 20 --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1 || exit 1
python3 tools/prof_top.py $OUT/${TAG}_stats/stats_results.db 40 --csv $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_kernel_stats.txt
echo "stats done"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_fetch.log 2>&1 || exit 1
echo "fetch pass done"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_write.log 2>&1 || exit 1
echo "write pass done"
F=$(find $OUT/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/${TAG}_pmc_write -name "*counter_collection.csv" | head -1)
cp "$F" $OUT/${TAG}_pmc_fetch_counter_collection.csv
cp "$W" $OUT/${TAG}_pmc_write_counter_collection.csv
ls -la $OUT | tail -12
