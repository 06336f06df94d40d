#!/bin/bash
# Round-end evidence run on the GPU box: bench line, rocprofv3 per-kernel statistics of the same command, and the
# two PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run with kernel-trace only).  Outputs under gpurun_out/.
#   usage (from the repo root on the box):  bash tools/final_profile.sh <tag>
set -u
TAG=${1:-rXX}
PART=${2:-all}     # "a": bench line + the two kernel-stats passes; "b": the PMC passes (a gpurun call lasts 20 minutes at most)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
mkdir -p $OUT
if [ "$PART" != "b" ]; then
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o stats -- python3 bench.py --steps 20 --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1 || exit 1
python3 tools/prof_top.py $OUT/${TAG}_stats/stats_results.db 40 --csv $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_kernel_stats.txt
echo "stats done"
# the single-stream bracketed pass alone: its per-kernel averages are the ones bench.py's roofline object reports
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_roofstats -o roof -- python3 bench.py --steps 20 --roofline-only > $OUT/${TAG}_roofline_only.json 2> $OUT/${TAG}_roofline_only.err || exit 1
python3 tools/prof_top.py $OUT/${TAG}_roofstats/roof_results.db 12 --csv $OUT/${TAG}_roofline_kernel_stats.csv > $OUT/${TAG}_roofline_kernel_stats.txt
echo "roofline stats done"
fi
if [ "$PART" = "a" ]; then ls -la $OUT | tail -8; exit 0; fi
# PMC passes (one counter per run, kernel-trace only): the headline legs without the extras, so that the per-kernel
# averages are those of the ViT-B/32 GEMMs and the 10M x 512 scans; the image-transform kernel from its own tool
for C in FETCH_SIZE WRITE_SIZE; do
  L=$(echo $C | tr 'A-Z' 'a-z' | cut -d_ -f1)
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$L -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --pmc-legs > $OUT/${TAG}_pmc_$L.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_${L}_pre -- python3 tools/preproc_bench.py 64 > $OUT/${TAG}_pmc_${L}_pre.log 2>&1 || exit 1
  cp "$(find $OUT/${TAG}_pmc_$L -name '*counter_collection.csv' | head -1)" $OUT/${TAG}_pmc_${L}_counter_collection.csv
  cp "$(find $OUT/${TAG}_pmc_${L}_pre -name '*counter_collection.csv' | head -1)" $OUT/${TAG}_pmc_${L}_pre_counter_collection.csv
  echo "$C passes done"
done
# HTSAT: every kernel of 3 whole forwards (the torch.randn fill is not a library kernel and is left out by name)
for C in FETCH_SIZE WRITE_SIZE; do
  L=$(echo $C | tr 'A-Z' 'a-z' | cut -d_ -f1)
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_${L}_htsat -- python3 tools/htsat_pmc.py 3 > $OUT/${TAG}_pmc_${L}_htsat.log 2>&1 || exit 1
  cp "$(find $OUT/${TAG}_pmc_${L}_htsat -name '*counter_collection.csv' | head -1)" $OUT/${TAG}_pmc_${L}_htsat_counter_collection.csv
done
echo "HTSAT PMC passes done"
# MS-CLAP 2022 Cnn14: per-kernel durations of whole forwards, one batch at a time (bs=128, 10-s clips)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_cnn14stats -o cnn -- python3 tools/cnn14_bench.py 128 480000 --serial-only > $OUT/${TAG}_cnn14.log 2>&1 || exit 1
python3 tools/prof_top.py $OUT/${TAG}_cnn14stats/cnn_results.db 14 --csv $OUT/${TAG}_cnn14_kernel_stats.csv > $OUT/${TAG}_cnn14_kernel_stats.txt
echo "Cnn14 stats done"
# matrix-core utilisation of the GEMM launches: busy cycles of the MFMA pipes over the launch's cycles (own pass)
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --roofline-only > $OUT/${TAG}_pmc_mfma.log 2>&1 || exit 1
cp "$(find $OUT/${TAG}_pmc_mfma -name '*counter_collection.csv' | head -1)" $OUT/${TAG}_pmc_mfma_counter_collection.csv
echo "MFMA pass done"
ls -la $OUT | tail -12
