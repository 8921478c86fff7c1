#!/bin/bash
# One rocprofv3 --kernel-trace --stats pass of the benchmark step in a given mode, reduced to per-step kernel times:
#   bash tools/quick_stats.sh <tag> [bench.py arguments ...]      -> gpurun_out/<tag>_kstats.txt, gpurun_out/<tag>_stats.json
set -euo pipefail
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-precision-table --no-rank-fidelity "$@" > $OUT/${TAG}_stats.json 2> $OUT/${TAG}_stats.err
cd $ROOT
python3 tools/kstats_trace.py $OUT/${TAG}_stats 2 4 > $OUT/${TAG}_kstats.txt 2>&1      # launches of the 4 timed steps only
find $OUT/${TAG}_stats -name "*kernel_trace.csv" -delete
cat $OUT/${TAG}_kstats.txt | head -30
