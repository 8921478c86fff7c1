#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r3
# -> gpurun_out/<tag>_stats (kernel-trace --stats), gpurun_out/<tag>_pmc_{sq,fetch,write} (one PMC pass each; --pmc is
#    never combined with other trace domains), aggregated by tools/kstats.py / tools/pmc_summary.py (which reads the
#    workload from the bench line each pass printed and stamps the hash of the kernel sources).
# The profiled program is `python3 bench.py ...` directly after `--` (no env / bash -c hop: the profiler's preloaded
# library has initialised the GPU before the program starts).
set -euo pipefail
TAG=${1:-r6}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-precision-table --no-rank-fidelity > $OUT/${TAG}_stats.json 2> $OUT/${TAG}_stats.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc_sq -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-precision-table --no-rank-fidelity > $OUT/${TAG}_pmc_sq.json 2> $OUT/${TAG}_pmc_sq.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-precision-table --no-rank-fidelity > $OUT/${TAG}_pmc_fetch.json 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-precision-table --no-rank-fidelity > $OUT/${TAG}_pmc_write.json 2> $OUT/${TAG}_pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_train_stats -o run -- python3 $ROOT/bench.py --mode train --image-size 384 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_train_stats.json 2> $OUT/${TAG}_train_stats.err
# the training step's kernels under the same three PMC passes (MFMA-pipe utilisation, fetched / written bytes per launch)
for grp in "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${grp%%:*}; ctr=${grp#*:}
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $OUT/${TAG}_train_pmc_$name -o run -- python3 $ROOT/bench.py --mode train --image-size 384 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_train_pmc_$name.json 2> $OUT/${TAG}_train_pmc_$name.err
done
cd $ROOT
find $OUT/${TAG}_train_pmc_sq $OUT/${TAG}_train_pmc_fetch $OUT/${TAG}_train_pmc_write -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py $OUT/${TAG}_train_pmc_summary.json $OUT/${TAG}_train_pmc_sq $OUT/${TAG}_train_pmc_fetch $OUT/${TAG}_train_pmc_write > $OUT/${TAG}_train_pmc_summary.txt 2>&1
python3 tools/train_trace.py $OUT/${TAG}_train_stats 3 4 50 > $OUT/${TAG}_train_kstats.txt 2>&1      # optimizer steps 3..6 = the 4 timed ones (2 warm-up before, 1 leg-instrumented after)
find $OUT/${TAG}_train_stats -name "*kernel_trace.csv" -delete
# keep what travels back small: the per-dispatch kernel traces are not needed (the counter CSVs carry timestamps)
python3 tools/kstats_trace.py $OUT/${TAG}_stats 2 6 > $OUT/${TAG}_kstats.txt 2>&1          # the 6 timed steps' launches only
# the mode the factories set for real weights (text32: split8 operands), same command
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_text32_stats -o run -- python3 $ROOT/bench.py --dtype text32 --steps 6 --warmup 2 --no-cpu-baseline --no-precision-table --no-rank-fidelity > $OUT/${TAG}_text32_stats.json 2> $OUT/${TAG}_text32_stats.err
cd $ROOT
python3 tools/kstats_trace.py $OUT/${TAG}_text32_stats 2 6 > $OUT/${TAG}_text32_kstats.txt 2>&1
find $OUT/${TAG}_stats $OUT/${TAG}_text32_stats $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_summary.json $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_pmc_summary.txt 2>&1
du -sh $OUT/${TAG}_* | tail -12
