#!/bin/bash
# The training step's profile artefacts alone (after a host-side change of the step): kernel statistics over the timed steps, PMC summary, the four bench lines.
#   bash tools/collect_train_r6.sh   -> gpurun_out/r6_train_* , gpurun_out/r6_sec2/bench_train_mode*.json
set -uo pipefail
TAG=r6
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT/r6_sec2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_train_stats -o run -- python3 $ROOT/bench.py --mode train --image-size 384 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_train_stats.json 2> $OUT/${TAG}_train_stats.err
for grp in "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${grp%%:*}; ctr=${grp#*:}
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $OUT/${TAG}_train_pmc_$name -o run -- python3 $ROOT/bench.py --mode train --image-size 384 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_train_pmc_$name.json 2> $OUT/${TAG}_train_pmc_$name.err
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/${TAG}_train_pmc_summary.json $OUT/${TAG}_train_pmc_sq $OUT/${TAG}_train_pmc_fetch $OUT/${TAG}_train_pmc_write > $OUT/${TAG}_train_pmc_summary.txt 2>&1
find $OUT/${TAG}_train_pmc_sq $OUT/${TAG}_train_pmc_fetch $OUT/${TAG}_train_pmc_write -name "*kernel_trace.csv" -delete
python3 tools/train_trace.py $OUT/${TAG}_train_stats 3 4 50 > $OUT/${TAG}_train_kstats.txt 2>&1
find $OUT/${TAG}_train_stats -name "*kernel_trace.csv" -delete
O=$OUT/r6_sec2
python bench.py --mode train --image-size 384 --steps 10 --warmup 3 > $O/bench_train_mode.json 2>> $O/err.txt
python bench.py --mode train --image-size 384 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_mode_bf16.json 2>> $O/err.txt
python bench.py --mode train --image-size 384 --img-tune --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_mode_img_tune.json 2>> $O/err.txt
python bench.py --mode train --image-size 224 --train-batch 32 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_train_mode_b32_224.json 2>> $O/err.txt
head -14 $OUT/${TAG}_train_kstats.txt
for f in $O/bench_train_mode*.json; do python3 - "$f" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print(sys.argv[1].split('/')[-1], d["value"], "ms/step", d.get("ms_per_step"), d["legs_ms"], d["roofline"]["frac"])
PY
done
