set -uo pipefail
bash tools/collect_profiles.sh r6 2>&1 | tail -3
O=gpurun_out/r6_sec2; mkdir -p $O
python bench.py --mode train --image-size 384 --steps 10 --warmup 3 > $O/bench_train_mode.json 2>> $O/err.txt
python bench.py --mode train --image-size 384 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_mode_bf16.json 2>> $O/err.txt
python bench.py --mode train --image-size 384 --img-tune --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_mode_img_tune.json 2>> $O/err.txt
python bench.py --mode train --image-size 224 --train-batch 32 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_train_mode_b32_224.json 2>> $O/err.txt
python bench.py --mode loop --dtype text32 --loop-queries 4181 --query-batch 64 > $O/bench_loop_cirr_val_4181_text32.json 2>> $O/err.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default_20steps_final.json 2>> $O/err.txt
for f in $O/bench_*.json gpurun_out/r6_stats.json gpurun_out/r6_text32_stats.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["unit"], "ms/step", d.get("ms_per_step"), "path frac", d.get("path_frac_of_mfma_peak"), "traffic", (d.get("roofline") or {}).get("traffic"), "t32", (d.get("precision_table") or {}).get("text32+split_stream"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
head -8 gpurun_out/r6_kstats.txt; head -4 gpurun_out/r6_text32_kstats.txt
