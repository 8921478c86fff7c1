#!/bin/bash
# Secondary-configuration bench lines, per-shape GEMM table, stand-alone kernel timings (GPU box; run through gpurun from
# the repo root):   bash tools/collect_secondary.sh r3   -> gpurun_out/<tag>_sec/*.json|txt   (copy what is judged into profiles/)
set -uo pipefail
TAG=${1:-r6}
O=gpurun_out/${TAG}_sec
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default_20steps.json 2> $O/err.txt                                   # the driver's command
python bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-precision-table > $O/bench_steady_80steps.json 2>> $O/err.txt
python bench.py --queries 16 --steps 20 --warmup 5 --no-cpu-baseline --no-precision-table > $O/bench_q16.json 2>> $O/err.txt   # round 2's step size
python bench.py --steps 10 --warmup 3 --skip-rate 0.25 --no-cpu-baseline --no-precision-table > $O/bench_skip025.json 2>> $O/err.txt
python bench.py --k 50 --subset 0 --steps 10 --warmup 3 --no-cpu-baseline --no-precision-table > $O/bench_k50_fiq.json 2>> $O/err.txt
python bench.py --k 200 --queries 32 --steps 10 --warmup 3 --no-cpu-baseline --no-precision-table > $O/bench_k200_f16.json 2>> $O/err.txt
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-precision-table > $O/bench_bf16.json 2>> $O/err.txt                 # rounds 1-3 headline mode
python bench.py --dtype mixed --steps 20 --warmup 5 --no-cpu-baseline --no-precision-table > $O/bench_mixed.json 2>> $O/err.txt
python bench.py --stream-dtype split --steps 20 --warmup 5 --no-cpu-baseline --no-precision-table > $O/bench_f16_split.json 2>> $O/err.txt   # text stream fp32, ViT fp16
python bench.py --image-size 384 --queries 16 --steps 6 --warmup 2 --no-cpu-baseline --no-precision-table > $O/bench_384px.json 2>> $O/err.txt
python bench.py --mode bank --steps 10 --warmup 3 > $O/bench_bank_mode.json 2>> $O/err.txt
python bench.py --mode loop > $O/bench_loop_mode.json 2>> $O/err.txt
python bench.py --mode loop --loop-queries 4181 --query-batch 64 > $O/bench_loop_cirr_val_4181.json 2>> $O/err.txt                               # the whole CIRR val split, K = 100 + 5, metrics included
python bench.py --mode loop --k 50 --subset 0 --loop-queries 2017 --query-batch 64 > $O/bench_loop_fiq_dress_2017.json 2>> $O/err.txt            # FashionIQ dress, K = 50
python bench.py --mode loop --k 100 --subset 0 --loop-queries 6016 --query-batch 64 > $O/bench_loop_config3_fiq_all_6016.json 2>> $O/err.txt   # BASELINE configs[3] whole (all three FashionIQ splits' query count) on ONE GPU
python bench.py --mode loop --k 200 --subset 5 --loop-queries 512 --query-batch 64 > $O/bench_loop_config4_k200_512.json 2>> $O/err.txt          # BASELINE configs[4] whole on ONE GPU
python bench.py --dtype text32 --steps 20 --warmup 5 --no-cpu-baseline --no-precision-table > $O/bench_text32.json 2>> $O/err.txt   # the factories' mode for real weights (round 6: split8 operands)
python bench.py --dtype text32x3 --steps 10 --warmup 3 --no-cpu-baseline --no-precision-table > $O/bench_text32x3.json 2>> $O/err.txt   # round 5's three-product form
python bench.py --dtype text32 --image-size 384 --queries 16 --steps 6 --warmup 2 --no-cpu-baseline --no-precision-table > $O/bench_384px_text32.json 2>> $O/err.txt
python bench.py --tokens 40 --steps 10 --warmup 3 --no-cpu-baseline --no-precision-table --no-rank-fidelity > $O/bench_tokens40.json 2>> $O/err.txt   # captions beyond the fold's 32 tokens: projected cross-attention path
python bench.py --tokens 40 --dtype text32 --steps 10 --warmup 3 --no-cpu-baseline --no-precision-table --no-rank-fidelity > $O/bench_tokens40_text32.json 2>> $O/err.txt
python tools/split8_bench.py > $O/split8_gemm_shapes.txt 2>> $O/err.txt
tools/f8_probe > $O/f8_probe.txt 2>> $O/err.txt
tools/attn_mem_probe > $O/attn_mem_probe.txt 2>> $O/err.txt
python bench.py --mode latency --k 100 --steps 40 --warmup 5 > $O/bench_latency_k100.json 2>> $O/err.txt                         # one query, launch by launch vs one HIP graph
CIR_VIT_LNFOLD=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-precision-table --no-rank-fidelity > $O/bench_lnfold_off.json 2>> $O/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-precision-table --no-rank-fidelity > $O/bench_lnfold_on.json 2>> $O/err.txt
python bench.py --mode train --image-size 384 --steps 10 --warmup 3 > $O/bench_train_mode.json 2>> $O/err.txt                    # stage2_train.py step, B = 16
python bench.py --mode train --image-size 384 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_mode_bf16.json 2>> $O/err.txt
python bench.py --mode train --image-size 384 --img-tune --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_mode_img_tune.json 2>> $O/err.txt                # --blip-img-tune: ViT reverse pass + second AdamW buffer
python bench.py --mode train --image-size 224 --train-batch 32 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_train_mode_b32_224.json 2>> $O/err.txt
Q=64 python tools/gemm_shapes.py $O/gemm_shapes.json > $O/gemm_shapes.txt 2>> $O/err.txt
python tools/gemm_ab.py > $O/gemm_ab.txt 2>> $O/err.txt
python tools/attn_bench.py > $O/attn_bench.txt 2>> $O/err.txt
python tools/cls_xattn_check.py > $O/cls_xattn.txt 2>> $O/err.txt
rocminfo | grep -E "Marketing Name|Max Clock|Compute Unit|gfx" > $O/rocminfo.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["unit"], "ms/step", d.get("ms_per_step"), "path frac", d.get("path_frac_of_mfma_peak"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
