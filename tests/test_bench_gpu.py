"""bench.py end to end on the GPU box: the single-rank contract line, and the multi-rank launch / gather / timing logic
exercised with TWO ranks sharing the one GPU (gloo instead of RCCL, which refuses two ranks on a device - the knobs
bench.py documents as CIR_BENCH_BACKEND / CIR_BENCH_DEVICE).  Child processes only: nothing is exec'ed."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text: str) -> dict:
    lines = [ln for ln in text.splitlines() if ln.startswith('{"metric"')]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_single_rank_contract_line():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--queries", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["value"] > 0 and d["unit"] == "triplets/s" and d["dtype"] == "f16"
    assert d["config"]["precision_mode"].startswith("fp16 MFMA operands") and d["config"]["residual_stream"] == "f16"     # the library default
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["kernel"].startswith("cir::gemm") and "traffic" in rf and rf["all_gemm_kernels"]["launches_per_step"] > 100
    assert d["config"]["subset"] == 5 and d["config"]["triplets_per_step_rank0"] == 2 * 105
    assert d["executed_gflop_per_triplet"] > 0 and d["device"]["compute_units"] == 256
    # round 5: the step re-run in the exact (fp32) mode referees this run's rank order
    fid = d["rank_fidelity"]
    assert fid["queries"] == 2 and fid["positions"] == 200 and 0.0 < fid["exact_positions"] <= 1.0 and fid["kendall_tau"] > 0.9
    assert fid["max_abs_dlogit"] < 0.02 and fid["subset"]["queries"] == 2
    ex = fid["exact_mode"]
    assert ex["triplets_per_s"] > 100 and 0.3 < ex["gemm_frac_of_f32_mfma_peak"] < 1.0 and ex["f32_mfma_peak_tflops"] == 157.3


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_two_rank_dry_run_on_one_gpu(launcher):
    """Both launch forms: `python bench.py --gpus 2` (bench.py starts its own two ranks as child processes before any GPU
    call) and the driver's `python -m torch.distributed.run ... bench.py --gpus 2`."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, CIR_BENCH_BACKEND="gloo", CIR_BENCH_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--queries", "3", "--skip-rate", "0.34",
            "--no-precision-table"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29533"] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["queries_per_step_per_gpu"] == 3 and d["config"]["subset"] == 5
    # 6 queries, skip rate 0.34: skipped queries score only their 5 subset members, and only scored pairs count
    assert d["config"]["triplets_per_step_rank0"] in (5 + 2 * 105, 2 * 5 + 105, 3 * 105, 15)
    assert r.stdout.count('{"metric"') == 1          # rank 0 alone prints the line
    c = d["collective"]
    assert c["ranks"] == 2 and c["own_block_bit_identical"] and c["checksum_equal_on_all_ranks"] and len(c["per_rank_triplets_per_step"]) == 2


def test_eight_rank_dry_run_and_a_dying_rank():
    """SURVEY 8(e) at the rank count BASELINE configs[3] / [4] name, as far as ONE GPU allows: `python bench.py --gpus 8` starts its own
    eight ranks (all pinned to device 0, gloo instead of RCCL - which refuses two ranks per device), every rank scores its block of
    the 16 queries, ONE all-gather of scores + indices closes the timed region; then the same launch with a rank that dies before the
    exchange: its peers block in the collective and the launcher must end them and return non-zero (bench.launch_ranks: 20-s window)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import time
    env = dict(os.environ, CIR_BENCH_BACKEND="gloo", CIR_BENCH_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--queries", "2", "--skip-rate", "0.2",
           "--no-precision-table"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    c = d["collective"]
    assert d["n_gpus"] == 8 and c["ranks"] == 8 and len(c["per_rank_triplets_per_step"]) == 8 and r.stdout.count('{"metric"') == 1
    assert c["own_block_bit_identical"] and c["checksum_equal_on_all_ranks"]
    assert sum(c["per_rank_triplets_per_step"]) > 0 and max(c["per_rank_triplets_per_step"]) - min(c["per_rank_triplets_per_step"]) <= 100   # balanced_order: blocks differ by <= 1 scored query
    assert "rank_fidelity" not in d and "cpu_baseline" not in d                 # single-rank blocks only
    t0 = time.time()
    cmd2 = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "1", "--queries", "2", "--no-precision-table"]
    r = subprocess.run(cmd2, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, CIR_BENCH_FAIL_RANK="1"))
    assert r.returncode != 0 and '{"metric"' not in r.stdout and "a rank exited with code 3" in r.stderr, (r.returncode, r.stderr[-1500:])
    print(f"\n[dying rank] launcher returned {r.returncode} after {time.time() - t0:.0f} s")


def test_bank_mode_line():
    """SURVEY 8(f)-1 regime (`--mode bank`): candidates drawn from a resident index bank with cached K/V."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "bank", "--queries", "2", "--k", "20", "--index-size", "64",
                        "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["value"] > 0 and "index-bank reuse" in d["metric"] and d["config"]["index_size"] == 64


def test_one_rank_rccl_exchange_runs_on_device_tensors():
    """The RCCL branch of bench.py (init_process_group('nccl'), all_gather_into_tensor on DEVICE tensors, per-rank timing and
    checksum gathers) executed with a one-rank group on the single GPU of the box: what the 2/4/8-GPU runs do, minus peers."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, CIR_BENCH_INIT_PG="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--queries", "2",
                        "--skip-rate", "0.5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    c = d["collective"]
    assert c["backend"].startswith("rccl") and c["ranks"] == 1 and c["own_block_bit_identical"] and c["checksum_equal_on_all_ranks"]
    assert c["per_rank_ms_per_step"]["min"] == c["per_rank_ms_per_step"]["max"] > 0
    assert c["gathered_bytes_per_rank"] == 2 * 2 * 105 * (4 + 8) and sum(c["per_rank_triplets_per_step"]) == d["config"]["triplets_per_step_rank0"]


def test_loop_mode_line():
    """`--mode loop`: the reference's own loop (extract_index_features + generate_cirr_val_predictions + metrics) timed end
    to end beside the direct-engine number, the K/V-bank variant and the Level-1 per-query call."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "loop", "--loop-queries", "6", "--query-batch", "4", "--k", "20",
                        "--index-size", "48", "--index-batch", "32", "--skip-rate", "0.2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["value"] > 0 and "scoring loop" in d["metric"] and d["host_overhead_frac"] < 1
    # (the loop folds the cross-attention K / V projections to the query side since round 5, the K/V bank holds PROJECTED keys / values: the two
    #  paths agree within the operand rounding, no longer bit for bit; test_model_gpu.py::test_kv_bank_reuse_is_bit_identical pins the identity
    #  with the fold switched off)
    assert d["logits_max_abs_diff_bank_vs_loop"] < 5e-3 and d["level1_ms_per_query"] > 0 and len(d["recall"]) == len(d["recall_kv_bank"])


def test_train_mode_line():
    """`--mode train`: one stage2_train.py-shaped step (ViT, z_t, fusion forward in .train() mode, backward, AdamW) on a small
    batch; the line carries the per-leg times, the flop model's roofline fraction and the CPU oracle's training step beside it."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--train-batch", "4", "--image-size", "64", "--tokens", "12",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["value"] > 0 and "training step" in d["metric"] and d["unit"] == "triplets/s"
    assert set(d["legs_ms"]) == {"vit", "z_t", "fusion_forward", "backward", "adamw"} and all(v > 0 for v in d["legs_ms"].values())
    assert 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert 0 < d["loss"] < 10


def test_text32_and_latency_lines():
    """Round 5: `--dtype text32` (its rank_fidelity against the exact mode must beat a 16-bit run's by a wide margin even on 2 queries) and
    `--mode latency` (one query launch by launch against one captured HIP graph: identical logits)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dtype", "text32", "--steps", "1", "--warmup", "1", "--queries", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["dtype"] == "text32" and d["value"] > 0 and d["config"]["residual_stream"] == "split" and "split8" in d["config"]["precision_mode"]
    assert d["rank_fidelity"]["kendall_tau"] > 0.99 and d["rank_fidelity"]["max_abs_dlogit"] < 2e-3 and "text32" not in d["rank_fidelity"]     # (no sub-block of itself)
    # the default mode's line carries the real-weights mode twice: its order against the exact referee, and its throughput at the headline's step count
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--queries", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    t32 = d["rank_fidelity"]["text32"]
    assert d["dtype"] == "f16" and t32["kendall_tau"] >= d["rank_fidelity"]["kendall_tau"] and t32["max_abs_dlogit"] < d["rank_fidelity"]["max_abs_dlogit"]
    assert d["precision_table"]["text32+split_stream"] > 0 and d["precision_table"]["text32x3+split_stream"] > 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "latency", "--k", "20", "--steps", "20", "--warmup", "3"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["identical_logits"] is True and d["higher_is_better"] is False and d["latency_ms"]["hip_graph"]["p50_ms"] > 0
    assert d["latency_ms"]["hip_graph"]["p50_ms"] <= d["latency_ms"]["launches_from_python"]["p50_ms"] * 1.1
