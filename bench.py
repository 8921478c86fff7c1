#!/usr/bin/env python3
"""Headline benchmark: query-candidate triplets scored / s at K=100 on synthetic 224x224 images and
32-token captions (BASELINE.json metric).

One step = one batch of `--queries` queries, each with its own K=100 candidate images, taken from
pixels and token ids already resident in HBM to sorted scores:
    ViT-B/16 over the Q*K candidate images and the Q reference images -> stage-I z_t per query ->
    two-branch fusion + cls_head per (query, candidate) -> per-query descending argsort
    [-> RCCL all-gather of the (Q,K) scores / indices when world_size > 1].
Queries shard across ranks with no data-path collective (weak scaling: per-GPU work fixed).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 5 --warmup 2

Rank 0 prints ONE JSON line.  `roofline` is measured on the dominant kernel (the MFMA GEMM):
algorithmic flops of every GEMM launch of one step / summed launch durations, from HIP events
recorded on the launch stream in an instrumented step after the timed region.  `cpu_baseline` is
the CPU oracle (a port of the reference's op sequence, fp32) on a bounded sample, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL / multi-process)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2516.6, "f16": 2516.6}  # dense MFMA peak, 256 CU x 2.4 GHz x 4096 flop/clk/CU (MI355X_MICROARCH.md)
D, H, F, LAYERS = 768, 12, 3072, 12


def algorithmic_gflop(n_tok: int, l: int, k: int):
    """SURVEY.md section 8(d) formulas (2*MAC, softmax/LN/GELU excluded)."""
    n = n_tok
    vit = 2 * (12 * (n * D * 3 * D + 2 * n * n * D + n * D * D + 2 * n * D * F) + (n - 1) * 768 * D)
    fuse = 0
    for layer in range(12):
        fuse += (2 * 3 * l * D * D + 4 * l * l * D + 2 * l * D * D) \
              + (2 * l * D * D + 4 * n * D * D + 4 * l * n * D + 2 * l * D * D + (l * 2 * D * D if layer >= 6 else 0)) \
              + 4 * l * D * F
    fuse = 2 * fuse + 2 * (2 * D * D + 2 * D)
    s1 = 2 * 12 * (3 * l * D * D + 2 * l * l * D + l * D * D + l * D * D + 2 * n * D * D + 2 * l * n * D + l * D * D + 2 * l * D * F)
    return dict(vit=vit / 1e9, fuse=fuse / 1e9, s1=s1 / 1e9, per_triplet=(vit + fuse + (vit + s1) / k) / 1e9)


def usable_cpus() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(threads: int):
    """Oracle (fp32 CPU port of the reference op sequence) on a bounded sample (~10-20 s of CPU work): 64
    images through the ViT, 8 queries through stage I, 2 queries x 100 candidates through the fusion;
    composed to triplets/s at K=100 with the same per-triplet accounting as the GPU metric."""
    from candidate_reranking_cir_amd import config, synthetic, weights
    from oracle import cir_oracle as O  # baseline leg only
    torch.set_num_threads(threads)
    g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
    sd2 = weights.synth_state_dict(weights.nlvr_param_spec(g, v), 0, "init")
    sd1 = weights.synth_state_dict(weights.retrieval_param_spec(g, v), 1, "init")
    ids = torch.stack([synthetic.caption_ids(q, 32) for q in range(8)])
    mask = torch.ones_like(ids)
    n_img, n_q, n_fq, k = 64, 8, 2, 100
    with torch.no_grad():
        imgs = synthetic.images(range(16), 224)
        O.img_embed(sd2, imgs[:2])                                  # warm the thread pool
        t0 = time.perf_counter()
        feats = torch.cat([O.img_embed(sd2, imgs) for _ in range(n_img // 16)])      # batches of 16 like utils.py:32
        t_vit = (time.perf_counter() - t0) / n_img
        t0 = time.perf_counter()
        zs = [O.stage1_z_t(sd1, feats[q:q + 1], ids[q:q + 1], mask[q:q + 1]) for q in range(n_q)]
        t_s1 = (time.perf_counter() - t0) / n_q
        cand = torch.cat([feats, feats[: k - n_img]])
        t0 = time.perf_counter()
        for q in range(n_fq):
            O.img_txt_fusion_val(sd2, zs[q], cand, ids[q:q + 1], mask[q:q + 1])
        t_fuse = (time.perf_counter() - t0) / (n_fq * k)
    per_triplet = t_vit + t_fuse + (t_vit + t_s1) / 100
    return {"value": round(1.0 / per_triplet, 3), "unit": "triplets/s", "cores": threads, "kind": "port",
            "sample": "fp32 oracle: %d images ViT-B/16@224 in batches of 16 (%.3fs/img), %d queries stage-I (%.3fs/query), "
                      "%d queries x %d candidates fusion L=32 (%.4fs/cand); composed as vit+fuse+(vit+s1)/100"
                      % (n_img, t_vit, n_q, t_s1, n_fq, k, t_fuse)}


def bank_mode(args, m2, m1, dev, dt, rank, world):
    """Real-dataset regime, reported separately from the headline metric: the index is encoded once
    (ViT tokens + per-layer cross-attention K/V stay resident in HBM), then queries draw their K candidates from it."""
    from candidate_reranking_cir_amd import ops, synthetic
    import torch.distributed as dist
    q_n, k, n_idx = args.queries, args.k, args.index_size
    gen = torch.Generator(device=dev).manual_seed(99)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bank = torch.cat([m2.img_embed16(torch.randn((min(512, n_idx - i), 3, args.image_size, args.image_size), generator=gen, device=dev).to(dt))
                      for i in range(0, n_idx, 512)])
    torch.cuda.synchronize(); t_vit = time.perf_counter() - t0
    kvb = m2.build_kv_bank(bank)
    torch.cuda.synchronize(); t_kv = time.perf_counter() - t0 - t_vit
    ids = torch.stack([synthetic.caption_ids(rank * q_n + q, args.tokens) for q in range(q_n)]).to(dev)
    mask = torch.ones_like(ids)
    qidx = torch.arange(q_n, device=dev).repeat_interleave(k)
    rng = torch.Generator(device="cpu").manual_seed(7 + rank)
    ref_rows = torch.randint(0, n_idx, (q_n,), generator=rng).to(dev)
    cand_rows = torch.stack([torch.randperm(n_idx, generator=rng)[:k] for _ in range(q_n)]).reshape(-1).to(dev)

    def step():
        z = m1.z_t(ops.gather_rows(bank, ref_rows), ids, mask)
        logits = m2.score(z.last_hidden_state, ids, mask, None, qidx, kv_bank=kvb, cand_rows=cand_rows).view(q_n, k)
        return logits, ops.argsort_desc(logits)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(); elapsed = time.perf_counter() - t0
    if rank == 0:
        n_tok = (args.image_size // 16) ** 2 + 1
        print(json.dumps({
            "metric": "query-candidate triplets scored/sec at K=100 (index-bank reuse, SURVEY 8(f)-1; not the headline metric)",
            "value": round(q_n * k * args.steps / elapsed, 1), "unit": "triplets/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{q_n} queries x {k} candidates per step drawn from a resident bank of {n_idx} index images "
                                   f"({n_tok} tokens): cached ViT tokens + 12-layer cross-attention K/V", "index_size": n_idx,
                       "bank_bytes": int(bank.numel() * 2 + sum(t.numel() for t in kvb) * 2),
                       "one_off_index_vit_s": round(t_vit, 3), "one_off_kv_bank_s": round(t_kv, 3)}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queries", type=int, default=16, help="queries per step per GPU")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="pixels", choices=["pixels", "bank"],
                    help="pixels: headline metric (every candidate encoded from pixels); bank: SURVEY 8(f)-1 real-dataset regime, "
                         "candidates drawn from a resident index bank with cached ViT tokens and cross-attention K/V")
    ap.add_argument("--index-size", type=int, default=2297, help="bank mode: number of index images (CIRR val: 2297)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    # (dry-run knobs for a box with ONE GPU: CIR_BENCH_DEVICE pins every rank to that device and CIR_BENCH_BACKEND=gloo
    #  replaces RCCL, which refuses two ranks on one device - exercises the launch / gather / timing logic only)
    dev_index = int(os.environ.get("CIR_BENCH_DEVICE", local_rank))
    backend = os.environ.get("CIR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from candidate_reranking_cir_amd import config, ops, synthetic, weights
    from candidate_reranking_cir_amd.blip_stage1 import BLIP_Retrieval
    from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR

    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    g, v = config.BertGeometry(), config.VitGeometry(image_size=args.image_size)
    m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    m2.load_state_dict(weights.synth_state_dict(weights.nlvr_param_spec(g, v), 0, "test"))
    m1 = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    m1.load_state_dict(weights.synth_state_dict(weights.retrieval_param_spec(g, v), 1, "test"))
    m2 = m2.to(dev).eval().set_compute_dtype(dt)
    m1 = m1.to(dev).eval().set_compute_dtype(dt)
    m2.engines(); m1.engines()

    q_n, k = args.queries, args.k
    if args.mode == "bank":
        return bank_mode(args, m2, m1, dev, dt, rank, world)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn((q_n + q_n * k, 3, args.image_size, args.image_size), generator=gen, device=dev, dtype=torch.float32).to(dt)
    ids = torch.stack([synthetic.caption_ids(rank * q_n + q, args.tokens) for q in range(q_n)]).to(dev)
    mask = torch.ones_like(ids)
    qidx = torch.arange(q_n, device=dev).repeat_interleave(k)
    # per-rank results of every step; ONE all-gather of scores + indices over RCCL at the end of the timed region
    # (the path has no other exchange step; 16 queries x 100 x (4 + 8) B per step and rank)
    n_buf = max(args.steps, args.warmup, 1)
    local_scores = torch.empty((n_buf, q_n, k), dtype=torch.float32, device=dev)
    local_order = torch.empty((n_buf, q_n, k), dtype=torch.int64, device=dev)
    # (gather buffers in the concatenated layout - rank-major along dim 0 - which every backend accepts)
    gathered_scores = torch.empty((world * n_buf, q_n, k), dtype=torch.float32, device=dev) if world > 1 else None
    gathered_order = torch.empty((world * n_buf, q_n, k), dtype=torch.int64, device=dev) if world > 1 else None

    def step(slot=0):
        toks = m2.img_embed16(images)                                   # reference images first, then candidates
        z = m1.z_t(toks[:q_n], ids, mask)
        logits = m2.score(z.last_hidden_state, ids, mask, toks[q_n:], qidx).view(q_n, k)
        order = ops.argsort_desc(logits)
        local_scores[slot].copy_(logits)
        local_order[slot].copy_(order)
        return logits, order

    def exchange():
        if world > 1 and backend == "nccl":
            dist.all_gather_into_tensor(gathered_scores, local_scores)
            dist.all_gather_into_tensor(gathered_order, local_order)
        elif world > 1:                                                 # dry run: the same gather through host memory
            for dst, src in ((gathered_scores, local_scores), (gathered_order, local_order)):
                host = torch.empty(dst.shape, dtype=dst.dtype)
                dist.all_gather_into_tensor(host, src.cpu())
                dst.copy_(host)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    exchange()                                                          # warm the communicator too
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    exchange()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = te.item()
    assert torch.isfinite(out[0]).all()

    # ---- instrumented step: HIP events around every GEMM launch on the launch stream -----------------
    ops.PROFILE_GEMM = []
    step()
    torch.cuda.synchronize()
    recs, ops.PROFILE_GEMM = ops.PROFILE_GEMM, None
    gemm_ms = sum(r[1].elapsed_time(r[2]) for r in recs)
    gemm_flop = sum(r[0] for r in recs)
    gemm_alg_bytes = sum(r[3] for r in recs)     # operands + outputs (+ residual, bias) read / written once
    t1 = time.perf_counter(); step(); torch.cuda.synchronize(); step_ms = (time.perf_counter() - t1) * 1e3

    if rank == 0:
        n_tok = (args.image_size // 16) ** 2 + 1
        alg = algorithmic_gflop(n_tok, args.tokens, k)
        traffic = None   # HBM bytes per GEMM launch from PMC passes collected with rocprofv3 on this same command
        tpath = os.path.join(ROOT, "profiles", "r1_gemm_traffic.json")
        if os.path.exists(tpath) and q_n == 16 and k == 100 and args.image_size == 224 and args.dtype == "bf16":
            traffic = round(json.load(open(tpath))["hbm_bytes_per_launch"])
        triplets = world * q_n * k * args.steps
        value = triplets / elapsed
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12
        line = {
            "metric": "query-candidate triplets scored/sec at K=100", "value": round(value, 2), "unit": "triplets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"CIRR-val-style K={k} re-rank from pixels, ViT-B/16 {args.image_size}px ({n_tok} tokens), "
                                   f"{args.tokens}-token captions, {q_n} queries x {k} candidates per step per GPU, random-init weights",
                       "queries_per_step_per_gpu": q_n, "k": k, "image_size": args.image_size, "tokens": args.tokens,
                       "parallelism": f"queries sharded over {world} GPU(s), all-gather of scores+indices"},
            "algorithmic_gflop_per_triplet": round(alg["per_triplet"], 2),
            "path_tflops": round(value * alg["per_triplet"] / 1e3, 1),
            "path_frac_of_mfma_peak": round(value * alg["per_triplet"] / 1e3 / (PEAK_TFLOPS[args.dtype] * world), 4),
            "roofline": {"bound": "mfma", "kernel": "cir::gemm256_kernel + cir::gemm_kernel (every GEMM launch of one step)", "achieved": round(achieved, 1),
                         "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS[args.dtype], 4),
                         "traffic": traffic, "algorithmic_bytes": round(gemm_alg_bytes / max(len(recs), 1)), "launches_per_step": len(recs), "avg_launch_us": round(gemm_ms * 1e3 / max(len(recs), 1), 2),
                         "gemm_share_of_step": round(gemm_ms / step_ms, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(usable_cpus())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
