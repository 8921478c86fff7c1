#!/usr/bin/env python3
"""Headline benchmark: query-candidate triplets scored / s at K=100 on synthetic 224x224 images and
32-token captions (BASELINE.json metric).

One step = one batch of `--queries` CIRR-val-style queries, each with its own K=100 stage-I candidates plus the 5
non-reference members of its CIRR subset (validate_stage2.py:261-269; SURVEY 8(d) config 3: 4181 x 105 triplets),
taken from pixels and token ids already resident in HBM to sorted scores:
    ViT-B/16 over the Q*(K+5) candidate images and the Q reference images -> stage-I z_t per query ->
    two-branch fusion + cls_head per (query, candidate) -> per-query descending argsort of the K logits and of the 5
    subset logits [-> RCCL all-gather of the scores / indices when world_size > 1].
`--skip-rate r` gives a fraction r of the queries no positive in their top-K: the reference's skip rule fills their
row with -99999.99 and scores only their subset (validate_stage2.py:239/258); ranks then take blocks of
`distributed.balanced_order`.  Only scored (query, candidate) pairs count as triplets.
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

PEAK_TFLOPS = {"bf16": 2516.6, "f16": 2516.6, "mixed": 2516.6, "text32": 2516.6, "text32x3": 2516.6,  # dense 16-bit MFMA peak, 256 CU x 2.4 GHz x 4096 flop/clk/CU (MI355X_MICROARCH.md); text32: the 16-bit peak too (its algorithmic flops are priced as 16-bit work; the fp8 correction products are overhead, not credit)
               "exact": 157.3}                                  # f32-input MFMA: 256 CU x 2.4 GHz x 256 flop/clk/CU (same guide: 1/16 of the 16-bit rate)
_STREAMS = {"f16": ("f16", "f16"), "f32": ("f32", "f32"), "split": ("f32", "f16")}     # name -> (text-side storage, ViT storage)


def apply_precision(model, dtype: str, stream: str):
    """`dtype`: operand precision mode (BLIP_NLVR.set_precision: bf16 / f16 / mixed); `stream`: residual-stream storage -
    auto (the library's rule), f16, f32, or split (text side fp32, ViT fp16).  Returns the model."""
    model.set_precision(dtype)
    if dtype in ("exact", "text32", "text32x3"):     # the streams follow the mode (set_precision did it): fp32 everywhere / fp32 text side over an fp16 ViT
        return model
    if stream == "auto":
        return model.set_stream_dtype(None, vit=None)
    t, v = (torch.float16 if x == "f16" else torch.float32 for x in _STREAMS[stream])
    return model.set_stream_dtype(t, vit=v)


def stream_name(model) -> str:
    """What a run actually used: f16 / f32 / split."""
    t, v = ("f16" if d == torch.float16 else "f32" for d in (model.stream_dtype, model.vit_stream_dtype))
    return t if t == v else ("split" if (t, v) == ("f32", "f16") else f"text-{t}+vit-{v}")
DEFAULT_DTYPE = "f16"
PRECISION_NOTE = {"f16": "fp16 MFMA operands everywhere, fp32 accumulate (the library's default: holds the reference's rank order - DESIGN.md section 2)",
                  "bf16": "bf16 MFMA operands everywhere, fp32 accumulate",
                  "exact": "fp32 everywhere like the reference (model.float()): f32-input MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32), fp32 streams, erf GELU, "
                           "no algebraic folds - the mode that holds the reference's rank order (DESIGN.md section 2)",
                  "text32": "text side (self-attention, FFN, cls_head of both text encoders) on fp32 rows carried as split8 operands - one fp16 MFMA product + two block-scaled "
                            "fp8 correction products per Linear in one fp32 accumulator (~16 bits), fp32 stream, fp32 self-attention, erf GELU; fp16 ViT and cross-attention block - "
                            "what the factories set for real weights (DESIGN.md section 2)",
                  "text32x3": "round 5's form of text32: the text side's Linears as three fp16 MFMA products on [hi | lo | hi] rows (~20 bits); fp16 ViT and cross-attention block",
                  "mixed": "bf16 operands for the ViT and the cross-attention block, fp16 for text-side self-attention / FFN / cls_head; fp32 accumulate"}
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


def csrc_sha16() -> str:
    """Hash of the kernel sources (csrc/*.hip, *.hpp, the C header): ties an offline PMC summary to the build it measured."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "candidate_reranking_cir_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "cirrank.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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


def cpu_baseline(threads: int, per_query: int = 105):
    """Oracle (fp32 CPU port of the reference op sequence) on a bounded sample (~25 s of CPU work): 64 images through
    the ViT, 8 queries through stage I, 4 queries x 100 candidates through the fusion, every leg run TWICE with a fixed
    thread count and the faster repeat kept; composed to triplets/s with the same per-triplet accounting as the GPU
    metric (vit + fuse + (vit + s1) / candidates-per-query)."""
    from candidate_reranking_cir_amd import config, synthetic, weights
    from oracle import cir_oracle as O  # baseline leg only
    torch.set_num_threads(threads)
    g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
    sd2 = weights.synth_state_dict(weights.nlvr_param_spec(g, v), 0, "init")
    sd1 = weights.synth_state_dict(weights.retrieval_param_spec(g, v), 1, "init")
    ids = torch.stack([synthetic.caption_ids(q, 32) for q in range(8)])
    mask = torch.ones_like(ids)
    n_img, n_q, n_fq, k, reps = 64, 8, 4, 100, 2
    t_vit = t_s1 = t_fuse = float("inf")
    with torch.no_grad():
        imgs = synthetic.images(range(16), 224)
        O.img_embed(sd2, imgs[:2])                                  # warm the thread pool
        for _ in range(reps):
            t0 = time.perf_counter()
            feats = torch.cat([O.img_embed(sd2, imgs) for _ in range(n_img // 16)])      # batches of 16 like utils.py:32
            t_vit = min(t_vit, (time.perf_counter() - t0) / n_img)
            t0 = time.perf_counter()
            zs = [O.stage1_z_t(sd1, feats[q:q + 1], ids[q:q + 1], mask[q:q + 1]) for q in range(n_q)]
            t_s1 = min(t_s1, (time.perf_counter() - t0) / n_q)
            cand = torch.cat([feats, feats[: k - n_img]])
            t0 = time.perf_counter()
            for q in range(n_fq):
                O.img_txt_fusion_val(sd2, zs[q], cand, ids[q:q + 1], mask[q:q + 1])
            t_fuse = min(t_fuse, (time.perf_counter() - t0) / (n_fq * k))
    per_triplet = t_vit + t_fuse + (t_vit + t_s1) / per_query
    return {"value": round(1.0 / per_triplet, 3), "unit": "triplets/s", "cores": threads, "kind": "port",
            "sample": "fp32 oracle, best of %d repeats per leg: %d images ViT-B/16@224 in batches of 16 (%.3fs/img), %d queries stage-I "
                      "(%.3fs/query), %d queries x %d candidates fusion L=32 (%.4fs/cand); composed as vit+fuse+(vit+s1)/%d"
                      % (reps, n_img, t_vit, n_q, t_s1, n_fq, k, t_fuse, per_query)}


def rank_fidelity_block(scored, scored_x, active, k, ns, exact_ms=None, n_cand=0, recs_x=None):
    """This run's logits against the EXACT mode's on the same step (same pixels, ids, weights): per scored query the fraction of
    sorted top-K positions that hold the same candidate, Kendall's tau, the top-10 overlap and top-1 agreement; likewise for the
    5-member subsets.  The exact mode is fp32 on the f32-input MFMA and is pinned to the reference's own outputs (logits 2e-6, sorted
    order identical on 1599 of 1600 positions: tests/test_exact_gpu.py) - the on-device referee at sizes the CPU reference cannot reach."""
    import numpy as np
    from scipy.stats import kendalltau
    a, b = scored.float().cpu().numpy(), scored_x.float().cpu().numpy()
    rows, sub_rows, o = [], [], 0
    for act in active:
        if act:
            rows.append((a[o:o + k], b[o:o + k])); o += k
        if ns:
            sub_rows.append((a[o:o + ns], b[o:o + ns])); o += ns

    def stats(pairs, top):
        ex, tau, ov, t1 = [], [], [], []
        for x, y in pairs:
            ox, oy = np.argsort(-x, kind="stable"), np.argsort(-y, kind="stable")
            ex.append(float((ox == oy).mean())); tau.append(float(kendalltau(x, y).statistic))
            ov.append(len(set(ox[:top]) & set(oy[:top])) / float(top)); t1.append(float(ox[0] == oy[0]))
        return ex, tau, ov, t1
    out = {"referee": "this step in set_precision('exact') - fp32 tensors, f32-input MFMA, pinned to the reference's own outputs by tests/test_exact_gpu.py",
           "max_abs_dlogit": round(float(np.abs(a - b).max()), 6), "logit_sigma_per_query": None}
    if rows:
        ex, tau, ov, t1 = stats(rows, min(10, k))
        out.update({"queries": len(rows), "positions": len(rows) * k, "exact_positions": round(float(np.mean(ex)), 4),
                    "kendall_tau": round(float(np.mean(tau)), 5), "kendall_tau_worst_query": round(float(np.min(tau)), 5),
                    "top10_overlap": round(float(np.mean(ov)), 4), "top1_agree": round(float(np.mean(t1)), 4),
                    "logit_sigma_per_query": round(float(np.mean([y.std() for _, y in rows])), 5),
                    "median_adjacent_gap": round(float(np.median(np.concatenate([np.diff(np.sort(y)) for _, y in rows]))), 7)})
    if sub_rows:
        ex, tau, ov, t1 = stats(sub_rows, min(3, ns))
        out["subset"] = {"queries": len(sub_rows), "exact_positions": round(float(np.mean(ex)), 4), "kendall_tau": round(float(np.mean(tau)), 5),
                         "top3_overlap": round(float(np.mean(ov)), 4), "top1_agree": round(float(np.mean(t1)), 4)}
    if exact_ms is None:
        return out
    gf = sum(r[0] for r in recs_x); gms = sum(r[1].elapsed_time(r[2]) for r in recs_x)
    out["exact_mode"] = {"triplets_per_s": round(n_cand / (exact_ms * 1e-3), 1), "ms_per_step": round(exact_ms, 1),
                         "gemm_kernel": "cir::gemm_kernel<float,1> (v_mfma_f32_16x16x4_f32)", "gemm_tflops": round(gf / (gms * 1e-3) / 1e12, 1),
                         "gemm_frac_of_f32_mfma_peak": round(gf / (gms * 1e-3) / 1e12 / PEAK_TFLOPS["exact"], 4), "f32_mfma_peak_tflops": PEAK_TFLOPS["exact"],
                         "gemm_share_of_step": round(gms / exact_ms, 3)}
    return out


def device_info():
    """What the box reports about the GPU (print the clocks rocminfo reports and state the peak used)."""
    import re
    import subprocess
    info = {"name": None, "compute_units": None, "max_clock_mhz": None}
    try:
        txt = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=30).stdout
        for blk in txt.split("*******")[1:]:
            if "Device Type:             GPU" not in blk and not re.search(r"Device Type:\s+GPU", blk):
                continue
            m = re.search(r"Name:\s+(gfx\w+)", blk)
            c = re.search(r"Compute Unit:\s+(\d+)", blk)
            f = re.search(r"Max Clock Freq\. \(MHz\):\s+(\d+)", blk)
            info = {"name": m.group(1) if m else None, "compute_units": int(c.group(1)) if c else None,
                    "max_clock_mhz": int(f.group(1)) if f else None}
            break
    except Exception as exc:  # noqa: BLE001
        info["error"] = str(exc)[:80]
    return info


class ClockSampler:
    """Samples the shader clock and the socket power of GPU `index` through librocm_smi64 (rsmi_dev_gpu_clk_freq_get /
    rsmi_dev_current_socket_power_get - what `rocm-smi --showclocks --showpower` prints) every 0.2 s on a host thread
    while the timed region runs: the clock the part HOLDS under this load (it is power-capped well below rocminfo's max
    clock; the in-kernel clock reads up to ~10 % lower still, MI355X_MICROARCH.md 'DVFS give-back')."""

    def __init__(self, index: int):
        import ctypes
        self.sclk, self.power, self._stop, self.index, self.mapping = [], [], None, index, None
        self.lib = None
        try:
            class Freqs(ctypes.Structure):
                _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                            ("frequency", ctypes.c_uint64 * 33)]
            self.Freqs = Freqs
            lib = ctypes.CDLL("librocm_smi64.so")
            if lib.rsmi_init(ctypes.c_uint64(0)) == 0:
                self.lib = lib
                # ROCm-SMI enumerates every physical GPU and ignores HIP_VISIBLE_DEVICES: find the SMI index of the HIP device
                # that is being timed by its PCI address (BDFID = domain << 32 | bus << 8 | device << 3 | function)
                p = torch.cuda.get_device_properties(index)
                want = (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None))
                n = ctypes.c_uint32(0)
                if want[1] is not None and lib.rsmi_num_monitor_devices(ctypes.byref(n)) == 0:
                    for i in range(n.value):
                        bdf = ctypes.c_uint64(0)
                        if lib.rsmi_dev_pci_id_get(ctypes.c_uint32(i), ctypes.byref(bdf)) == 0:
                            got = ((bdf.value >> 32) & 0xffffffff, (bdf.value >> 8) & 0xff, (bdf.value >> 3) & 0x1f)
                            if got == want:
                                self.index, self.mapping = i, "hip device %d = rocm-smi device %d (pci %04x:%02x:%02x)" % (index, i, *got)
                                break
                if self.mapping is None:
                    self.mapping = "hip ordinal used as the rocm-smi index (no PCI match found)"
        except (OSError, AttributeError):
            pass

    def _sample(self):
        import ctypes
        f = self.Freqs()
        if self.lib.rsmi_dev_gpu_clk_freq_get(ctypes.c_uint32(self.index), ctypes.c_int(0), ctypes.byref(f)) == 0 and f.current < 33:
            self.sclk.append(f.frequency[f.current] / 1e6)
        p = ctypes.c_uint64(0)
        if self.lib.rsmi_dev_current_socket_power_get(ctypes.c_uint32(self.index), ctypes.byref(p)) == 0:
            self.power.append(p.value / 1e6)

    def _run(self):
        while not self._stop.is_set():
            self._sample()
            self._stop.wait(0.2)

    def __enter__(self):
        import threading
        if self.lib is not None:
            self._stop = threading.Event()
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        if self._stop is not None:
            self._stop.set()
            self._th.join()

    def summary(self):
        med = lambda v: round(sorted(v)[len(v) // 2], 1) if v else None
        return {"sclk_mhz_under_load_median": med(self.sclk), "socket_power_w_median": med(self.power), "samples": len(self.sclk),
                "smi_device": self.mapping,
                "source": "librocm_smi64 (rsmi_dev_gpu_clk_freq_get SYS / rsmi_dev_current_socket_power_get) every 0.2 s during the timed region"}


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
                       "bank_bytes": int(bank.numel() * 2 + sum(t.numel() for t in kvb if t is not None) * 2),
                       "one_off_index_vit_s": round(t_vit, 3), "one_off_kv_bank_s": round(t_kv, 3)}}), flush=True)


def latency_mode(args, m2, m1, dev, dt):
    """Single-query serving (the reference's own call: img_txt_fusion_val, blip_stage2.py:101-136 - ONE query against its K stage-I
    candidates, tokens precomputed): wall time per call with every call synchronised, issued launch by launch from Python against one
    captured HIP graph per shape (engine.ScoreGraph).  Reported separately from the headline metric."""
    from candidate_reranking_cir_amd import synthetic
    k, l = args.k, args.tokens
    n_tok = (args.image_size // 16) ** 2 + 1
    gen = torch.Generator(device=dev).manual_seed(3)
    cand = m2.img_embed16(torch.randn((k, 3, args.image_size, args.image_size), generator=gen, device=dev).to(dt))
    ref = m2.img_embed(torch.randn((1, 3, args.image_size, args.image_size), generator=gen, device=dev).to(dt))
    ids = synthetic.caption_ids(0, l).unsqueeze(0).to(dev)
    enc = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    z = m1.img_txt_fusion(ref, ref, enc, train=False, return_raw=True)
    z = z.last_hidden_state if hasattr(z, "last_hidden_state") else z
    sync = torch.cuda.synchronize

    def timed(n):
        ts = []
        for _ in range(n):
            sync(); t0 = time.perf_counter()
            out = m2.img_txt_fusion_val(z, cand, enc)
            sync(); ts.append(time.perf_counter() - t0)
        return sorted(ts), out

    res, outs = {}, {}
    for name, lim in (("launches_from_python", 0), ("hip_graph", max(512, k))):
        m2.enable_graphs(lim)
        timed(max(3, args.warmup))
        ts, outs[name] = timed(max(20, args.steps))
        res[name] = {"p50_ms": round(ts[len(ts) // 2] * 1e3, 3), "p90_ms": round(ts[int(len(ts) * 0.9)] * 1e3, 3), "min_ms": round(ts[0] * 1e3, 3)}
    m2.enable_graphs(0)
    # GPU time of the same call (events around a burst of un-synchronised direct calls): what the graph can approach
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync(); e0.record()
    for _ in range(10):
        m2.img_txt_fusion_val(z, cand, enc)
    e1.record(); sync()
    print(json.dumps({
        "metric": "single-query re-ranking latency, one query x K candidates, tokens precomputed (img_txt_fusion_val; not the headline metric)",
        "value": res["hip_graph"]["p50_ms"], "unit": "ms per query (p50, synchronised)", "n_gpus": 1, "steps": max(20, args.steps), "warmup": max(3, args.warmup),
        "ms_per_step": res["hip_graph"]["p50_ms"], "higher_is_better": False, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"1 query x {k} candidates, {l} caption tokens, {n_tok} image tokens ({args.image_size} px)"},
        "latency_ms": res, "pipelined_direct_ms_per_call": round(e0.elapsed_time(e1) / 10, 3),
        "speedup_p50": round(res["launches_from_python"]["p50_ms"] / res["hip_graph"]["p50_ms"], 2),
        "identical_logits": bool(torch.equal(outs["launches_from_python"], outs["hip_graph"]))}), flush=True)


def loop_mode(args, m2, m1, dev, dt):
    """The loop the reference actually has (validate_stage2.py:209-298, utils.py:43-55), timed end to end on a synthetic
    CIRR-val-sized split; reported separately from the headline metric (the index images are encoded ONCE, so a triplet
    here is fusion + stage-I / 105 only).  Legs:
      index   extract_index_features over the whole index (images resident in HBM)
      loop    generate_cirr_val_predictions (host batching, tokenised captions, skip rule, subset) + compute_cirr_val_metrics
      direct  the same device work with every host-side tensor prebuilt (z_t + score per batch): what the loop costs the host
      bank    the loop with the per-image cross-attention K/V bank
      level1  img_txt_fusion_val called once per query like validate_stage2.py:254 (Level-1 drop-in), ms per query"""
    import numpy as np
    from candidate_reranking_cir_amd import ops, synthetic, validate_stage2 as V
    q_n, k, ns, n_idx, qb = args.loop_queries, args.k, args.subset, args.index_size, args.query_batch
    rng = np.random.default_rng(11)
    gen = torch.Generator(device=dev).manual_seed(99)
    images = torch.cat([torch.randn((min(256, n_idx - i), 3, args.image_size, args.image_size), generator=gen, device=dev).to(dt)
                        for i in range(0, n_idx, 256)])
    cand = np.stack([rng.permutation(n_idx)[:k] for _ in range(q_n)])
    labels = np.zeros((q_n, k), dtype=bool)
    has = rng.random(q_n) >= args.skip_rate
    labels[np.arange(q_n)[has], np.minimum(rng.geometric(0.15, q_n) - 1, k - 1)[has]] = True
    group = np.stack([rng.permutation(n_idx)[:ns] for _ in range(q_n)]) if ns else None
    ids = torch.stack([synthetic.caption_ids(q, args.tokens) for q in range(q_n)])
    ds = V.RelativeValSet(ref_index=rng.integers(0, n_idx, q_n), cand_index=cand, labels=labels, input_ids=ids,
                          attention_mask=torch.ones_like(ids), group_index=group,
                          target_index=(group[:, 0] if ns else None))
    sync = torch.cuda.synchronize

    def timed(fn, reps=1):
        fn(); sync()                                                    # warm
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        sync()
        return (time.perf_counter() - t0) / reps, out

    t_index, bank = timed(lambda: V.extract_index_features(images, m2, batch_size=args.index_batch))
    def loop(kv=None):
        lg = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=qb, kv_bank=kv)
        return (lg, V.compute_cirr_val_metrics(lg[0], lg[1], ds)) if ns else (lg, V.compute_fiq_val_metrics(lg, ds))
    t_loop, (lg, metrics) = timed(loop)
    # direct: identical batches, every index tensor already on the device
    batches = []
    act = [q for q in range(q_n) if has[q] or ns]
    for s0 in range(0, len(act), qb):
        qs = act[s0:s0 + qb]
        rows, qidx = [], []
        for j, q in enumerate(qs):
            if has[q]:
                rows.append(cand[q]); qidx += [j] * k
            if ns:
                rows.append(group[q]); qidx += [j] * ns
        batches.append((ids[qs].to(dev), torch.ones((len(qs), args.tokens), dtype=torch.int64, device=dev),
                        torch.as_tensor(ds.ref_index[qs], device=dev), torch.as_tensor(np.concatenate(rows), device=dev),
                        torch.as_tensor(qidx, device=dev)))
    def direct():
        for bi, bm, br, bc, bq in batches:
            z = m1.z_t(ops.gather_rows(bank, br), bi, bm)
            out = m2.score(z.last_hidden_state, bi, bm, ops.gather_rows(bank, bc), bq)
        return out
    t_direct, _ = timed(direct)
    t_kv0 = time.perf_counter(); kvb = m2.build_kv_bank(bank); sync(); t_kv = time.perf_counter() - t_kv0
    t_bank, (lg_b, metrics_b) = timed(lambda: loop(kvb))
    # Level 1: one img_txt_fusion_val call per query, the way the reference's loop drives the model
    n1 = min(32, q_n)
    def level1():
        for q in range(n1):
            enc = {"input_ids": ids[q:q + 1], "attention_mask": torch.ones((1, args.tokens), dtype=torch.int64)}
            z = m1.img_txt_fusion(ops.gather_rows(bank, torch.as_tensor(ds.ref_index[q:q + 1], device=dev)), None, enc, train=False, return_raw=True)
            out = m2.img_txt_fusion_val(z, ops.gather_rows(bank, torch.as_tensor(cand[q], device=dev)), enc)
        return out
    t_l1, _ = timed(level1)
    n_trip = int(has.sum()) * k + q_n * ns
    n_tok = (args.image_size // 16) ** 2 + 1
    alg = algorithmic_gflop(n_tok, args.tokens, k)
    print(json.dumps({
        "metric": "query-candidate triplets scored/sec, scoring loop over a resident index (validate_stage2.py loop; not the headline metric)",
        "value": round(n_trip / t_loop, 1), "unit": "triplets/s", "n_gpus": 1, "higher_is_better": True, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"CIRR-val-style split: {q_n} queries x ({k} candidates + {ns} subset members) from an index of {n_idx} images "
                               f"({n_tok} tokens), skip rate {args.skip_rate:g}, query_batch {qb}", "residual_stream": args.stream_dtype},
        "legs_s": {"extract_index_features": round(t_index, 4), "loop_plus_metrics": round(t_loop, 4), "direct_engine": round(t_direct, 4),
                   "build_kv_bank_one_off": round(t_kv, 4), "loop_with_kv_bank": round(t_bank, 4)},
        "index_images_per_s": round(n_idx / t_index, 1),
        "loop_triplets_per_s": round(n_trip / t_loop, 1), "direct_triplets_per_s": round(n_trip / t_direct, 1),
        "host_overhead_frac": round(1.0 - t_direct / t_loop, 4),
        "kv_bank_loop_triplets_per_s": round(n_trip / t_bank, 1),
        "level1_ms_per_query": round(t_l1 / n1 * 1e3, 3), "level1_triplets_per_s": round(n1 * k / t_l1, 1),
        "fusion_tflops_loop": round(n_trip / t_loop * (alg["fuse"] + alg["s1"] / (k + ns)) / 1e3, 1),
        "recall": [round(x, 3) for x in metrics], "recall_kv_bank": [round(x, 3) for x in metrics_b],
        "logits_max_abs_diff_bank_vs_loop": float((lg[0] - lg_b[0]).abs().max()) if ns else float((lg - lg_b).abs().max())}), flush=True)


def train_gflop(b: int, l: int, n: int, d: int = 768, layers: int = 12, merge_from: int = 6):
    """Algorithmic GFLOP of one training step's fusion part as THIS library runs it (each target's cross-attention K/V projected
    once per step; the reference's loop projects them B times): (forward, backward).  Backward = dgrad + wgrad of every Linear
    (K/V projections: wgrad only - the image tokens are inputs) + the four adjoint products of each attention."""
    r = b * b * l
    lin = 6 * 2 * r * d * d + 16 * r * d * d                      # q k v o cq d + the shared FFN, per branch
    att = 4 * r * l * d + 4 * r * n * d                            # QK^T + PV, self and cross, per branch
    kv = 2 * 2 * (b * n) * d * d                                   # cross K, V of the B targets, per branch
    fwd = layers * 2 * (lin + att + kv) + (layers - merge_from) * 4 * r * d * d + 2 * b * b * (2 * d * d + 2 * d)
    bwd = layers * 2 * (2 * lin + 2 * att + kv) + (layers - merge_from) * 8 * r * d * d + 4 * b * b * (2 * d * d + 2 * d)
    return fwd / 1e9, bwd / 1e9


def train_cpu_baseline(threads: int, sd, l: int, n_tok: int, b: int = 8):
    """The fp32 oracle's training step (forward + torch-autograd backward, no optimizer) on the host cores, on a bounded sample:
    B = 8 (64 triplets, ~10 s) at the same caption / image token counts."""
    import torch.nn.functional as F
    from candidate_reranking_cir_amd import synthetic
    from oracle import cir_oracle as O
    torch.set_num_threads(threads)
    w = {k: t.detach().float().cpu().clone() for k, t in sd.items()}
    for k, t in w.items():
        if k.startswith(("text_encoder.", "cls_head.")) and t.is_floating_point():
            t.requires_grad_(True)
    gen = torch.Generator().manual_seed(3)
    z_t, feats = torch.randn((b, l, 768), generator=gen), torch.randn((b, n_tok, 768), generator=gen)
    ids = torch.stack([synthetic.caption_ids(q, l) for q in range(b)])
    t0 = time.perf_counter()
    logits = O.img_txt_fusion_train(w, z_t, feats, ids, torch.ones_like(ids))
    F.cross_entropy(logits, torch.arange(b)).backward()
    dt_s = time.perf_counter() - t0
    return {"value": round(b * b / dt_s, 3), "unit": "triplets/s", "cores": threads, "kind": "port",
            "sample": f"one training step (fp32 forward + autograd backward, no optimizer, ViT and stage I excluded) of oracle/cir_oracle.py at B = {b} "
                      f"({b * b} triplets, {l} caption tokens, {n_tok} image tokens): {dt_s:.1f} s"}


def train_mode(args, m2, m1, dev, dt):
    """One stage-II TRAINING step (stage2_train.py:176-218 with the default frozen ViT), timed end to end:
    reference + target images through the ViT (no grad), z_t from the frozen stage-I model, `img_txt_fusion` in .train() mode
    (dropout 0.1, B x B triplets), cross-entropy against arange(B), backward through the hand-written reverse pass, AdamW.
    Reported separately from the headline metric; unit: triplets (B*B per step) forward+backward per second."""
    import torch.nn.functional as F
    from candidate_reranking_cir_amd import synthetic
    from candidate_reranking_cir_amd.train import AdamW, _Lin2
    if os.environ.get("CIR_TRAIN_PAIRS") == "0":            # A/B: the branch twins' dense layers as two launches instead of one batched GEMM
        _Lin2.BATCHED = False
    b, l = args.train_batch, args.tokens
    gen = torch.Generator(device=dev).manual_seed(5)
    ref = torch.randn((b, 3, args.image_size, args.image_size), generator=gen, device=dev).to(dt)
    tgt = torch.randn((b, 3, args.image_size, args.image_size), generator=gen, device=dev).to(dt)
    ids = torch.stack([synthetic.caption_ids(q, l) for q in range(b)])
    enc = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    for n, p in m2.named_parameters():
        p.requires_grad_(args.img_tune or not n.startswith("visual_encoder."))   # blip_img_tune (stage2_train.py:87-92)
    m2.train()
    m1.train()                                              # stage2_train.py:165-166: both models; z_t is formed with the stage-I dropout on
    opt = AdamW([p for p in m2.parameters() if p.requires_grad], lr=2e-5, weight_decay=0.05)
    gt = torch.arange(b, device=dev)
    sync = torch.cuda.synchronize
    legs = {"vit": 0.0, "z_t": 0.0, "fusion_forward": 0.0, "backward": 0.0, "adamw": 0.0}
    host = dict(legs)

    def step(timed_legs=False):
        def mark(name, t0):
            if timed_legs:
                host[name] += time.perf_counter() - t0                 # the Python calls returned: launches enqueued
                sync(); legs[name] += time.perf_counter() - t0
            return time.perf_counter()
        t = time.perf_counter()
        with torch.no_grad():
            rf = m2.img_embed(ref).float()
            if not args.img_tune:
                tf = m2.img_embed(tgt).float()
        if args.img_tune:                                       # stage2_train.py:191-199: the target tokens carry a graph (the reference's
            tf = m2.img_embed(tgt).float()                      # own second graph, over the reference images, is never differentiated)
        t = mark("vit", t)
        with torch.no_grad():
            z = m1.img_txt_fusion(rf, rf, enc, train=False, return_raw=True)
            t = mark("z_t", t)
        opt.zero_grad()
        logits = m2.img_txt_fusion(z, tf, enc, train=True)
        loss = F.cross_entropy(logits, gt)
        t = mark("fusion_forward", t)
        loss.backward()
        t = mark("backward", t)
        opt.step()
        mark("adamw", t)
        return loss

    for _ in range(max(1, args.warmup)):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt_s = (time.perf_counter() - t0) / args.steps
    step(True)
    n_tok = (args.image_size // 16) ** 2 + 1
    gf, gb = train_gflop(b, l, n_tok)
    if args.img_tune:                                       # the backward leg then also holds the ViT's reverse pass over the B target images:
        d = 768                                             # 2 x its forward flops (dgrad + wgrad of qkv / proj / fc1 / fc2, the attention adjoint)
        gb += 2 * b * 12 * (2 * n_tok * d * (3 * d + d + 8 * d) + 4 * n_tok * n_tok * d) / 1e9
    fb_s = legs["fusion_forward"] + legs["backward"]
    cpu = None if args.no_cpu_baseline else train_cpu_baseline(usable_cpus(), m2.state_dict(), l, n_tok)
    print(json.dumps({
        "metric": "stage-II training step, query-target pairs (B x B) forward+backward per second (stage2_train.py loop; not the headline metric)",
        "value": round(b * b / dt_s, 1), "unit": "triplets/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt_s * 1e3, 2), "higher_is_better": True, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"batch {b} (B x B = {b * b} triplets), {l} caption tokens, {n_tok} image tokens ({args.image_size} px), {'ViT fine-tuned' if args.img_tune else 'ViT frozen'}, "
                               f"dropout 0.1, AdamW; fp32 residual stream" + ("; --blip-img-tune: DropPath 0.1, the image encoder's reverse pass is part of the backward leg and of its flops"
                                                                            if args.img_tune else "")},
        "legs_ms": {k: round(v * 1e3, 2) for k, v in legs.items()}, "legs_host_enqueue_ms": {k: round(v * 1e3, 2) for k, v in host.items()}, "loss": round(float(loss.detach()), 4),
        "algorithmic_gflop_per_step": {"fusion_forward": round(gf, 1), "backward": round(gb, 1)},
        "roofline": {"bound": "mfma", "achieved": round((gf + gb) / fb_s / 1e3, 1), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                     "frac": round((gf + gb) / fb_s / 1e3 / PEAK_TFLOPS[args.dtype], 4), "traffic": None,
                     "note": "fusion forward + backward legs (synchronised) over their algorithmic flops; see DESIGN section 9"},
        "cpu_baseline": cpu}), flush=True)


def launch_ranks(n: int) -> int:
    """Start one child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, rendezvous on 127.0.0.1)
    running this same command line, let them write to this process's stdout / stderr (rank 0 alone prints the JSON line) and
    return the first non-zero exit code (0 if every rank succeeded).  The launcher itself makes no GPU call."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, failed_at = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0) and rc == 0:
                rc, failed_at = p.returncode, time.time()
        if failed_at is not None and time.time() - failed_at > 20.0:      # a rank died: its peers would wait in a collective forever
            for p in procs:
                if p.poll() is None:
                    p.terminate()                                         # exactly the PIDs started above
            failed_at = time.time() + 1e9
        time.sleep(0.2)
    for p in procs:
        if p.returncode != 0 and rc == 0:
            rc = p.returncode
    if rc != 0:
        print(f"bench.py launcher: a rank exited with code {rc}", file=sys.stderr)
    return rc if rc > 0 else (1 if rc else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queries", type=int, default=64, help="queries per step per GPU (one stage-II batch; the ViT runs in chunks of <= 4096 images)")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--subset", type=int, default=5, help="CIRR subset members scored per query besides the K candidates (0: FashionIQ style)")
    ap.add_argument("--skip-rate", type=float, default=0.0, help="fraction of queries without a positive in their top-K (skip rule)")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--dtype", default=DEFAULT_DTYPE, choices=["bf16", "f16", "mixed", "text32", "text32x3", "exact"],
                    help="MFMA operand precision (fp32 accumulate in all): exact = fp32 everywhere on the f32-input MFMA (the reference's precision), f16, bf16, or mixed = bf16 for the ViT and the cross-attention "
                         "block, fp16 for the text-side self-attention / FFN / cls_head (DESIGN.md section 2: rank fidelity per mode)")
    ap.add_argument("--stream-dtype", default="auto", choices=["auto", "f16", "f32", "split"],
                    help="storage of the residual stream: auto (the library's rule), f16, f32, or split (text side fp32, ViT fp16); "
                         "sums are formed in fp32 either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-precision-table", action="store_true", help="skip the short re-runs at the other operand / stream precisions")
    ap.add_argument("--no-rank-fidelity", action="store_true", help="skip the exact-mode (fp32) re-run of the step that referees the run's rank order")
    ap.add_argument("--loop-queries", type=int, default=512, help="loop mode: queries of the synthetic split (CIRR val: 4181)")
    ap.add_argument("--query-batch", type=int, default=16, help="loop mode: queries per stage-II batch")
    ap.add_argument("--index-batch", type=int, default=256, help="loop mode: images per extract_index_features batch")
    ap.add_argument("--mode", default="pixels", choices=["pixels", "bank", "loop", "train", "latency"],
                    help="pixels: headline metric (every candidate encoded from pixels); bank: SURVEY 8(f)-1 real-dataset regime, "
                         "candidates drawn from a resident index bank with cached ViT tokens and cross-attention K/V")
    ap.add_argument("--img-tune", action="store_true", help="train mode: fine-tune the ViT too (stage2_train.py --blip-img-tune): target tokens with a "
                    "graph, the reverse pass continues through the image encoder, AdamW over both parameter buffers")
    ap.add_argument("--train-batch", type=int, default=16, help="train mode: B of the B x B training step (Instructions_*.md: --batch-size 16)")
    ap.add_argument("--index-size", type=int, default=2297, help="bank mode: number of index images (CIRR val: 2297)")
    args = ap.parse_args()

    # the box shows every host core but grants a cgroup quota: the default pool thrashes; N ranks of one node share that quota
    torch.set_num_threads(min(16, max(1, usable_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It has not touched the GPU (no HIP
        # call precedes this line) and never does; it starts one CHILD per GPU and relays rank 0's line - nothing is exec'ed.
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one process per GPU (drop the launcher and `--gpus N` starts its own ranks)")
    # (dry-run knobs for a box with ONE GPU: CIR_BENCH_DEVICE pins every rank to that device and CIR_BENCH_BACKEND=gloo
    #  replaces RCCL, which refuses two ranks on one device - exercises the launch / gather / timing logic only)
    dev_index = int(os.environ.get("CIR_BENCH_DEVICE", local_rank))
    backend = os.environ.get("CIR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    # (CIR_BENCH_INIT_PG=1: create the process group and run the exchange also with ONE rank - the RCCL communicator and the
    #  device-tensor all-gather then execute on a single-GPU box; tests/test_distributed_gpu.py)
    use_pg = world > 1 or os.environ.get("CIR_BENCH_INIT_PG") == "1"
    if use_pg:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from candidate_reranking_cir_amd import config, distributed as D, ops, synthetic, weights
    from candidate_reranking_cir_amd.blip_stage1 import BLIP_Retrieval
    from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR
    from candidate_reranking_cir_amd.validate_stage2 import SKIP_FILL

    if os.environ.get("CIR_TUNE"):                                      # A/B runs only: "knob=value,knob=value" -> cir_set_tuning
        from candidate_reranking_cir_amd import lib as _lib
        for kv in os.environ["CIR_TUNE"].split(","):
            _lib.set_tuning(*(int(x) for x in kv.split("=")))
    g, v = config.BertGeometry(), config.VitGeometry(image_size=args.image_size)
    m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    m2.load_state_dict(weights.synth_state_dict(weights.nlvr_param_spec(g, v), 0, "test"))
    m1 = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    m1.load_state_dict(weights.synth_state_dict(weights.retrieval_param_spec(g, v), 1, "test"))
    m2 = apply_precision(m2.to(dev).eval(), args.dtype, args.stream_dtype)
    m1 = apply_precision(m1.to(dev).eval(), args.dtype, args.stream_dtype)
    args.stream_dtype = stream_name(m2)                                              # what the run actually used
    dt = m2.token_dtype                                                              # 16-bit type of pixels / image tokens in HBM
    m2.engines(); m1.engines()
    if os.environ.get("CIR_VIT_LNFOLD") in ("0", "1", "2"):             # A/B runs only: LayerNorm folded into the ViT's qkv (1) / qkv + fc1 (2) GEMMs
        for v in (m2.engines()[0], m1.engines()[1]):                    # the two ViT engines (stage II's and stage I's)
            v.ln_fold = min(int(os.environ["CIR_VIT_LNFOLD"]), 2 if "qkv_f" in v.blocks[0] else 0)
    if os.environ.get("CIR_FOLD_CROSS") in ("0", "1"):                  # A/B runs only: the query-side fold of the cross-attention K / V projections
        m2.engines()[1].fold_cross_kv = os.environ["CIR_FOLD_CROSS"] == "1"

    q_n, k, ns = args.queries, args.k, args.subset
    if args.mode == "bank":
        return bank_mode(args, m2, m1, dev, dt, rank, world)
    if args.mode == "latency":
        if world != 1:
            raise SystemExit("--mode latency is a single-GPU measurement")
        return latency_mode(args, m2, m1, dev, dt)
    if args.mode == "train":
        if world != 1:
            raise SystemExit("--mode train is a single-GPU measurement")
        return train_mode(args, m2, m1, dev, dt)
    if args.mode == "loop":
        if world != 1:
            raise SystemExit("--mode loop is a single-GPU measurement")
        if args.skip_rate == 0.0:
            args.skip_rate = 0.006                                       # CIRR val: 0.6 % of the queries have no positive in the top-K
        return loop_mode(args, m2, m1, dev, dt)
    # ---- the step's queries: a global list of world * q_n queries, `skip_rate` of them without a positive in their top-K;
    #      ranks take contiguous blocks of distributed.balanced_order (equal numbers of scored queries per rank) ------------
    rng = torch.Generator(device="cpu").manual_seed(4242)
    active_all = (torch.rand(world * q_n, generator=rng) >= args.skip_rate).tolist()
    order_all = D.balanced_order(active_all)
    lo, hi, _ = D.shard_bounds(world * q_n, rank, world)
    mine = order_all[lo:hi]
    active = [active_all[q] for q in mine]
    per_q = [(k if a else 0) + ns for a in active]                       # candidates scored for each of my queries
    n_cand = sum(per_q)
    if n_cand == 0:
        raise SystemExit("nothing to score: --skip-rate 1 with --subset 0")
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn((q_n + n_cand, 3, args.image_size, args.image_size), generator=gen, device=dev, dtype=torch.float32).to(dt)
    ids = torch.stack([synthetic.caption_ids(q, args.tokens) for q in mine]).to(dev)
    mask = torch.ones_like(ids)
    qidx = torch.repeat_interleave(torch.arange(q_n), torch.tensor(per_q)).to(dev)
    # where each scored row lands: top-K logits (Q, K) pre-filled with the skip value, subset logits (Q, ns)
    top_src, top_dst, sub_src, o = [], [], [], 0
    for j, a in enumerate(active):
        if a:
            top_src += range(o, o + k); top_dst += range(j * k, (j + 1) * k); o += k
        sub_src += range(o, o + ns); o += ns
    top_src, top_dst, sub_src = (torch.tensor(t, dtype=torch.int64, device=dev) for t in (top_src, top_dst, sub_src))
    # per-rank results of every step; ONE all-gather of [scores | subset scores] and one of the indices over RCCL at the
    # end of the timed region (the path has no other exchange step)
    n_buf = max(args.steps, args.warmup, 1)
    w = k + ns
    local_scores = torch.empty((n_buf, q_n, w), dtype=torch.float32, device=dev)
    local_order = torch.empty((n_buf, q_n, w), dtype=torch.int64, device=dev)
    # (gather buffers in the concatenated layout - rank-major along dim 0 - which every backend accepts)
    gathered_scores = torch.empty((world * n_buf, q_n, w), dtype=torch.float32, device=dev) if use_pg else None
    gathered_order = torch.empty((world * n_buf, q_n, w), dtype=torch.int64, device=dev) if use_pg else None

    def step(slot=0):
        toks = m2.img_embed16(images)                                   # reference images first, then candidates
        z = m1.z_t(toks[:q_n], ids, mask)
        scored = m2.score(z.last_hidden_state, ids, mask, toks[q_n:], qidx)
        logits = torch.full((q_n * k,), SKIP_FILL, dtype=torch.float32, device=dev)   # skip rule (validate_stage2.py:258)
        logits[top_dst] = scored[top_src]
        logits = logits.view(q_n, k)
        local_scores[slot, :, :k] = logits
        local_order[slot, :, :k] = ops.argsort_desc(logits)
        if ns:
            sub = scored[sub_src].view(q_n, ns)
            local_scores[slot, :, k:] = sub
            local_order[slot, :, k:] = ops.argsort_desc(sub)
        return scored

    def exchange():
        if use_pg and backend == "nccl":
            dist.all_gather_into_tensor(gathered_scores, local_scores)
            dist.all_gather_into_tensor(gathered_order, local_order)
        elif use_pg:                                                    # dry run: the same gather through host memory
            for dst, src in ((gathered_scores, local_scores), (gathered_order, local_order)):
                host = torch.empty(dst.shape, dtype=dst.dtype)
                dist.all_gather_into_tensor(host, src.cpu())
                dst.copy_(host)

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if os.environ.get("CIR_BENCH_FAIL_RANK") == str(rank):              # test hook (tests/test_bench_gpu.py): this rank dies before the
        sys.exit(3)                                                     # exchange - its peers block in the collective, the launcher must end them
    exchange()                                                          # warm the communicator too
    fence()
    with ClockSampler(dev_index) as clocks:
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(i)
        exchange()
        fence()
        elapsed = time.perf_counter() - t0
    collective = None
    if use_pg:
        cdev = dev if backend == "nccl" else "cpu"
        mine_t = torch.tensor([elapsed, float(n_cand)], dtype=torch.float64, device=cdev)
        all_t = torch.empty((world * 2,), dtype=torch.float64, device=cdev)   # (concatenated layout: every backend accepts it)
        dist.all_gather_into_tensor(all_t, mine_t)                      # per-rank step time and work: load balance is visible per N
        all_t = all_t.view(world, 2)
        per_rank_ms = (all_t[:, 0] / args.steps * 1e3).tolist()
        elapsed = float(all_t[:, 0].max())                             # the job is as slow as its slowest rank
        total_cand = int(all_t[:, 1].sum())
        # the gathered matrix must hold this rank's block bit for bit, and every rank must hold the same matrix
        mine_blk = gathered_scores.view(world, n_buf, q_n, w)[rank]
        chk = torch.tensor([float(gathered_scores.double().sum())], dtype=torch.float64, device=cdev)
        chk_all = torch.empty((world,), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(chk_all, chk)
        collective = {"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend, "ranks": world,
                      "per_rank_ms_per_step": {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3),
                                               "mean": round(sum(per_rank_ms) / world, 3)},
                      "per_rank_triplets_per_step": [int(x) for x in all_t[:, 1].tolist()],
                      "gathered_bytes_per_rank": int(local_scores.numel() * 4 + local_order.numel() * 8),
                      "own_block_bit_identical": bool(torch.equal(mine_blk, local_scores)),
                      "gathered_checksum": float(chk_all[0]), "checksum_equal_on_all_ranks": bool((chk_all == chk_all[0]).all())}
    else:
        total_cand = n_cand
    assert torch.isfinite(out).all()

    # ---- instrumented step: HIP events around every GEMM / attention launch on the launch stream ----------------------
    ops.PROFILE_GEMM, ops.PROFILE_ATTN = [], []
    step()
    torch.cuda.synchronize()
    recs, arecs = ops.PROFILE_GEMM, ops.PROFILE_ATTN
    ops.PROFILE_GEMM = ops.PROFILE_ATTN = None
    by_kernel = {}
    for fl, e0, e1, nbytes, name in recs:
        d = by_kernel.setdefault(name, dict(flop=0.0, ms=0.0, launches=0, alg_bytes=0.0))
        d["flop"] += fl; d["ms"] += e0.elapsed_time(e1); d["launches"] += 1; d["alg_bytes"] += nbytes
    gemm_ms = sum(d["ms"] for d in by_kernel.values())
    gemm_flop = sum(d["flop"] for d in by_kernel.values())
    attn_flop = sum(r[0] for r in arecs)
    attn_ms = sum(r[1].elapsed_time(r[2]) for r in arecs)
    t1 = time.perf_counter(); step(); torch.cuda.synchronize(); step_ms = (time.perf_counter() - t1) * 1e3

    # ---- rank fidelity of THIS run against the exact (fp32) mode on the same step: thousands of sorted positions ---------
    fidelity = None
    if world == 1 and not args.no_rank_fidelity and args.dtype != "exact":
        ref_scored = step().clone()
        t32_scored = None
        if args.dtype not in ("text32", "text32x3"):                  # the mode the factories set for real weights, on the same step and referee
            apply_precision(m2, "text32", "auto"); apply_precision(m1, "text32", "auto")
            toks = m2.img_embed16(images.to(m2.token_dtype))
            z = m1.z_t(toks[:q_n], ids, mask)
            t32_scored = m2.score(z.last_hidden_state, ids, mask, toks[q_n:], qidx).clone()
            del toks, z
        apply_precision(m2, "exact", "auto"); apply_precision(m1, "exact", "auto")
        images_x = images.float()                                      # the same (16-bit-rounded) pixel values, as fp32 tensors
        def xstep():
            toks = m2.img_embed16(images_x)
            z = m1.z_t(toks[:q_n], ids, mask)
            return m2.score(z.last_hidden_state, ids, mask, toks[q_n:], qidx)
        xstep(); torch.cuda.synchronize()
        ops.PROFILE_GEMM = []
        tx = time.perf_counter(); scored_x = xstep(); torch.cuda.synchronize(); exact_ms = (time.perf_counter() - tx) * 1e3
        recs_x, ops.PROFILE_GEMM = ops.PROFILE_GEMM, None
        fidelity = rank_fidelity_block(ref_scored, scored_x, active, k, ns, exact_ms, n_cand, recs_x)
        if t32_scored is not None:
            fidelity["text32"] = rank_fidelity_block(t32_scored, scored_x, active, k, ns)
            fidelity["text32"]["mode"] = PRECISION_NOTE["text32"]
        del images_x, scored_x
        apply_precision(m2, args.dtype, args.stream_dtype); apply_precision(m1, args.dtype, args.stream_dtype)
        m2.engines(); m1.engines()
        torch.cuda.empty_cache()

    # ---- the speed / precision trade in the same record: 3 timed steps at each other (operand, stream) precision ---------
    precision = None
    if world == 1 and not args.no_precision_table and not args.no_cpu_baseline:
        precision = {f"{args.dtype}+{args.stream_dtype}_stream": round(total_cand * args.steps / elapsed, 1)}
        for od, sd_ in (("f16", "f16"), ("f16", "split"), ("f16", "f32"), ("mixed", "f16"), ("bf16", "f16"), ("bf16", "f32"), ("text32", "split"), ("text32x3", "split")):
            if (od, sd_) == (args.dtype, args.stream_dtype) or (od == args.dtype and od.startswith("text32")):
                continue
            n_steps = args.steps if od == "text32" else 3             # the real-weights mode at the headline's step count
            apply_precision(m2, od, sd_); apply_precision(m1, od, sd_)
            images_v = images.to(m2.token_dtype)
            def vstep():
                toks = m2.img_embed16(images_v)
                z = m1.z_t(toks[:q_n], ids, mask)
                return m2.score(z.last_hidden_state, ids, mask, toks[q_n:], qidx)
            vstep(); torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(n_steps):
                vstep()
            torch.cuda.synchronize()
            precision[f"{od}+{sd_}_stream"] = round(n_cand * n_steps / (time.perf_counter() - tv), 1)
            del images_v
        precision["note"] = (f"triplets/s of the same step at each operand mode + residual-stream storage (3 steps each, text32 - what the factories set for real weights - {args.steps} steps "
                             "like the headline; split = text side fp32, ViT fp16; text32 = text side on split8 rows (fp16 + 2 scaled-fp8 products), text32x3 = three fp16 products, both over the "
                             "fp16 ViT / cross block).  Rank fidelity of every mode against the reference's fp32 outputs: profiles/r6_precision_modes.json, DESIGN.md section 2")

    if rank == 0:
        n_tok = (args.image_size // 16) ** 2 + 1
        alg = algorithmic_gflop(n_tok, args.tokens, k)
        # algorithmic work of one step of THIS rank: every image through the ViT, every scored pair through the fusion,
        # every query through stage I (SURVEY 8(d) formulas; with all queries active = vit + fuse + (vit + s1) / (K + subset))
        alg_step = (q_n + n_cand) * alg["vit"] + n_cand * alg["fuse"] + q_n * alg["s1"]
        alg_per_triplet = alg_step / n_cand
        triplets = total_cand * args.steps
        value = triplets / elapsed
        peak = PEAK_TFLOPS[args.dtype]
        dom_name = max(by_kernel, key=lambda n: by_kernel[n]["ms"])
        dom = by_kernel[dom_name]
        dom_tf = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
        all_tf = gemm_flop / (gemm_ms * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel: offline PMC passes of this same command (tools/collect_profiles.sh); the
        # summary names the workload and the hash of the kernel sources it was measured on - any other build reports null
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r6_pmc_summary.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            same = (tj.get("csrc_sha16") == csrc_sha16() and tj.get("queries") == q_n and tj.get("k") == k and tj.get("subset") == ns
                    and tj.get("image_size") == args.image_size and tj.get("dtype") == args.dtype and args.skip_rate == 0
                    and tj.get("residual_stream") == args.stream_dtype)
            ent = tj.get("by_kernel", {}).get(dom_name) if same else None
            if ent:
                traffic = round(ent["hbm_bytes_per_launch"])
                traffic_src = ("profiles/r6_pmc_summary.json: offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on these "
                               "kernel sources (csrc_sha16 matches; not this run)")
        line = {
            "metric": "query-candidate triplets scored/sec at K=100", "value": round(value, 2), "unit": "triplets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"CIRR-val-style K={k} re-rank from pixels (BASELINE configs[2]): ViT-B/16 {args.image_size}px ({n_tok} tokens), "
                                   f"{args.tokens}-token captions, {q_n} queries x ({k} candidates + {ns} subset members) per step per GPU, "
                                   f"skip rate {args.skip_rate:g}, random-init weights",
                       "queries_per_step_per_gpu": q_n, "k": k, "subset": ns, "skip_rate": args.skip_rate, "image_size": args.image_size,
                       "tokens": args.tokens, "triplets_per_step_rank0": n_cand, "precision_mode": PRECISION_NOTE[args.dtype],
                       "residual_stream": args.stream_dtype,
                       "parallelism": f"queries sharded over {world} GPU(s) (balanced_order blocks), all-gather of scores+indices"},
            "algorithmic_gflop_per_triplet": round(alg_per_triplet, 2),
            "executed_gflop_per_triplet": round((gemm_flop + attn_flop) / 1e9 / n_cand, 2),
            "path_tflops": round(value * alg_per_triplet / 1e3, 1),
            "path_frac_of_mfma_peak": round(value * alg_per_triplet / 1e3 / (peak * world), 4),
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": round(dom_tf, 1), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(dom_tf / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": round(dom["alg_bytes"] / dom["launches"]), "launches_per_step": dom["launches"],
                         "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2), "share_of_step": round(dom["ms"] / step_ms, 3),
                         "all_gemm_kernels": {"achieved": round(all_tf, 1), "frac": round(all_tf / peak, 4), "launches_per_step": len(recs),
                                              "avg_launch_us": round(gemm_ms * 1e3 / max(len(recs), 1), 2), "share_of_step": round(gemm_ms / step_ms, 3),
                                              "by_kernel": {n: {"tflops": round(d["flop"] / (d["ms"] * 1e-3) / 1e12, 1), "ms_per_step": round(d["ms"], 2),
                                                                "launches": d["launches"]} for n, d in sorted(by_kernel.items())}},
                         "attention_kernels": {"tflops": round(attn_flop / (attn_ms * 1e-3) / 1e12, 1), "ms_per_step": round(attn_ms, 2), "launches": len(arecs)},
                         "peak_definition": ("256 CU x 2.4 GHz x 256 flop/clk/CU f32-input MFMA (MI355X_MICROARCH.md)" if args.dtype == "exact" else
                                             "256 CU x 2.4 GHz x 4096 flop/clk/CU dense bf16/f16 MFMA (MI355X_MICROARCH.md)")},
            "device": dict(device_info(), **clocks.summary()),
        }
        if collective is not None:
            line["collective"] = collective
        if fidelity is not None:
            line["rank_fidelity"] = fidelity
        if precision is not None:
            line["precision_table"] = precision
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(usable_cpus(), k + ns)
        print(json.dumps(line), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
