"""A/B of cir_gemm_bias_act builds on the benchmark step's main shapes (GPU box only):
   CIR_LIB=candidate_reranking_cir_amd/libcirrank_x.so python tools/gemm_ab.py [warm_seconds]
Prints us / TF/s per shape and the calls-per-step weighted total (the GEMM share of a Q=16 step)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import ops

WARM = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6
f16, bf = torch.float16, torch.bfloat16
# (label, M, N, K, act, fp16-stream out + in-place fp16 residual, calls per step)
SHAPES = [("vit fc1+gelu", 334112, 3072, 768, 1, False, 12), ("cross K|V", 330960, 3072, 768, 0, False, 12),
          ("vit fc2+res", 334112, 768, 3072, 0, True, 12), ("vit qkv", 334112, 2304, 768, 0, False, 12),
          ("ffn fc1+gelu", 107520, 3072, 768, 1, False, 11), ("vit proj+res", 334112, 768, 768, 0, True, 12),
          ("ffn fc2+res", 107520, 768, 3072, 0, True, 11)]
total = 0.0
for label, m, n, k, act, res, calls in SHAPES:
    a = torch.randn((m, k), device="cuda").to(bf)
    w = (torch.randn((n, k), device="cuda") * 0.02).to(bf)
    b = torch.randn((n,), device="cuda")
    if res:
        x = torch.randn((m, n), device="cuda").to(f16)
        fn = lambda: ops.gemm(a, w, b, residual=x, out_dtype=f16, out=x)
    else:
        o = torch.empty((m, n), device="cuda", dtype=bf)
        fn = lambda: ops.gemm(a, w, b, act=act, out=o)
    t_end = time.time() + WARM
    while time.time() < t_end:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    total += us * calls / 1e3
    print(f"{label:14s} M={m:7d} N={n:5d} K={k:5d}: {us:8.1f} us {2.0 * m * n * k / us / 1e6:7.1f} TF/s", flush=True)
    del a, w
print(f"weighted GEMM ms per step: {total:.2f}")
