"""Folded cross-attention (cir_cross_attention_folded) against the projected path (K|V GEMM + cir_attention) at the benchmark step's shape:
6720 candidates x 197 image tokens, 32 caption tokens, both branches (one fusion layer).  python tools/fold_bench.py [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from candidate_reranking_cir_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6720
L, N, D = 32, int(os.environ.get('FOLD_N', '197')), 768
dt = torch.float16 if len(sys.argv) < 3 or sys.argv[2] != "bf16" else torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda shape, s: (torch.randn(shape, generator=g, device="cuda") * s).to(dt)
q, x = r((2, T * L, D), 1.0), r((T, N, D), 1.0)
wk, wv = r((2, D, D), 0.03), r((2, D, D), 0.03)
bk, bv = torch.randn((2, D), device="cuda") * 0.5, torch.randn((2, D), device="cuda") * 0.5
wkt, wvp = ops.fold_pack_key(wk), ops.fold_pack_value(wv)
wkv, bkv = torch.cat([wk[0], wv[0], wk[1], wv[1]]), torch.cat([bk[0], bv[0], bk[1], bv[1]])
out = torch.empty((T, L, 2, D), dtype=dt, device="cuda")
o2 = torch.empty((T, L, 2, D), dtype=dt, device="cuda")


def folded():
    ops.cross_attention_folded(q, x, wkt, wvp, bv, out, L, 0.125)


def projected():
    kv = ops.gemm(x.view(T * N, D), wkv, bkv).view(T, N, 4, D)
    ops.attention(q.view(2, T, L, D).permute(1, 0, 2, 3), kv[:, :, 0::2].permute(0, 2, 1, 3), kv[:, :, 1::2].permute(0, 2, 1, 3), o2.permute(0, 2, 1, 3), 0.125)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    # warm the part (power / clocks) like the benchmark step does
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if len(sys.argv) > 3 and sys.argv[3] == "quick":       # profiler runs: a few launches of the folded kernel only
    for _ in range(5):
        folded()
    torch.cuda.synchronize()
    sys.exit(0)
if os.environ.get("FOLD_ONLY"):
    print(f"T {T} {dt}: folded {timeit(folded) * 1e3:.0f} us", flush=True)
    sys.exit(0)
for rep in range(2):
    tf, tp = timeit(folded), timeit(projected)
    fl_f = T * 2 * 2 * (384 * 64 * 768 * 2 + 384 * 768 * N * 2)
    fl_p = 2 * T * N * D * 4 * D + 4 * T * 2 * L * N * D
    print(f"T {T} {dt}: folded {tf * 1e3:.0f} us ({fl_f / tf / 1e9:.0f} TFLOP/s of its {fl_f / 1e9 / T:.0f} MFLOP per candidate)   projected {tp * 1e3:.0f} us "
          f"({fl_p / tp / 1e9:.0f} TFLOP/s of its {fl_p / 1e9 / T:.0f} MFLOP)   ratio {tf / tp:.3f}", flush=True)
print("max |folded - projected|", (out.float() - o2.float()).abs().max().item())
