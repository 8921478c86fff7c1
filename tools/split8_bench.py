"""Timing of the text-side Linear shapes of one benchmark step (2 x 6720 x 32 = 430 080 rows) in the three arithmetics of the text32 mode:
one fp16 product (the default mode's GEMM), three fp16 products on [hi | lo | hi] rows (round 5), fp16 + two block-scaled fp8 products on
split8 rows (round 6).  HIP events around 5 back-to-back launches after 2 warm-up launches."""
import json
import sys

import torch

sys.path.insert(0, ".")
from candidate_reranking_cir_amd import ops  # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 430080
    g = torch.Generator(device="cuda").manual_seed(1)
    rows = []
    for name, n, k, res, act in (("qkv", 2304, 768, False, ops.ACT_NONE), ("self-out / cross-q", 768, 768, True, ops.ACT_NONE),
                                 ("fc1 + GELU", 3072, 768, False, ops.ACT_GELU), ("fc2", 768, 3072, True, ops.ACT_NONE)):
        a32 = torch.randn((m, k), generator=g, device="cuda")
        w32 = torch.randn((n, k), generator=g, device="cuda") * 0.03
        b = torch.randn((n,), generator=g, device="cuda") * 0.1
        r32 = torch.randn((m, n), generator=g, device="cuda") if res else None
        a16, w16 = a32.half(), w32.half()
        r16 = r32.half() if res else None
        t1 = timed(lambda: ops.gemm(a16, w16, b, residual=r16, act=act, out_dtype=torch.float16 if res else None))
        w3 = ops.split_weight(w32.clone())
        s3 = ops.split16(a32)
        t3 = timed(lambda: ops.gemm(s3.cat, w3._split3, b, residual=r32, out_dtype=torch.float32))
        t3s = timed(lambda: ops.split16(a32, act)) if act != ops.ACT_NONE else 0.0
        del s3
        w8 = ops.split_weight8(w32.clone())
        s8 = ops.split8(a32)
        t8 = timed(lambda: ops.gemm(s8, w8, b, residual=r32, act=act))
        fl = 2.0 * m * n * k
        rows.append(dict(shape=name, m=m, n=n, k=k, fp16_ms=round(t1, 3), split3_ms=round(t3, 3), split3_gelu_pass_ms=round(t3s, 3), split8_ms=round(t8, 3),
                         fp16_tflops=round(fl / t1 / 1e9, 1), split8_useful_tflops=round(fl / t8 / 1e9, 1), split8_executed_fp16_equiv_tflops=round(2 * fl / t8 / 1e9, 1)))
        print(rows[-1], flush=True)
        del a32, w32, a16, w16, r32, r16, s8
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
