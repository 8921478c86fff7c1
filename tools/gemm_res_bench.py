"""Epilogue variants of the ViT block GEMMs (plain, bias, fp32 out, fp32 residual) under start-stagger settings.
python tools/gemm_res_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import ops  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dt = torch.bfloat16
    m = 1616 * 197
    staggers = [int(x) for x in os.environ.get("STAGGERS", "0,400").split(",")]
    for (n, k) in ((768, 768), (768, 3072), (3072, 768), (2304, 768)):
        a = (torch.randn((m, k), device="cuda") * 0.5).to(dt)
        w = (torch.randn((n, k), device="cuda") * 0.02).to(dt)
        bias = torch.randn(n, device="cuda")
        x = torch.randn((m, n), device="cuda")
        o16 = torch.empty((m, n), dtype=dt, device="cuda")
        o32 = torch.empty((m, n), dtype=torch.float32, device="cuda")
        variants = {
            "plain16": lambda: ops.gemm(a, w, None, out=o16),
            "bias16": lambda: ops.gemm(a, w, bias, out=o16),
            "gelu16": lambda: ops.gemm(a, w, bias, act=ops.ACT_GELU, out=o16),
            "bias32": lambda: ops.gemm(a, w, bias, out_dtype=torch.float32, out=o32),
            "res32": lambda: ops.gemm(a, w, bias, residual=x, out_dtype=torch.float32, out=o32),
            "res32-inplace": lambda: ops.gemm(a, w, bias, residual=x, out_dtype=torch.float32, out=x),
        }
        for st in staggers:
            os.environ["CIR_GEMM_STAGGER"] = str(st)
            row = "  ".join(f"{name} {timeit(fn):7.1f}" for name, fn in variants.items())
            print(f"N={n} K={k} st={st}: {row}", flush=True)


if __name__ == "__main__":
    main()
