"""Calibration only (never on the product path): what the vendor GEMM (hipBLASLt through torch.matmul) reaches on this
box for the path's shapes, next to cir_gemm_bias_act.  python tools/blas_probe.py"""
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
    shapes = [(318352, 2304, 768), (318352, 768, 768), (318352, 3072, 768), (318352, 768, 3072), (8192, 8192, 8192),
              (102400, 3072, 768), (315200, 3072, 768)]
    for m, n, k in shapes:
        a = (torch.randn((m, k), device="cuda") * 0.5).to(dt)
        w = (torch.randn((n, k), device="cuda") * 0.02).to(dt)
        out = torch.empty((m, n), dtype=dt, device="cuda")
        fl = 2.0 * m * n * k
        t_v = timeit(lambda: torch.matmul(a, w.t(), out=out))
        t_c = timeit(lambda: ops.gemm(a, w, None, out=out))
        print(f"M={m} N={n} K={k}: hipBLASLt {t_v:8.1f} us {fl / t_v / 1e6:7.1f} TF/s | cir {t_c:8.1f} us {fl / t_c / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
