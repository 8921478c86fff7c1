"""LayerNorm folded into the GEMM behind it (cir_gemm_ln_bias_act) against LayerNorm pass + GEMM: accuracy vs fp64 and time at the
ViT shapes of the benchmark step.  python tools/lnfold_bench.py [images]"""
import sys

import torch

sys.path.insert(0, ".")
from candidate_reranking_cir_amd import ops  # noqa: E402


def ref64(x, w, b, g, be, eps, gelu):
    x64 = x.double()
    y = torch.nn.functional.layer_norm(x64, (x.shape[1],), g.double(), be.double(), eps)
    o = y @ w.double().t() + b.double()
    return torch.nn.functional.gelu(o) if gelu else o


def timeit(fn, n=10, warm_s=1.0):
    import time
    t_end = time.time() + warm_s                       # sustained load first: the clock the part holds under THIS kernel
    while time.time() < t_end:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    imgs = int(sys.argv[1]) if len(sys.argv) > 1 else 3392
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    k = 768
    for m, n, gelu in [(197, 2304, False), (1000, 320, True), (4096, 2304, False), (4096, 3072, True), (imgs * 197, 2304, False), (imgs * 197, 3072, True)]:
        x = torch.randn(m, k, device=dev) * 1.5 + 0.3
        x[:, 5] += 40.0                                                  # an outlier channel, as ViT residual streams have
        x[:, 77] -= 25.0
        x = x.half()
        w = torch.randn(n, k, device=dev) * 0.03
        b = torch.randn(n, device=dev) * 0.1
        g = 1.0 + 0.2 * torch.randn(k, device=dev)
        be = 0.1 * torch.randn(k, device=dev)
        eps = 1e-6
        wg, cs, bb = ops.ln_fold_pack(w, b, g, be)
        w16 = w.half()
        act = ops.ACT_GELU if gelu else ops.ACT_NONE
        out_f = ops.gemm_ln(x, wg, cs, bb, eps, act)
        _, xb = ops.layernorm(x, g, be, eps, want32=False, dtype16=torch.float16, stream_dtype=torch.float16)
        out_p = ops.gemm(xb, w16, b, act=act)
        line = f"M={m:7d} N={n:5d} gelu={int(gelu)}"
        if m <= 4096:
            r = ref64(x, w, b, g, be, eps, gelu)
            sc = r.abs().max().item()
            line += f"  err/max|ref|: folded {(out_f.double() - r).abs().max().item() / sc:.2e}  pass+gemm {(out_p.double() - r).abs().max().item() / sc:.2e}"
            line += f"  rms: folded {(out_f.double() - r).pow(2).mean().sqrt().item():.3e} pass+gemm {(out_p.double() - r).pow(2).mean().sqrt().item():.3e}"
        else:
            line += f"  folded vs pass+gemm max diff {(out_f.float() - out_p.float()).abs().max().item():.3e}"
            t_f = timeit(lambda: ops.gemm_ln(x, wg, cs, bb, eps, act, out=out_f))
            t_g = timeit(lambda: ops.gemm(xb, w16, b, act=act, out=out_p))
            t_l = timeit(lambda: ops.layernorm(x, g, be, eps, want32=False, dtype16=torch.float16, stream_dtype=torch.float16))
            t_r = timeit(lambda: ops.gemm(x, wg, bb, act=act, out=out_p))       # the plain kernel on the RAW rows (data-dependent power)
            line += f"  [plain kernel on raw rows {t_r:.0f} us]"
            fl = 2.0 * m * n * k
            line += f"  folded {t_f:.0f} us ({fl / t_f / 1e6:.0f} TF/s)  gemm {t_g:.0f} us ({fl / t_g / 1e6:.0f} TF/s) + layernorm {t_l:.0f} us"
        print(line, flush=True)
    # batch independence: the rows of a small call equal the same rows of a big one, bit for bit
    x = (torch.randn(5000, k, device=dev) * 2).half()
    w = torch.randn(2304, k, device=dev) * 0.03
    wg, cs, bb = ops.ln_fold_pack(w, torch.zeros(2304, device=dev), torch.ones(k, device=dev), torch.zeros(k, device=dev))
    big = ops.gemm_ln(x, wg, cs, bb, 1e-6)
    small = ops.gemm_ln(x[300:497], wg, cs, bb, 1e-6)
    print("rows 300-496 of a 5000-row call == a 197-row call:", bool(torch.equal(big[300:497], small)))


if __name__ == "__main__":
    main()
