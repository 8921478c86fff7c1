"""Stand-alone timings of cir_bmm on the training step's shapes: weight gradients (trans_a, split over row chunks) and the
stacked cross-attention products.  python tools/bmm_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from candidate_reranking_cir_amd import train_ops as T

dt = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6

# keep the clocks up
warm = torch.randn(8192, 8192, device="cuda", dtype=dt)
for _ in range(30): warm @ warm
for rows, n, k in ((8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (9232, 768, 768)):
    dy = torch.randn(rows, n, device="cuda", dtype=dt); x = torch.randn(rows, k, device="cuda", dtype=dt)
    for nb in (1, 2, 4, 8, 16, 32):
        if rows % nb: continue
        us = timeit(lambda: T.bmm(dy.view(nb, rows // nb, n), x.view(nb, rows // nb, k), True, False, out_dtype=torch.float32))
        print(f"wgrad rows {rows} N {n} K {k} nb {nb:2d}: {us:8.1f} us  {2 * rows * n * k / us / 1e6:7.1f} TFLOP/s")
# round 4: cir_wgrad (train_wgrad.hip) on the training step's weight-gradient shapes, over the split count
for rows, n, k in ((8192, 768, 768), (8192, 2304, 768), (9232, 1536, 768), (8192, 768, 1536), (16384, 3072, 768), (16384, 768, 3072)):
    dy = torch.randn(rows, n, device="cuda", dtype=dt); x = torch.randn(rows, k, device="cuda", dtype=dt)
    dw = torch.zeros(n, k, device="cuda")
    for sp in (0, 2, 4, 8, 12, 16, 24, 32):
        us = timeit(lambda: T.wgrad(dy, x, dw, splits=sp))
        print(f"cir_wgrad rows {rows} N {n} K {k} splits {sp:2d}: {us:8.1f} us  {2 * rows * n * k / us / 1e6:7.1f} TFLOP/s")
# one BertLayer's 13 weight gradients of the training step (R = 8192 rows per branch, 16 x 577 candidate rows) in ONE grouped launch
R, D = 8192, 768
layer = [(2 * R, D, 4 * D), (2 * R, 4 * D, D)] + [(R, 3 * D, D)] * 2 + [(R, D, D)] * 6 + [(9232, 2 * D, D)] * 2 + [(R, D, 2 * D)]
probs = [(torch.randn(rows, n, device="cuda", dtype=dt), torch.randn(rows, k, device="cuda", dtype=dt), torch.zeros(n, k, device="cuda")) for rows, n, k in layer]
fl = sum(2 * rows * n * k for rows, n, k in layer)
us = timeit(lambda: T.wgrad_grouped(probs))
print(f"cir_wgrad_grouped, one layer (13 problems, {fl / 1e9:.0f} GFLOP): {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
import ctypes
from candidate_reranking_cir_amd import lib as L
from candidate_reranking_cir_amd.ops import _DT, _stream
def grouped_with(split_fn):
    arr = (L.WgradDesc * len(probs))()
    for dsc, (dy, x, dw) in zip(arr, probs):
        rows, n = dy.shape; k = x.shape[1]
        dsc.dy, dsc.ldy, dsc.x, dsc.ldx, dsc.dw, dsc.ldw, dsc.rows, dsc.N, dsc.K, dsc.splits = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0), rows, n, k, split_fn(rows)
    return lambda: L.check(L.load().cir_wgrad_grouped(ctypes.addressof(arr), len(probs), _DT[dt], _stream()), "cir_wgrad_grouped")
for name, fn in (("unsplit units (FFN halved: 1224 units of 128 steps)", lambda r: 1 if r <= 9232 else 2), ("every unit halved (2448 units)", lambda r: 2 if r <= 9232 else 4),
                 ("in thirds (3672 units)", lambda r: 3 if r <= 9232 else 6)):
    us = timeit(grouped_with(fn))
    print(f"cir_wgrad_grouped, explicit splits - {name}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
us = timeit(lambda: [T.wgrad(*p) for p in probs])
print(f"the same 13 as single launches:                   {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
b, h, l, n_tok, d = 16, 12, 32, 577, 768
q = torch.randn(b * b * l, d, device="cuda", dtype=dt); kk = torch.randn(b * n_tok, d, device="cuda", dtype=dt)
heads = lambda x, g, r: x.view(g, r, h, 64).permute(0, 2, 1, 3)
ld = 584
s = torch.empty(b, h, b * l, ld, device="cuda")
us = timeit(lambda: T.bmm(heads(q, b, b * l), heads(kk, b, n_tok), False, True, out=s[..., :n_tok]))
print(f"cross QK^T (16 x 12 x 512 x 577 x 64): {us:8.1f} us  {2 * b * h * b * l * n_tok * 64 / us / 1e6:7.1f} TFLOP/s")
p = torch.randn(b, h, b * l, ld, device="cuda").to(dt)
ctx = torch.empty(b * b * l, d, device="cuda", dtype=dt)
us = timeit(lambda: T.bmm(p[..., :n_tok], heads(kk, b, n_tok), False, False, out=heads(ctx, b, b * l)))
print(f"cross P V   (16 x 12 x 512 x 64 x 577): {us:8.1f} us  {2 * b * h * b * l * n_tok * 64 / us / 1e6:7.1f} TFLOP/s")
dv = torch.empty(b * n_tok, d, device="cuda")
us = timeit(lambda: T.bmm(p[..., :n_tok], heads(ctx, b, b * l), True, False, out=heads(dv, b, n_tok)))
print(f"cross dV    (16 x 12 x 577 x 64 x 512): {us:8.1f} us  {2 * b * h * b * l * n_tok * 64 / us / 1e6:7.1f} TFLOP/s")
