"""Training-mode operators (include/cirrank.h) against plain fp32 PyTorch on the MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import helpers as H

pytestmark = pytest.mark.gpu
BF, HF = torch.bfloat16, torch.float16


@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from candidate_reranking_cir_amd import train_ops
    return train_ops


def _r(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


@pytest.mark.parametrize("shape", [(1, 1), (33, 70), (3, 197, 64), (2, 64, 300)])
def test_transpose16(T, shape):
    x = _r(shape, 1, dtype=BF)
    assert torch.equal(T.transpose16(x), x.transpose(-1, -2).contiguous())
    xs = _r(shape[:-1] + (shape[-1] + 8,), 2, dtype=HF)[..., :shape[-1]]          # row stride > cols
    assert torch.equal(T.transpose16(xs), xs.transpose(-1, -2).contiguous())


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("dtype", [BF, torch.float32])
def test_bmm(T, ta, tb, dtype):
    b, m, n, k = 5, 32, 197, 64
    a = _r((b, k, m) if ta else (b, m, k), 3, dtype=dtype)
    w = _r((b, n, k) if tb else (b, k, n), 4, dtype=dtype)
    ref = (a.float().transpose(1, 2) if ta else a.float()) @ (w.float().transpose(1, 2) if tb else w.float())
    out = T.bmm(a, w, ta, tb, out_dtype=torch.float32, alpha=0.5)
    torch.testing.assert_close(out, 0.5 * ref, atol=1e-3, rtol=1e-4)
    out2 = T.bmm(a, w, ta, tb, out=out.clone(), accumulate=True)
    torch.testing.assert_close(out2, 1.5 * ref, atol=2e-3, rtol=1e-4)
    # strided views (heads inside a packed qkv tensor)
    big = _r((b, 40, 3 * 64), 5, dtype=dtype)
    q, kk = big[:, :32, :64], big[:, :, 64:128]
    s = T.bmm(q, kk, False, True, out_dtype=torch.float32)
    torch.testing.assert_close(s, q.float() @ kk.float().transpose(1, 2), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("dtype", [BF, HF])
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (70, 33, 45), (512, 577, 64), (577, 64, 512), (64, 64, 100)])
def test_bmm_mfma_two_level_batches(T, ta, tb, dtype, m, n, k):
    """The MFMA kernel at ragged extents (odd leading dimensions -> 2-byte loads; aligned ones -> 16-byte loads with masked
    edges), both storage orders of both operands, two batch levels with a broadcast stride, 16-bit and fp32 outputs."""
    b1, b2 = 3, 2
    a = _r((b1, b2, k, m) if ta else (b1, b2, m, k), 7, dtype=dtype)
    w = _r((b1, 1, n, k) if tb else (b1, 1, k, n), 8, dtype=dtype).expand(b1, b2, -1, -1)          # level-2 stride 0
    ref = (a.float().transpose(2, 3) if ta else a.float()) @ (w.float().transpose(2, 3) if tb else w.float())
    out = T.bmm(a, w, ta, tb, out_dtype=torch.float32)
    torch.testing.assert_close(out, ref, atol=1e-3 * max(1.0, k ** 0.5 / 8), rtol=1e-4)
    out16 = T.bmm(a, w, ta, tb)
    assert out16.dtype == dtype
    torch.testing.assert_close(out16.float(), ref, atol=(0.06 if dtype == BF else 0.008) * max(1.0, k ** 0.5 / 4), rtol=2e-2)
    # padded leading dimensions (what the attention score tensors use): same numbers through the 16-byte path
    pad = lambda t: torch.nn.functional.pad(t.contiguous(), (0, (-t.shape[-1]) % 8 + 8))[..., :t.shape[-1]]
    out_p = T.bmm(pad(a), pad(w), ta, tb, out_dtype=torch.float32)
    torch.testing.assert_close(out_p, out, atol=1e-5, rtol=1e-5)


def test_bmm_split_k_weight_gradient(T):
    """dW = dy^T x as cir_bmm sees it in train.py: trans_a = 1, a batch over row chunks into partial sums, then column sums."""
    rows, n, k, nb = 16 * 37, 96, 128, 16
    dy, x = _r((rows, n), 9, dtype=BF), _r((rows, k), 10, dtype=BF)
    part = T.bmm(dy.view(nb, rows // nb, n), x.view(nb, rows // nb, k), True, False, out_dtype=torch.float32)
    dw = torch.zeros((n * k,), device="cuda")
    T.colsum(part.view(nb, n * k), dw)
    torch.testing.assert_close(dw.view(n, k), dy.float().t() @ x.float(), atol=2e-3, rtol=1e-4)
    # round 4: the chunks add their partial sums straight into ONE dW (stride-0 batch view, atomic accumulate), strided operands
    wide = _r((rows, n + k + 8), 11, dtype=HF)
    dy2, x2 = wide[:, :n], wide[:, n:n + k]
    dw2 = torch.full((n, k), 3.0, device="cuda")
    T.bmm(dy2.unflatten(0, (nb, rows // nb)), x2.unflatten(0, (nb, rows // nb)), True, False, out=dw2.unsqueeze(0).expand(nb, n, k), accumulate="atomic")
    torch.testing.assert_close(dw2 - 3.0, dy2.float().t() @ x2.float(), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
@pytest.mark.parametrize("rows,n,k,splits", [(64, 128, 128, 0), (336, 256, 384, 0), (9232, 1536, 768, 0), (1024, 768, 128, 3), (40, 128, 256, 0),
                                              (4096, 128, 128, 64)])
def test_wgrad_kernel(T, dtype, rows, n, k, splits):
    """cir_wgrad (LDS-DMA staging, transposing LDS reads, atomic split accumulation) against fp32 torch: dW += dy^T x with both operands
    read as stored - row counts with and without a tail below 64 (577-token candidates: 9232 = 144 * 64 + 16), strided operand views
    (column slices of wider buffers), explicit and automatic splits, accumulation into a non-zero dW."""
    wide_y, wide_x = _r((rows, n + 128), 51, dtype=dtype), _r((rows, k + 256), 52, dtype=dtype)
    dy, x = wide_y[:, 128:], wide_x[:, 128:128 + k]
    dw = torch.full((n, k), 0.25, device="cuda")
    T.wgrad(dy, x, dw, splits=splits)
    ref = dy.float().t() @ x.float()
    err = ((dw - 0.25) - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err
    # integer-valued operands: every partial sum is exact in fp32, so the result does not depend on the (unordered) atomic additions
    gi = torch.Generator().manual_seed(rows + n)
    yi = torch.randint(-3, 4, (rows, n), generator=gi).to(dtype).cuda()
    xi = torch.randint(-3, 4, (rows, k), generator=gi).to(dtype).cuda()
    dwi = torch.zeros((n, k), device="cuda")
    T.wgrad(yi, xi, dwi, splits=splits)
    assert torch.equal(dwi, yi.float().t() @ xi.float())
    with pytest.raises(RuntimeError):
        T.wgrad(yi[:, :64], xi, torch.zeros((64, k), device="cuda"))              # N % 128


@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
def test_wgrad_grouped_layer_shapes(T, dtype):
    """cir_wgrad_grouped on a scaled-down BertLayer's 13 weight gradients (+ 5 more: two launches): unequal row counts (the FFN's stacked rows
    are split to the attention projections' length, a 9232-like count with a tail below 64), shared dy between two problems, exact
    integer-valued operands (no dependence on the order of any atomic addition), accumulation into non-zero dW."""
    r, d = 448, 128
    gi = torch.Generator().manual_seed(7)
    mk = lambda rows, cols: torch.randint(-3, 4, (rows, cols), generator=gi).to(dtype).cuda()
    shapes = [(2 * r, d, 4 * d), (2 * r, 4 * d, d), (r, 3 * d, d), (r, 3 * d, d), (r, d, d), (r, d, d), (r, d, d), (r, d, d), (592, 2 * d, d), (592, 2 * d, d),
              (r, d, 2 * d), (r, d, d), (r, d, d), (64, 128, 128), (40, 128, 128), (r, 2 * d, 3 * d), (r, d, d), (1024, 256, 128)]
    probs, refs = [], []
    shared = mk(r, d)
    for i, (rows, n, k) in enumerate(shapes):
        dy = shared if (rows, n) == (r, d) and i in (11, 12) else mk(rows, n)
        x = mk(rows, k)
        dw = torch.full((n, k), float(i), device="cuda")
        probs.append((dy, x, dw))
        refs.append(dy.float().t() @ x.float() + float(i))
    T.wgrad_grouped(probs)
    for (dy, x, dw), ref in zip(probs, refs):
        assert torch.equal(dw, ref), (dy.shape, x.shape)


@pytest.mark.parametrize("cols", [197, 900], ids=["row-in-registers", "long-row-loops"])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_softmax_dropout_fwd_bwd(T, p_drop, cols):
    groups, rpm = 6, 32
    rows = groups * rpm
    s = _r((rows, cols), 6, 3.0)
    mask = torch.zeros((groups, cols), device="cuda")
    mask[:, cols - 7:] = -10000.0
    p, pd = T.softmax_dropout(s, mask, rpm, 0.125, p_drop, 1234, BF)
    ref = torch.softmax(s * 0.125 + mask.repeat_interleave(rpm, 0), -1)
    torch.testing.assert_close(p.float(), ref, atol=4e-3, rtol=0)
    keep = (pd != 0) | (p == 0)
    if p_drop == 0:
        assert torch.equal(p, pd)
    else:
        frac = 1.0 - keep.float().mean().item()
        assert abs(frac - p_drop) < 0.01
        torch.testing.assert_close(pd.float()[keep], (p.float() / (1 - p_drop))[keep], atol=8e-3, rtol=1e-2)
        p2, pd2 = T.softmax_dropout(s, mask, rpm, 0.125, p_drop, 1234, BF)
        assert torch.equal(pd, pd2)                                           # same seed, same mask
        assert not torch.equal(pd, T.softmax_dropout(s, mask, rpm, 0.125, p_drop, 99, BF)[1])
    # backward against autograd of softmax + the SAME mask
    dpd = _r((rows, cols), 7)
    ds = T.softmax_dropout_bwd(p, dpd, 0.125, p_drop, 1234)
    sx = s.clone().requires_grad_(True)
    pr = torch.softmax(sx * 0.125 + mask.repeat_interleave(rpm, 0), -1)
    m = keep.float() / (1 - p_drop) if p_drop > 0 else torch.ones_like(pr)
    (pr * m * dpd).sum().backward()
    torch.testing.assert_close(ds.float(), sx.grad, atol=3e-3, rtol=2e-2)


def test_softmax_padded_rows(T):
    """Score rows padded to a multiple of 8 (what train.py allocates for 577 / 197 keys): only the first `cols` entries are read
    and written; dropout indices follow (row, col), not the padded offset."""
    rows, cols, ld = 70, 197, 200
    s = _r((rows, ld), 21, 3.0)
    p, pd = T.softmax_dropout(s, None, 1, 0.125, 0.1, 77, HF, cols=cols)
    pc, pdc = T.softmax_dropout(s[:, :cols].contiguous(), None, 1, 0.125, 0.1, 77, HF)
    assert torch.equal(p[:, :cols], pc) and torch.equal(pd[:, :cols], pdc)
    dpd = _r((rows, ld), 22)
    ds = T.softmax_dropout_bwd(p, dpd, 0.125, 0.1, 77, cols=cols)
    dsc = T.softmax_dropout_bwd(pc, dpd[:, :cols].contiguous(), 0.125, 0.1, 77)
    assert torch.equal(ds[:, :cols], dsc)


@pytest.mark.parametrize("rows,cols", [(203, 768), (5, 128), (64, 1024)])
def test_layernorm_bwd(T, rows, cols):
    x, dy = _r((rows, cols), 8, 2.0) + 0.3, _r((rows, cols), 9)
    g, b = _r((cols,), 10) * 0.1 + 1.0, _r((cols,), 11)
    dg, db = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
    dx = T.layernorm_bwd(x, g, dy, dg, db, 1e-12)
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (F.layer_norm(xr, (cols,), gr, br, 1e-12) * dy).sum().backward()
    torch.testing.assert_close(dx, xr.grad, atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(dg, gr.grad, atol=2e-3, rtol=1e-4)
    torch.testing.assert_close(db, br.grad, atol=2e-3, rtol=1e-4)


def test_eltwise_modes(T):
    z, dy = _r((1000, 37), 12, 2.0), _r((1000, 37), 13)
    torch.testing.assert_close(T.eltwise(z, T.MODE_GELU), F.gelu(z), atol=1e-6, rtol=1e-5)
    zr = z.clone().requires_grad_(True)
    (F.gelu(zr) * dy).sum().backward()
    torch.testing.assert_close(T.eltwise(z, T.MODE_GELU_BWD, dy), zr.grad, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(T.eltwise(z, T.MODE_RELU), F.relu(z))
    torch.testing.assert_close(T.eltwise(z, T.MODE_RELU_BWD, dy), dy * (z > 0))
    torch.testing.assert_close(T.eltwise(z, T.MODE_ADD, dy), z + dy)
    d = T.eltwise(z, T.MODE_DROPOUT, p_drop=0.1, seed=5)
    kept = d != 0
    assert abs(1 - kept.float().mean().item() - 0.1) < 0.02
    torch.testing.assert_close(d[kept], (z / 0.9)[kept], atol=1e-6, rtol=1e-5)
    assert torch.equal(d, T.eltwise(z, T.MODE_DROPOUT, p_drop=0.1, seed=5))
    zb = z.to(BF)
    torch.testing.assert_close(T.eltwise(zb, T.MODE_GELU, out_dtype=BF).float(), F.gelu(zb.float()), atol=2e-2, rtol=1e-2)


def test_colsum_embed_adamw(T):
    x = _r((777, 300), 14)
    out = torch.zeros(300, device="cuda")
    torch.testing.assert_close(T.colsum(x, out), x.sum(0), atol=1e-3, rtol=1e-4)
    ids = torch.randint(0, 50, (96,), device="cuda")
    dy = _r((96, 64), 15)
    dw, dp = torch.zeros((50, 64), device="cuda"), torch.zeros((32, 64), device="cuda")
    T.embed_bwd(ids, dy, dw, dp, 32)
    torch.testing.assert_close(dw, torch.zeros_like(dw).index_add_(0, ids, dy), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(dp, dy.view(3, 32, 64).sum(0), atol=1e-4, rtol=1e-4)
    p = _r((1000,), 16); g = _r((1000,), 17)
    pt = torch.nn.Parameter(p.clone()); pt.grad = g.clone()
    opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.98), eps=1e-7, weight_decay=0.05)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in (1, 2, 3):
        opt.step()
        T.adamw_step(p, g, m, v, 1e-3, (0.9, 0.98), 1e-7, 0.05, step)
    torch.testing.assert_close(p, pt.data, atol=1e-6, rtol=1e-5)


# ------------------------------------------------------------------------------------------------ fused training attention (round 4)
_pair_keep = H.pair_keep


def _keep_mask(seed: int, g: int, h: int, lq: int, lk: int, p: float) -> torch.Tensor:
    """cir_attention_train_fwd / _bwd: row = (g * H + h) * Lq + query, col = key."""
    return _pair_keep(seed, g * h * lq, lk, p).view(g, h, lq, lk)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("g,h,lq,lk,masked,p", [(5, 2, 9, 9, True, 0.0), (3, 12, 32, 32, True, 0.1), (2, 3, 70, 45, False, 0.1),
                                                  (2, 12, 512, 577, False, 0.1), (4, 2, 33, 100, True, 0.25)])
def test_fused_training_attention_matches_torch_autograd(T, dtype, g, h, lq, lk, masked, p):
    """cir_attention_train_fwd / _bwd against torch autograd of dropout(softmax(q k^T * scale + mask)) v with the SAME dropout mask
    (regenerated on the host from the kernel's counter): context, log-sum-exp, dQ, dK, dV - self-attention shapes (ragged 9 x 9 with a
    key mask), the stacked cross-attention shape of the training step (512 queries x 577 keys), strided head views (fused q|k|v rows)."""
    dev = torch.device("cuda")
    gen = torch.Generator().manual_seed(g * 1000 + lq * 10 + lk)
    qkv = torch.randn((g * lq, 3, h * 64), generator=gen).to(dtype).to(dev)                       # fused projection rows: strided head views
    kvx = torch.randn((g * lk, 2, h * 64), generator=gen).to(dtype).to(dev)
    heads = lambda x, rows, part, parts: x.view(g, rows, parts, h, 64)[:, :, part].permute(0, 2, 1, 3)
    q4 = heads(qkv, lq, 0, 3)
    k4, v4 = (heads(kvx, lk, 0, 2), heads(kvx, lk, 1, 2))
    mask = None
    if masked:
        valid = torch.randint(1, lk + 1, (g,), generator=gen)
        mask = ((torch.arange(lk)[None] >= valid[:, None]).float() * -10000.0).to(dev).contiguous()
    scale, seed = 0.125, 123456789 + lq
    ctx = torch.empty((g * lq, h * 64), dtype=dtype, device=dev)
    out4 = ctx.view(g, lq, h, 64).permute(0, 2, 1, 3)
    lse = T.attention_train_fwd(q4, k4, v4, mask, out4, scale, p, seed)
    dout = (torch.randn((g * lq, h * 64), generator=gen) * 0.5).to(dtype).to(dev)
    dq = torch.full((g * lq, 3, h * 64), float("nan"), dtype=torch.float32, device=dev)
    dkv = torch.full((g * lk, 2, h * 64), float("nan"), dtype=torch.float32, device=dev)
    T.attention_train_bwd(q4, k4, v4, mask, out4, dout.view(g, lq, h, 64).permute(0, 2, 1, 3), lse, heads(dq, lq, 0, 3), heads(dkv, lk, 0, 2),
                          heads(dkv, lk, 1, 2), scale, p, seed)
    torch.cuda.synchronize()
    # ---- torch reference (fp32, same operands, same mask)
    qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q4, k4, v4))
    s = qf @ kf.transpose(-1, -2) * scale
    if mask is not None:
        s = s + mask[:, None, None, :]
    pr = torch.softmax(s, -1)
    keep = _keep_mask(seed, g, h, lq, lk, p).to(dev) if p > 0 else torch.ones_like(pr, dtype=torch.bool)
    o_ref = (pr * keep / (1.0 - p)) @ vf
    o_ref.backward(dout.float().view(g, lq, h, 64).permute(0, 2, 1, 3))
    lse_ref = torch.logsumexp(s, -1) * 1.4426950408889634
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    e_o = (out4.float() - o_ref).abs().max().item()
    e_l = (lse - lse_ref).abs().max().item()
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    e_q, e_k, e_v = rel(heads(dq, lq, 0, 3), qf.grad), rel(heads(dkv, lk, 0, 2), kf.grad), rel(heads(dkv, lk, 1, 2), vf.grad)
    print(f"\\n[fused train attention {dtype} {g}x{h} {lq}x{lk} mask={masked} p={p}] ctx {e_o:.2e}  lse {e_l:.2e}  dq {e_q:.2e}  dk {e_k:.2e}  dv {e_v:.2e}")
    assert e_o < tol and e_l < 2e-3 and max(e_q, e_k, e_v) < (2e-2 if dtype == torch.bfloat16 else 3e-3)
    assert torch.isfinite(dq[:, 0]).all() and torch.isfinite(dkv).all() and torch.isnan(dq[:, 1:]).all()      # only the q slice of the fused buffer is written
    # with the fp32 twin of O (what the trainer passes): D = rowsum(dO * O) equals sum_j Pd_ij dPd_ij term for term
    ctx32 = torch.empty((g * lq, h * 64), dtype=torch.float32, device=dev)
    o32 = ctx32.view(g, lq, h, 64).permute(0, 2, 1, 3)
    lse2 = T.attention_train_fwd(q4, k4, v4, mask, out4, scale, p, seed, out32=o32)
    assert torch.equal(lse2, lse) and (o32 - o_ref).abs().max().item() < tol
    assert (o32 - o_ref).abs().max().item() <= e_o + 1e-6                                              # fp32 twin: no output rounding
    dq2, dkv2 = torch.zeros_like(dq), torch.zeros_like(dkv)
    T.attention_train_bwd(q4, k4, v4, mask, out4, dout.view(g, lq, h, 64).permute(0, 2, 1, 3), lse, heads(dq2, lq, 0, 3), heads(dkv2, lk, 0, 2),
                          heads(dkv2, lk, 1, 2), scale, p, seed, out32=o32)
    e_q2, e_k2 = rel(heads(dq2, lq, 0, 3), qf.grad), rel(heads(dkv2, lk, 0, 2), kf.grad)
    print(f"   with the fp32 twin of O: dq {e_q2:.2e}  dk {e_k2:.2e}")
    assert max(e_q2, e_k2) < (2e-2 if dtype == torch.bfloat16 else 3e-3) and e_q2 < 1.1 * e_q + 1e-5


def test_fused_training_attention_16bit_gradients(T):
    """grad_dtype = operand type (ABI v10): the same gradients, rounded once on the way out, into strided head views."""
    dev, dtype, g, h, lq, lk = torch.device("cuda"), HF, 3, 4, 40, 70
    gen = torch.Generator().manual_seed(5)
    qkv = torch.randn((g * lq, 3, h * 64), generator=gen).to(dtype).to(dev)
    kvx = torch.randn((g * lk, 2, h * 64), generator=gen).to(dtype).to(dev)
    heads = lambda x, rows, part, parts: x.view(g, rows, parts, h, 64)[:, :, part].permute(0, 2, 1, 3)
    q4, k4, v4 = heads(qkv, lq, 0, 3), heads(kvx, lk, 0, 2), heads(kvx, lk, 1, 2)
    ctx = torch.empty((g * lq, h * 64), dtype=dtype, device=dev)
    out4 = ctx.view(g, lq, h, 64).permute(0, 2, 1, 3)
    lse = T.attention_train_fwd(q4, k4, v4, None, out4, 0.125, 0.1, 99)
    dout = (torch.randn((g * lq, h * 64), generator=gen) * 0.5).to(dtype).to(dev).view(g, lq, h, 64).permute(0, 2, 1, 3)
    res = {}
    for gd in (torch.float32, dtype):
        dq = torch.full((g * lq, 3, h * 64), float("nan"), dtype=gd, device=dev)
        dkv = torch.full((g * lk, 2, h * 64), float("nan"), dtype=gd, device=dev)
        T.attention_train_bwd(q4, k4, v4, None, out4, dout, lse, heads(dq, lq, 0, 3), heads(dkv, lk, 0, 2), heads(dkv, lk, 1, 2), 0.125, 0.1, 99)
        res[gd] = (dq, dkv)
    assert torch.equal(res[dtype][0][:, 0], res[torch.float32][0][:, 0].to(dtype)) and torch.isnan(res[dtype][0][:, 1:]).all()
    assert torch.equal(res[dtype][1], res[torch.float32][1].to(dtype))
    with pytest.raises(AssertionError):
        T.attention_train_bwd(q4, k4, v4, None, out4, dout, lse, heads(res[dtype][0], lq, 0, 3), heads(res[torch.float32][1], lk, 0, 2),
                              heads(res[torch.float32][1], lk, 1, 2), 0.125, 0.1, 99)


# ------------------------------------------------------------------------------------------------ fused row passes (round 4, train_fused.hip)
@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
@pytest.mark.parametrize("rows,cols,two,p", [(203, 768, False, 0.1), (64, 1024, True, 0.1), (7, 128, True, 0.0), (130, 132, False, 0.25)])
def test_residual_layernorm_train_and_adjoint(T, dtype, rows, cols, two, p):
    """cir_residual_layernorm_train / cir_layernorm_bwd_fused against torch autograd of LayerNorm(dropout(alpha (t0 + t1)) + res) with the
    kernel's own dropout mask (host-regenerated): pre, y (fp32 and 16-bit); d pre, dgamma, dbeta, the dense branch's 16-bit gradient incl.
    a second LayerNorm's contribution (t_add), and its column sums into two bias gradients."""
    dev = torch.device("cuda")
    t0, t1, res = _r((rows, cols), 31), (_r((rows, cols), 32) if two else None), _r((rows, cols), 33, 2.0)
    gam, bet = _r((cols,), 34) * 0.1 + 1.0, _r((cols,), 35)
    alpha, seed, eps = (0.5 if two else 1.0), 4242 + rows, 1e-12
    pre, y32, y16 = T.residual_layernorm_train(t0, t1, res, gam, bet, eps, dtype, alpha, p, seed)
    keep = _pair_keep(seed, rows, cols, p).to(dev) if p > 0 else torch.ones((rows, cols), dtype=torch.bool, device=dev)
    tr = (t0 + t1 if two else t0.clone()).requires_grad_(True)
    rr, gr, br = res.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    pre_ref = tr * alpha * keep / (1.0 - p) + rr
    y_ref = F.layer_norm(pre_ref, (cols,), gr, br, eps)
    torch.testing.assert_close(pre, pre_ref, atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(y32, y_ref, atol=2e-5, rtol=1e-5)
    assert torch.equal(y16, y32.to(dtype))
    # outputs into row ranges of larger buffers, no fp32 y
    big_pre, big16 = torch.zeros((rows + 8, cols), device=dev), torch.zeros((rows + 8, cols), dtype=dtype, device=dev)
    _, none32, _ = T.residual_layernorm_train(t0, t1, res, gam, bet, eps, dtype, alpha, p, seed, pre=big_pre[4:4 + rows], y16=big16[4:4 + rows], want32=False)
    assert none32 is None and torch.equal(big_pre[4:4 + rows], pre) and torch.equal(big16[4:4 + rows], y16) and not big_pre[:4].any() and not big16[-4:].any()
    # ---- adjoint
    dy, t_add = _r((rows, cols), 36), _r((rows, cols), 37, 0.5)
    (y_ref * dy).sum().backward()
    dg, db = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    b1, b2 = torch.zeros(cols, device=dev), torch.ones(cols, device=dev)
    dx, dt16 = T.layernorm_bwd_fused(pre, gam, dy, dg, db, eps, dtype, t_add=t_add, dbias=b1, dbias2=b2, alpha=alpha, p_drop=p, seed=seed)
    torch.testing.assert_close(dx, rr.grad, atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(dg, gr.grad, atol=2e-3, rtol=1e-4)
    torch.testing.assert_close(db, br.grad, atol=2e-3, rtol=1e-4)
    dt_ref = (rr.grad + t_add) * alpha * keep / (1.0 - p)
    tol = dict(atol=2e-2, rtol=1e-2) if dtype == BF else dict(atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(dt16.float(), dt_ref, **tol)
    torch.testing.assert_close(b1, dt_ref.sum(0), atol=3e-3, rtol=1e-4)
    torch.testing.assert_close(b2 - 1.0, dt_ref.sum(0), atol=3e-3, rtol=1e-4)
    # without t_add the dense gradient is the dropout adjoint of d pre itself (= torch's gradient of t)
    dg2, db2 = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    dx2, dt2 = T.layernorm_bwd_fused(pre, gam, dy, dg2, db2, eps, dtype, alpha=alpha, p_drop=p, seed=seed)
    assert torch.equal(dx2, dx)
    torch.testing.assert_close(dt2.float(), tr.grad, **tol)
    only_dx, none16 = T.layernorm_bwd_fused(pre, gam, dy, dg2, db2, eps, dtype, want_dt=False)
    assert none16 is None and torch.equal(only_dx, dx)


@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
def test_rows16_colsum_and_gelu_adjoint(T, dtype):
    dev = torch.device("cuda")
    rows, cols = 333, 3072
    a, z = _r((rows, cols), 41, dtype=dtype), _r((rows, cols), 42, 2.0, dtype=dtype)
    sums = torch.full((cols,), 2.0, device=dev)
    T.colsum16(a, sums)
    torch.testing.assert_close(sums - 2.0, a.float().sum(0), atol=2e-3, rtol=1e-4)
    wide = _r((70, 2 * 776), 43, dtype=dtype)                                         # strided halves (the merge layer's d cat)
    for half in (0, 1):
        v = wide[:, half * 776:(half + 1) * 776]
        s2 = torch.zeros(776, device=dev)
        torch.testing.assert_close(T.colsum16(v, s2), v.float().sum(0), atol=1e-3, rtol=1e-4)
    zr = z.float().requires_grad_(True)
    (F.gelu(zr) * a.float()).sum().backward()
    sb = torch.zeros(cols, device=dev)
    dz = T.gelu_bwd16(a, z, sums=sb)
    tol = dict(atol=2e-2, rtol=1e-2) if dtype == BF else dict(atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(dz.float(), zr.grad, **tol)
    torch.testing.assert_close(sb, zr.grad.sum(0), atol=5e-2 if dtype == BF else 5e-3, rtol=1e-3)
    assert torch.equal(T.gelu_bwd16(a, z), dz)


def test_transpose16_multi(T):
    """cir_transpose16_multi: every matrix of a flat buffer transposed at its own offset in one launch (ragged extents, a 1-row matrix)."""
    shapes = [(70, 33), (768, 128), (1, 40), (2304, 96), (32, 32)]
    entries, off = [], 0
    for r, c in shapes:
        entries.append((off, r, c))
        off += (r * c + 7) // 8 * 8
    src = _r((off,), 61, dtype=HF)
    dst = torch.zeros_like(src)
    plan = T.TransposePlan(entries, src.device)
    plan.run(src, dst)
    for o, r, c in entries:
        assert torch.equal(dst[o:o + r * c].view(c, r), src[o:o + r * c].view(r, c).t())


@pytest.mark.parametrize("out_dtype", [torch.float32, HF, BF], ids=["f32", "fp16", "bf16"])
def test_rows_scale_add(T, out_dtype):
    """cir_rows_scale_add (DropPath): out = a + scale[row // rows_per_group] * b, and the scaled b alone as a 16-bit operand."""
    groups, rpg, cols = 5, 17, 132
    a, b = _r((groups * rpg, cols), 71), _r((groups * rpg, cols), 72)
    sc = torch.tensor([0.0, 2.5, 1.0, 0.0, 1.25], device="cuda")
    ref = a + sc.repeat_interleave(rpg)[:, None] * b
    out = T.rows_scale_add(a, b, sc, rpg, out_dtype=out_dtype)
    assert out.dtype == out_dtype and torch.equal(out, ref.to(out_dtype))
    only = T.rows_scale_add(None, b, sc, rpg, out_dtype=out_dtype)
    assert torch.equal(only, (sc.repeat_interleave(rpg)[:, None] * b).to(out_dtype))
