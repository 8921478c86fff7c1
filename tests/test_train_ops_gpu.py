"""Training-mode operators (include/cirrank.h) against plain fp32 PyTorch on the MI355X."""
import pytest
import torch
import torch.nn.functional as F

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
