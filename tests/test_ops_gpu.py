"""Per-kernel parity on a real MI355X: each C-ABI entry point against a plain torch fp32 reference
of the same op on the same (16-bit-rounded) inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from candidate_reranking_cir_amd import ops as _ops
    return _ops


def _rand(shape, dtype, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.fixture(params=[128, 256], ids=["128", "256"])
def gemm_tile(request):
    """Run the GEMM tests once per kernel (cir_set_tuning forces the 128x128 or the persistent 256x256 kernel)."""
    from candidate_reranking_cir_amd import lib
    lib.set_tuning(lib.TUNE_GEMM_TILE, request.param)
    yield str(request.param)
    lib.set_tuning(lib.TUNE_GEMM_TILE, 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k", [(1, 16, 64), (16, 128, 64), (130, 128, 128), (257, 768, 768), (300, 2304, 768), (64, 768, 3072), (1000, 1536, 1536), (515, 528, 192), (700, 256, 64), (513, 768, 128)])
def test_gemm_exact_integers(ops, gemm_tile, dtype, m, n, k):
    """Small-integer operands make every product and sum exact: any fragment/layout slip shows up
    as a wrong integer (asymmetric data on both sides)."""
    g = torch.Generator(device="cpu").manual_seed(m * 7 + n)
    a = torch.randint(-3, 4, (m, k), generator=g).to(dtype).cuda()
    w = torch.randint(-3, 4, (n, k), generator=g).to(dtype).cuda()
    bias = torch.randint(-5, 6, (n,), generator=g).float().cuda()
    ref = a.float() @ w.float().T + bias
    out = ops.gemm(a, w, bias, out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("out32", [True, False])
def test_gemm_epilogues(ops, gemm_tile, dtype, act, out32):
    m, n, k = 333, 256, 256          # K % 128 == 0: the forced 256-tile run really is the 256 x 256 kernel
    a, w = _rand((m, k), dtype, seed=1), _rand((n, k), dtype, 0.1, seed=2)
    bias = _rand((n,), torch.float32, seed=3)
    res = _rand((m, n), torch.float32, seed=4)
    y = a.float() @ w.float().T + bias
    y = F.gelu(y) if act == 1 else (F.relu(y) if act == 2 else y)
    ref = y + res
    out = ops.gemm(a, w, bias, residual=res, act=act, out_dtype=torch.float32 if out32 else dtype)
    torch.cuda.synchronize()
    tol = 1e-4 if out32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)
    assert out.dtype == (torch.float32 if out32 else dtype)
    torch.testing.assert_close(out.float(), ref, atol=tol * 4, rtol=tol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [1, 2])
@pytest.mark.parametrize("out32", [True, False])
def test_gemm_activation_without_residual(ops, gemm_tile, dtype, act, out32):
    """GELU / ReLU epilogues with no residual: the combination the 256 x 256 kernel itself serves (fc1 of every FFN),
    on a ragged M and several K-tile pairs so that tile seams and the peeled last pair are on the path."""
    m, n, k = 777, 512, 384
    a, w = _rand((m, k), dtype, seed=11), _rand((n, k), dtype, 0.1, seed=12)
    bias = _rand((n,), torch.float32, seed=13)
    y = a.float() @ w.float().T + bias
    ref = F.gelu(y) if act == 1 else F.relu(y)
    out = ops.gemm(a, w, bias, act=act, out_dtype=torch.float32 if out32 else dtype)
    torch.cuda.synchronize()
    tol = 1e-4 if out32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)
    torch.testing.assert_close(out.float(), ref, atol=tol * 4, rtol=tol)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_batched_strided_inplace_residual(ops, gemm_tile, dtype):
    nb, m, n, k = 2, 200, 128, 256
    big = _rand((nb, m, k + 64), dtype, seed=5)
    a = big[:, :, :k]                       # row stride k+64
    w = _rand((nb, n, k), dtype, 0.1, seed=6)
    bias = _rand((nb, n), torch.float32, seed=7)
    res = _rand((nb, m, n), torch.float32, seed=8)
    ref = torch.einsum("bmk,bnk->bmn", a.float(), w.float()) + bias[:, None, :] + res
    cbuf = torch.zeros((m, nb, n), dtype=torch.float32, device="cuda")
    out = ops.gemm(a, w, bias, residual=res, out_dtype=torch.float32, out=cbuf.permute(1, 0, 2))  # interleaved output
    torch.cuda.synchronize()
    torch.testing.assert_close(out, ref, atol=1e-3, rtol=1e-4)
    # C aliasing the residual (x += f(x) pattern of the ViT blocks)
    x = res.clone()
    ops.gemm(a, w, bias, residual=x, out_dtype=torch.float32, out=x)
    torch.cuda.synchronize()
    torch.testing.assert_close(x, ref, atol=1e-3, rtol=1e-4)


def test_gemm_argument_errors(ops):
    from candidate_reranking_cir_amd.lib import CirrankError
    a = torch.zeros((4, 60), dtype=torch.bfloat16, device="cuda")
    w = torch.zeros((16, 60), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(CirrankError):
        ops.gemm(a, w)                       # K % 64 != 0
    with pytest.raises(CirrankError):
        ops.gemm(a.cpu(), w.cpu())           # no CPU fallback


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cols,eps", [(768, 1e-12), (768, 1e-6), (128, 1e-12), (1024, 1e-6), (64, 1e-5)])
def test_layernorm(ops, dtype, cols, eps):
    rows = 203
    x = _rand((rows, cols), torch.float32, 2.0, seed=1) + 0.5
    res = _rand((rows, cols), torch.float32, seed=2)
    g, b = _rand((cols,), torch.float32, seed=3), _rand((cols,), torch.float32, seed=4)
    ref = F.layer_norm(x + res, (cols,), g, b, eps)
    y32, y16 = ops.layernorm(x, g, b, eps, residual=res, dtype16=dtype)
    torch.cuda.synchronize()
    torch.testing.assert_close(y32, ref, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(y16.float(), ref.to(dtype).float(), atol=1e-3, rtol=8e-3 if dtype == torch.bfloat16 else 1e-3)  # <= 1 ulp


def test_layernorm_twin_shared_input(ops):
    """LayerNormA(m + h0), LayerNormB(m + h1): shared x, per-branch residual and affine."""
    rows, cols = 70, 768
    m = _rand((rows, cols), torch.float32, seed=1)
    h = _rand((2, rows, cols), torch.float32, seed=2)
    g, b = _rand((2, cols), torch.float32, seed=3), _rand((2, cols), torch.float32, seed=4)
    y32, y16 = ops.layernorm(m, g, b, 1e-12, residual=h)
    torch.cuda.synchronize()
    for br in (0, 1):
        torch.testing.assert_close(y32[br], F.layer_norm(m + h[br], (cols,), g[br], b[br], 1e-12), atol=2e-5, rtol=1e-5)
    assert y16.shape == (2, rows, cols)


def test_embed_layernorm(ops):
    vocab, cols, r, l = 1000, 768, 7, 13
    word, pos = _rand((vocab, cols), torch.float32, seed=1), _rand((512, cols), torch.float32, seed=2)
    g, b = _rand((cols,), torch.float32, seed=3), _rand((cols,), torch.float32, seed=4)
    ids = torch.randint(0, vocab, (r, l), generator=torch.Generator().manual_seed(0)).cuda()
    ref = F.layer_norm(word[ids] + pos[:l][None], (cols,), g, b, 1e-12)
    y32, y16 = ops.embed_layernorm(ids, word, pos, g, b, 1e-12)
    torch.cuda.synchronize()
    torch.testing.assert_close(y32, ref, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(y16.float(), ref.bfloat16().float(), atol=1e-3, rtol=8e-3)


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, scale, mask):
    b1, b0, lq, d = q.shape
    h = d // 64
    qh = q.float().reshape(b1, b0, lq, h, 64).transpose(2, 3)
    kh = k.float().reshape(b1, b0, -1, h, 64).transpose(2, 3)
    vh = v.float().reshape(b1, b0, -1, h, 64).transpose(2, 3)
    s = qh @ kh.transpose(-1, -2) * scale
    if mask is not None:
        s = s + mask[:, :, None, None, :]
    return (torch.softmax(s, -1) @ vh).transpose(2, 3).reshape(b1, b0, lq, d)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("lq,lk", [(1, 1), (5, 7), (32, 32), (33, 64), (32, 197), (197, 197), (11, 577), (64, 40),
                                   (577, 577), (300, 608), (70, 640), (520, 300)])
def test_attention_shapes(ops, dtype, lq, lk):
    b1, b0, h = 3, 2, 2
    q, k, v = _rand((b1, b0, lq, h * 64), dtype, seed=1), _rand((b1, b0, lk, h * 64), dtype, seed=2), _rand((b1, b0, lk, h * 64), dtype, seed=3)
    out = torch.full_like(q, float("nan"))
    ops.attention(q, k, v, out, 0.125)
    torch.cuda.synchronize()
    ref = _attn_ref(q, k, v, 0.125, None)
    torch.testing.assert_close(out.float(), ref, atol=2e-2 if dtype == torch.bfloat16 else 4e-3, rtol=0)


def test_attention_exact_uniform(ops):
    """q = 0 -> uniform softmax -> out = mean of v over the valid keys: checks the V^T fragment
    addressing / key masking with values whose mean is exactly representable."""
    lq, lk, h = 40, 48, 1
    q = torch.zeros((1, 1, lq, 64), dtype=torch.bfloat16, device="cuda")
    k = _rand((1, 1, lk, 64), torch.bfloat16, seed=1)
    v = torch.zeros((1, 1, lk, 64), dtype=torch.bfloat16, device="cuda")
    key_ids = torch.arange(lk, device="cuda")
    v[0, 0] = ((key_ids[:, None] % 4) * 16 + torch.arange(64, device="cuda")[None] % 16).to(torch.bfloat16)
    out = torch.empty_like(q)
    ops.attention(q, k, v, out, 0.125)
    torch.cuda.synchronize()
    ref = v.float().mean(dim=2, keepdim=True).expand(-1, -1, lq, -1)
    torch.testing.assert_close(out.float(), ref, atol=0.13, rtol=0)   # bf16 rounding of 1/48 weights


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_mask_and_strides(ops, dtype):
    """ViT-style packed qkv, BERT-style additive key masks (-10000) and the cross-attention layout
    (keys/values interleaved per branch, output interleaved per branch)."""
    b, n, h = 4, 50, 3
    d = h * 64
    qkv = _rand((b, n, 3, d), dtype, seed=1)
    out = torch.empty((b, n, d), dtype=dtype, device="cuda")
    ops.attention(qkv[:, :, 0].unsqueeze(1), qkv[:, :, 1].unsqueeze(1), qkv[:, :, 2].unsqueeze(1), out.unsqueeze(1), 0.125)
    ref = _attn_ref(qkv[:, :, 0].unsqueeze(1), qkv[:, :, 1].unsqueeze(1), qkv[:, :, 2].unsqueeze(1), 0.125, None)[:, 0]
    torch.cuda.synchronize()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    torch.testing.assert_close(out.float(), ref, atol=tol, rtol=0)

    t, l, nk = 5, 12, 37
    qb = _rand((2, t, l, d), dtype, seed=2)
    kv = _rand((t, nk, 4, d), dtype, seed=3)                # [K0, V0, K1, V1]
    cc = torch.empty((t, l, 2, d), dtype=dtype, device="cuda")
    valid = torch.tensor([37, 30, 1, 36, 20], device="cuda")
    mask = ((torch.arange(nk, device="cuda")[None] >= valid[:, None]).float() * -10000.0)  # (t, nk)
    q4 = qb.permute(1, 0, 2, 3)
    k4 = kv[:, :, 0::2].permute(0, 2, 1, 3)
    v4 = kv[:, :, 1::2].permute(0, 2, 1, 3)
    m3 = mask[:, None, :].expand(t, 2, nk)
    ops.attention(q4, k4, v4, cc.permute(0, 2, 1, 3), 0.125, mask=m3)
    torch.cuda.synchronize()
    ref = _attn_ref(q4, k4, v4, 0.125, m3)
    torch.testing.assert_close(cc.permute(0, 2, 1, 3).float(), ref, atol=tol, rtol=0)
    # finfo.min style encoder mask on every key -> uniform attention, like the reference's softmax
    mall = torch.full((t, 2, nk), torch.finfo(torch.float32).min, device="cuda")
    ops.attention(q4, k4, v4, cc.permute(0, 2, 1, 3), 0.125, mask=mall)
    torch.cuda.synchronize()
    ref = _attn_ref(q4, k4, v4, 0.125, mall)
    assert torch.isfinite(cc.float()).all()
    torch.testing.assert_close(cc.permute(0, 2, 1, 3).float(), ref, atol=tol, rtol=0)


# ------------------------------------------------------------------------------------------------ small ops
@pytest.mark.parametrize("src", [torch.float32, torch.bfloat16])
def test_patchify_matches_conv(ops, src):
    b, c, hw, p, d = 3, 3, 64, 16, 128
    img = _rand((b, c, hw, hw), src, seed=1)
    wconv = _rand((d, c, p, p), torch.bfloat16, 0.05, seed=2)
    patches = ops.patchify(img, p)
    torch.cuda.synchronize()
    ref_patches = F.unfold(img.float(), kernel_size=p, stride=p).transpose(1, 2).reshape(-1, c * p * p)
    assert torch.equal(patches.float(), ref_patches.bfloat16().float())
    out = ops.gemm(patches, wconv.reshape(d, -1), out_dtype=torch.float32)
    ref = F.conv2d(img.bfloat16().float(), wconv.float(), stride=p).flatten(2).transpose(1, 2).reshape(-1, d)
    torch.testing.assert_close(out, ref, atol=2e-3, rtol=1e-3)


def test_patchify_patch8_scalar_path(ops):
    """Patch sizes other than 16 take the element-wise path."""
    b, c, hw, p = 2, 3, 32, 8
    img = _rand((b, c, hw, hw), torch.bfloat16, seed=4)
    patches = ops.patchify(img, p)
    torch.cuda.synchronize()
    ref = F.unfold(img.float(), kernel_size=p, stride=p).transpose(1, 2).reshape(-1, c * p * p)
    assert torch.equal(patches.float(), ref)


def test_vit_assemble(ops):
    b, p, d = 3, 16, 128
    proj, cls, pos = _rand((b * p, d), torch.float32, seed=1), _rand((d,), torch.float32, seed=2), _rand((p + 1, d), torch.float32, seed=3)
    x = ops.vit_assemble(proj, cls, pos, b)
    ref = torch.cat([cls.expand(b, 1, d), proj.view(b, p, d)], 1) + pos[None]
    torch.cuda.synchronize()
    assert torch.equal(x, ref)


def test_small_linear(ops):
    m, n, k = 101, 2, 768
    x, w, bias = _rand((m, k), torch.bfloat16, seed=1), _rand((n, k), torch.bfloat16, 0.1, seed=2), _rand((n,), torch.float32, seed=3)
    y = ops.small_linear(x, w, bias)
    torch.cuda.synchronize()
    torch.testing.assert_close(y, x.float() @ w.float().T + bias, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("q,k", [(1, 1), (7, 5), (33, 100), (4, 205), (3, 2048)])
def test_argsort_desc(ops, q, k):
    g = torch.Generator().manual_seed(k)
    logits = torch.randn((q, k), generator=g)
    logits[0, : k // 2] = -99999.99                       # skipped-row style ties
    if k > 3:
        logits[-1, 1] = logits[-1, 3]
    idx = ops.argsort_desc(logits.cuda())
    torch.cuda.synchronize()
    ref = torch.argsort(logits, dim=-1, descending=True, stable=True)
    assert torch.equal(idx.cpu(), ref)


@pytest.mark.parametrize("src,dst", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                     (torch.float16, torch.float32), (torch.float32, torch.float16)])
def test_gather_rows(ops, src, dst):
    bank = _rand((9, 5, 64), src, seed=1)
    idx = torch.tensor([3, 3, 0, 8, 1, 7], device="cuda")
    out = ops.gather_rows(bank, idx, dst)
    torch.cuda.synchronize()
    assert torch.equal(out, bank[idx].to(dst))
    assert torch.equal(ops.gather_rows(bank, None, dst), bank.to(dst))


def test_attention_kv_bank_index(ops):
    """kv_index: items attend to rows of a K/V bank (cross-query K/V cache) - equals gathering first."""
    t, l, nk, h, rows = 6, 9, 37, 2, 4
    d = h * 64
    q = _rand((t, 2, l, d), torch.bfloat16, seed=1)
    bank = _rand((rows, nk, 4, d), torch.bfloat16, seed=2)
    idx = torch.tensor([3, 0, 0, 2, 1, 3], device="cuda")
    k4, v4 = bank[:, :, 0::2].permute(0, 2, 1, 3), bank[:, :, 1::2].permute(0, 2, 1, 3)
    out_a, out_b = torch.empty_like(q), torch.empty_like(q)
    ops.attention(q, k4, v4, out_a, 0.125, kv_index=idx)
    g = bank[idx]
    ops.attention(q, g[:, :, 0::2].permute(0, 2, 1, 3), g[:, :, 1::2].permute(0, 2, 1, 3), out_b, 0.125)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)


# ------------------------------------------------------------------------------------------------ 16-bit residual stream
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_stream16_epilogue(ops, gemm_tile, dtype):
    """fp16 residual stream: C (and R) fp16 from bf16 or fp16 operands; the residual is added in fp32 (128-tile kernel: one
    rounding; 256-tile kernel: on the fp16-rounded GEMM result, two roundings)."""
    m, n, k = 333, 256, 256
    a, w = _rand((m, k), dtype, seed=1), _rand((n, k), dtype, 0.1, seed=2)
    bias = _rand((n,), torch.float32, seed=3)
    res = _rand((m, n), torch.float16, 2.0, seed=4)
    ref = a.float() @ w.float().T + bias
    out = ops.gemm(a, w, bias, out_dtype=torch.float16)
    torch.testing.assert_close(out.float(), ref.half().float(), atol=2e-3, rtol=1e-3)
    out = ops.gemm(a, w, bias, residual=res, out_dtype=torch.float16)
    torch.cuda.synchronize()
    assert out.dtype == torch.float16
    torch.testing.assert_close(out.float(), (ref + res.float()).half().float(), atol=4e-3, rtol=1.5e-3)   # <= 1-2 ulp of the fp16 sum
    x = res.clone()
    ops.gemm(a, w, bias, residual=x, out_dtype=torch.float16, out=x)                                     # in place (ViT blocks)
    torch.cuda.synchronize()
    assert torch.equal(x, out)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_stream16(ops, dtype):
    rows, cols = 203, 768
    x = (_rand((rows, cols), torch.float32, 2.0, seed=1) + 0.5).half()
    res = _rand((rows, cols), torch.float32, seed=2).half()
    g, b = _rand((cols,), torch.float32, seed=3), _rand((cols,), torch.float32, seed=4)
    ref = F.layer_norm(x.float() + res.float(), (cols,), g, b, 1e-12)
    ys, y16 = ops.layernorm(x, g, b, 1e-12, residual=res, dtype16=dtype)
    torch.cuda.synchronize()
    assert ys.dtype == torch.float16 and y16.dtype == dtype
    torch.testing.assert_close(ys.float(), ref.half().float(), atol=1e-3, rtol=1e-3)
    torch.testing.assert_close(y16.float(), ref.to(dtype).float(), atol=1e-3, rtol=8e-3 if dtype == torch.bfloat16 else 1e-3)
    y32, _ = ops.layernorm(x, g, b, 1e-12, residual=res, dtype16=None, stream_dtype=torch.float32)      # fp16 in, fp32 out
    torch.testing.assert_close(y32, ref, atol=2e-5, rtol=1e-5)
    yh, _ = ops.layernorm(x.float(), g, b, 1e-12, residual=res.float(), dtype16=None, stream_dtype=torch.float16)   # fp32 in, fp16 out
    torch.testing.assert_close(yh.float(), ref.half().float(), atol=1e-3, rtol=1e-3)


def test_vit_assemble_and_embed_stream16(ops):
    b, p, d = 3, 16, 128
    proj, cls, pos = _rand((b * p, d), torch.float16, seed=1), _rand((d,), torch.float32, seed=2), _rand((p + 1, d), torch.float32, seed=3)
    x = ops.vit_assemble(proj, cls, pos, b)
    ref = (torch.cat([cls.expand(b, 1, d), proj.float().view(b, p, d)], 1) + pos[None]).half()
    torch.cuda.synchronize()
    assert x.dtype == torch.float16 and torch.equal(x, ref)
    vocab, cols, r, l = 1000, 768, 7, 13
    word, posw = _rand((vocab, cols), torch.float32, seed=1), _rand((512, cols), torch.float32, seed=2)
    g, be = _rand((cols,), torch.float32, seed=3), _rand((cols,), torch.float32, seed=4)
    ids = torch.randint(0, vocab, (r, l), generator=torch.Generator().manual_seed(0)).cuda()
    refe = F.layer_norm(word[ids] + posw[:l][None], (cols,), g, be, 1e-12)
    ys, y16 = ops.embed_layernorm(ids, word, posw, g, be, 1e-12, stream_dtype=torch.float16)
    torch.cuda.synchronize()
    torch.testing.assert_close(ys.float(), refe.half().float(), atol=1e-3, rtol=1e-3)
    torch.testing.assert_close(y16.float(), refe.bfloat16().float(), atol=1e-3, rtol=8e-3)


# ------------------------------------------------------------------------------------------------ CLS cross-attention (K / V folded out)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("t,n,d", [(1, 1, 128), (3, 17, 128), (5, 197, 768), (2, 577, 768), (4, 33, 256), (300, 70, 384)])
def test_cls_cross_attention_matches_fp32_torch(ops, dtype, t, n, d):
    """out[t, r] = sum_j softmax_j(qp[t, r] . x[t, j] * scale) x[t, j] for 32 query rows per item (cir_cls_cross_attention):
    ragged key counts (one key, < one tile, several tiles), every supported width class, more items than CUs x 2."""
    g = torch.Generator(device="cpu").manual_seed(t * 31 + n)
    x = torch.randn((t, n, d), generator=g).to(dtype).cuda()
    qp = (torch.randn((t, 32, d), generator=g) * 0.3).to(dtype).cuda()
    out = ops.cls_cross_attention(x, qp, 0.125)
    torch.cuda.synchronize()
    s = torch.einsum("trd,tnd->trn", qp.float(), x.float()) * 0.125
    ref = torch.softmax(s, -1) @ x.float()
    assert out.shape == (t, 32, d) and out.dtype == dtype
    torch.testing.assert_close(out.float(), ref, atol=1.5e-2 if dtype == torch.bfloat16 else 2e-3, rtol=0)


def test_cls_cross_attention_bank_index(ops):
    """`x_index`: item t attends row x_index[t] of a token bank - bit-identical to gathering the rows first."""
    dtype = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(9)
    bank = torch.randn((11, 70, 256), generator=g).to(dtype).cuda()
    idx = torch.tensor([3, 3, 10, 0, 7, 1], dtype=torch.int64).cuda()
    qp = (torch.randn((6, 32, 256), generator=g) * 0.3).to(dtype).cuda()
    a = ops.cls_cross_attention(bank, qp, 0.125, x_index=idx)
    b = ops.cls_cross_attention(bank[idx].contiguous(), qp, 0.125)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_cls_cross_attention_equals_projected_attention(ops):
    """The identity the last fusion layer relies on: attention over K = x W_k^T + b_k, V = x W_v^T + b_v with ONE query row
    per head equals W_v (sum_j p_j x_j) + b_v with p from (W_k^T q) . x_j - the key bias drops out of the softmax."""
    dtype, t, n, d, h = torch.float16, 6, 197, 768, 12
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((t, n, d), generator=g).to(dtype).cuda()
    q = torch.randn((t, d), generator=g).to(dtype).cuda()
    wk, wv = ((torch.randn((d, d), generator=g) * 0.03).to(dtype).cuda() for _ in range(2))
    bk, bv = (torch.randn((d,), generator=g).cuda() * 0.5 for _ in range(2))
    # reference: projected K / V through the ordinary operators
    k = ops.gemm(x.view(t * n, d), wk, bk).view(t, 1, n, d)
    v = ops.gemm(x.view(t * n, d), wv, bv).view(t, 1, n, d)
    ctx = torch.empty((t, 1, 1, d), dtype=dtype, device="cuda")
    ops.attention(q.view(t, 1, 1, d), k, v, ctx, 0.125)
    # folded: qp[t, head] = W_k[head rows]^T q[t, head slice]; out rows -> W_v[head rows] (.) + b_v
    qp = torch.zeros((t, 32, d), dtype=dtype, device="cuda")
    wkt = wk.view(h, 64, d).transpose(1, 2).contiguous()
    ops.gemm(q.view(t, h, 64).permute(1, 0, 2), wkt, None, out=qp[:, :h, :].permute(1, 0, 2))
    o = ops.cls_cross_attention(x, qp, 0.125)
    folded = torch.empty((t, d), dtype=dtype, device="cuda")
    ops.gemm(o[:, :h, :].permute(1, 0, 2), wv.view(h, 64, d), bv.view(h, 64), out=folded.view(t, h, 64).permute(1, 0, 2))
    torch.cuda.synchronize()
    torch.testing.assert_close(folded.float(), ctx.view(t, d).float(), atol=4e-3, rtol=0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", ["plain16", "gelu16", "relu16", "out32", "out32_res", "stream16", "stream16_res"])
def test_gemm_tile_choice_is_bit_invariant(ops, dtype, case):
    """The dispatcher picks the 128- or the 256-wide tile from the number of tiles, i.e. from the batch size.  Both kernels start
    their accumulators at the bias, walk K in the same order through the same MFMA shape and run the same epilogue arithmetic
    (packed GELU; fp16-stream output rounded before AND after the residual joins it), so the choice never changes a bit:
    results are independent of how a batch is split (what the distributed bit-identity tests rely on)."""
    from candidate_reranking_cir_amd import lib
    m, n, k = 1024 + 40, 512, 256                                  # ragged M edge included
    a, w = _rand((m, k), dtype, seed=21), _rand((n, k), dtype, 0.1, seed=22)
    bias = _rand((n,), torch.float32, seed=23)
    kw = dict(plain16={}, gelu16=dict(act=ops.ACT_GELU), relu16=dict(act=ops.ACT_RELU), out32=dict(out_dtype=torch.float32),
              out32_res=dict(out_dtype=torch.float32, residual=_rand((m, n), torch.float32, 2.0, seed=24)),
              stream16=dict(out_dtype=torch.float16), stream16_res=dict(out_dtype=torch.float16, residual=_rand((m, n), torch.float16, 2.0, seed=24)))[case]
    outs = {}
    try:
        for tile in (128, 256):
            lib.set_tuning(lib.TUNE_GEMM_TILE, tile)
            outs[tile] = ops.gemm(a, w, bias, **kw)
    finally:
        lib.set_tuning(lib.TUNE_GEMM_TILE, 0)
    torch.cuda.synchronize()
    assert torch.equal(outs[128], outs[256]), f"{case}: {(outs[128].float() - outs[256].float()).abs().max().item():.3e}"
    if case == "stream16_res":                                      # two roundings: half an ulp of the pre-residual value and of the sum
        mid = a.float() @ w.float().T + bias
        ref = mid + kw["residual"].float()
        ulp_of = lambda t: torch.maximum(t.abs(), torch.tensor(2.0 ** -14, device="cuda")).log2().floor().exp2() * 2.0 ** -10
        assert bool(((outs[128].float() - ref).abs() <= ulp_of(mid) * 0.51 + ulp_of(ref) * 0.51 + 1e-5).all())
