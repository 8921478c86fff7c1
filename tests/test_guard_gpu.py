"""Bounds ("guard band") and persistent-tile tests on a real MI355X (compute-sanitizer-style bounds
test on padded tiles; GPU AddressSanitizer is not available on this pool).

  * Every kernel with hand-predicated stores writes into a view that sits INSIDE a larger canary-filled allocation
    (slack rows before / after and, where the ABI takes a leading dimension, slack columns left / right); after the
    launch every canary element must be bit-identical - a store outside the declared output is the one class of bug an
    output comparison cannot see.
  * The persistent 256 x 256 GEMM is driven with MORE tiles than CUs (> 256), so every workgroup walks several tiles:
    streamed tile seams, counted store waits, dead-slot refills and the skipped prologue waits are all on the path, and
    the result is compared EXACTLY (small-integer operands: every product and partial sum is exact in fp32) with an
    fp32 torch matmul, for all three instantiations (16-bit out / fp32 out / fp32 out + residual, incl. C aliasing R).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
CANARY = {torch.float32: 0x7FC0DEAD, torch.bfloat16: 0x7FC1, torch.float16: 0x7E01, torch.int64: 0x7EADBEEF7EADBEEF}   # NaN patterns
_INT = {torch.float32: torch.int32, torch.bfloat16: torch.int16, torch.float16: torch.int16, torch.int64: torch.int64}


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from candidate_reranking_cir_amd import ops as _ops
    return _ops


class Guarded:
    """A (rows, cols) output view inside a canary-filled (rows + 2*pr, cols + 2*pc) allocation."""

    def __init__(self, rows, cols, dtype, pr=40, pc=64, batch=None):
        self.dtype, self.pr, self.pc, self.rows, self.cols = dtype, pr, pc, rows, cols
        shape = (rows + 2 * pr, cols + 2 * pc) if batch is None else (batch, rows + 2 * pr, cols + 2 * pc)
        self.big = torch.empty(shape, dtype=dtype, device="cuda")
        self.big.view(_INT[dtype]).fill_(CANARY[dtype])
        self.view = self.big[..., pr:pr + rows, pc:pc + cols]

    def fill_view(self, src):
        self.view.copy_(src)
        return self.view

    def assert_intact(self, what=""):
        bits = self.big.view(_INT[self.dtype]).clone()
        inside = bits[..., self.pr:self.pr + self.rows, self.pc:self.pc + self.cols]
        inside.fill_(CANARY[self.dtype])
        bad = (bits != CANARY[self.dtype])
        assert not bool(bad.any()), f"{what}: {int(bad.sum())} canary elements overwritten, first at {bad.nonzero()[0].tolist()}"


def _ints(shape, lo, hi, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).to(dtype).cuda()


# ------------------------------------------------------------------------------------------------ persistent GEMM, > 256 tiles
BIG_SHAPES = [(79588, 768, 768), (79588, 2304, 768), (79588, 3072, 768), (79588, 768, 3072), (66049, 2304 + 16, 768), (70001, 784, 256)]


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("m,n,k", BIG_SHAPES)
@pytest.mark.parametrize("variant", ["out16", "out32", "out32_res", "out32_res_alias", "stream16", "stream16_res", "stream16_res_alias"])
def test_persistent_gemm_exact_with_guard_bands(ops, dtype, m, n, k, variant):
    """> 256 tiles of 256 x 256 (e.g. 311 x 3 for the ViT proj shape of a 404-image chunk): every workgroup of the
    persistent kernel processes 3+ tiles.  Exact integers; output inside canaries (ragged M edge, ragged N for the
    N = 2320 / 784 cases, where the last n-tile is 16 wide)."""
    tiles = -(-m // 256) * -(-n // 256)
    assert tiles > 256
    amp = 1 if variant.startswith("stream16") else 3    # fp16 stream: keep |sum| < 2048 so that every rounding is exact
    a = _ints((m, k), -amp, amp, dtype, seed=m + k)
    w = _ints((n, k), -amp, amp, dtype, seed=n + k + 1)
    bias = _ints((n,), -5, 5, torch.float32, seed=3)
    ref = a.float() @ w.float().T + bias               # exact: |sum| <= 9 * 3072 + 5 < 2^24
    out_dtype = dtype if variant == "out16" else (torch.float16 if variant.startswith("stream16") else torch.float32)
    guard = Guarded(m, n, out_dtype)
    res = None
    if variant in ("out32_res", "stream16_res"):
        res = _ints((m, n), -7, 7, out_dtype, seed=5)
        ref = ref + res.float()
    elif variant in ("out32_res_alias", "stream16_res_alias"):
        res = guard.fill_view(_ints((m, n), -7, 7, out_dtype, seed=6))
        ref = ref + res.float()
    out = ops.gemm(a, w, bias, residual=res, out_dtype=out_dtype, out=guard.view)
    torch.cuda.synchronize()
    if variant == "out16" or variant.startswith("stream16"):
        assert torch.equal(out, ref.to(out_dtype))      # exact fp32 sum (+ bias + residual); roundings exact or one RNE on both sides
    else:
        assert torch.equal(out, ref)
    guard.assert_intact(f"gemm {variant} {m}x{n}x{k}")


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("act", [1, 2], ids=["gelu", "relu"])
def test_persistent_gemm_activation_many_tiles(ops, dtype, act):
    """fc1 shape of the FFN (GELU epilogue) with > 256 tiles against fp32 torch on the same 16-bit operands."""
    m, n, k = 40000, 3072, 768
    g = torch.Generator(device="cpu").manual_seed(9)
    a = (torch.randn((m, k), generator=g)).to(dtype).cuda()
    w = (torch.randn((n, k), generator=g) * 0.03).to(dtype).cuda()
    bias = torch.randn((n,), generator=g).cuda()
    y = a.float() @ w.float().T + bias
    ref = F.gelu(y) if act == 1 else F.relu(y)
    guard = Guarded(m, n, dtype)
    out = ops.gemm(a, w, bias, act=act, out=guard.view)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs()
    tol = 8e-3 if dtype == torch.bfloat16 else 1e-3        # one 16-bit rounding of O(1) outputs (+1.2e-5 GELU fit)
    assert float((err / (1.0 + ref.abs())).max()) < tol
    guard.assert_intact("gemm act")


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("m,n,k", [(1, 16, 64), (130, 144, 128), (257, 768, 256), (515, 528, 128), (1000, 272, 384)])
def test_small_gemm_guard_bands(ops, tile, m, n, k):
    from candidate_reranking_cir_amd import lib
    lib.set_tuning(lib.TUNE_GEMM_TILE, tile)
    try:
        for out_dtype, with_res in ((torch.bfloat16, False), (torch.float32, False), (torch.float32, True), (torch.float16, False), (torch.float16, True)):
            amp = 1 if out_dtype == torch.float16 else 3
            a = _ints((m, k), -amp, amp, torch.bfloat16, seed=1)
            w = _ints((n, k), -amp, amp, torch.bfloat16, seed=2)
            bias = _ints((n,), -5, 5, torch.float32, seed=3)
            guard = Guarded(m, n, out_dtype)
            res = guard.fill_view(_ints((m, n), -7, 7, out_dtype, seed=4)) if with_res else None
            ref = a.float() @ w.float().T + bias + (res.float() if with_res else 0)
            out = ops.gemm(a, w, bias, residual=res, out_dtype=out_dtype, out=guard.view)
            torch.cuda.synchronize()
            assert torch.equal(out.float(), ref.to(out_dtype).float())
            guard.assert_intact(f"gemm tile {tile} {m}x{n}x{k} {out_dtype}")
    finally:
        lib.set_tuning(lib.TUNE_GEMM_TILE, 0)


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, scale):
    b1, b0, lq, d = q.shape
    h = d // 64
    qh = q.float().reshape(b1, b0, lq, h, 64).transpose(2, 3)
    kh = k.float().reshape(b1, b0, -1, h, 64).transpose(2, 3)
    vh = v.float().reshape(b1, b0, -1, h, 64).transpose(2, 3)
    return (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).transpose(2, 3).reshape(b1, b0, lq, d)


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("lq,lk", [(1, 1), (5, 7), (32, 197), (33, 70), (197, 197), (100, 33), (577, 577), (300, 608), (70, 640), (1, 32)])
def test_attention_guard_bands(ops, dtype, lq, lk):
    """Ragged Lq / Lk in both kernels (shared-LDS and streamed): the output rows of each (item, branch) sit between
    canary rows and canary columns (row stride and item strides larger than the data)."""
    b1, b0, h = 3, 2, 2
    d = h * 64
    g = torch.Generator(device="cpu").manual_seed(lq * 1000 + lk)
    q = torch.randn((b1, b0, lq, d), generator=g).to(dtype).cuda()
    k = torch.randn((b1, b0, lk, d), generator=g).to(dtype).cuda()
    v = torch.randn((b1, b0, lk, d), generator=g).to(dtype).cuda()
    guard = Guarded(lq, d, dtype, pr=3, pc=32, batch=b1 * b0)
    out = guard.big.view(b1, b0, lq + 6, d + 64)[:, :, 3:3 + lq, 32:32 + d]
    ops.attention(q, k, v, out, 0.125)
    torch.cuda.synchronize()
    torch.testing.assert_close(out.float(), _attn_ref(q, k, v, 0.125), atol=2e-2 if dtype == torch.bfloat16 else 4e-3, rtol=0)
    guard.assert_intact(f"attention {lq}x{lk}")


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("t,n,d", [(3, 5, 128), (7, 197, 768), (600, 40, 256)])
def test_cls_cross_attention_guard_bands(ops, dtype, t, n, d):
    """cir_cls_cross_attention writes exactly T x 32 x D elements: canaries before and after the result stay intact, and
    reading past the last token row (clamped ragged tiles) never shows up in the result."""
    g = torch.Generator(device="cpu").manual_seed(t + n)
    x = torch.randn((t, n, d), generator=g).to(dtype).cuda()
    qp = (torch.randn((t, 32, d), generator=g) * 0.3).to(dtype).cuda()
    buf, out = _flat_guard(t * 32 * d, dtype)
    ops.cls_cross_attention(x, qp, 0.125, out=out.view(t, 32, d))
    torch.cuda.synchronize()
    ref = torch.softmax(torch.einsum("trd,tnd->trn", qp.float(), x.float()) * 0.125, -1) @ x.float()
    torch.testing.assert_close(out.view(t, 32, d).float(), ref, atol=1.5e-2 if dtype == torch.bfloat16 else 2e-3, rtol=0)
    assert _flat_intact(buf, t * 32 * d, dtype)


# ------------------------------------------------------------------------------------------------ row kernels
def _flat_guard(n_elems, dtype, slack=4096):
    buf = torch.empty((n_elems + 2 * slack,), dtype=dtype, device="cuda")
    buf.view(_INT[dtype]).fill_(CANARY[dtype])
    return buf, buf[slack:slack + n_elems]


def _flat_intact(buf, n_elems, dtype, slack=4096):
    bits = buf.view(_INT[dtype])
    return bool((bits[:slack] == CANARY[dtype]).all()) and bool((bits[slack + n_elems:] == CANARY[dtype]).all())


@pytest.mark.parametrize("sdt", [torch.float32, torch.float16], ids=["stream32", "stream16"])
@pytest.mark.parametrize("rows,cols", [(1, 768), (203, 768), (5, 64), (77, 1024), (1001, 128)])
def test_layernorm_guard_bands(ops, rows, cols, sdt):
    g = torch.Generator(device="cpu").manual_seed(rows)
    x = torch.randn((rows, cols), generator=g).to(sdt).cuda()
    res = torch.randn((rows, cols), generator=g).to(sdt).cuda()
    gam, bet = torch.randn((cols,), generator=g).cuda(), torch.randn((cols,), generator=g).cuda()
    b32, y32 = _flat_guard(rows * cols, sdt)
    b16, y16 = _flat_guard(rows * cols, torch.bfloat16)
    ops.layernorm(x, gam, bet, 1e-6, residual=res, out32=y32.view(rows, cols), out16=y16.view(rows, cols))
    torch.cuda.synchronize()
    ref = F.layer_norm(x.float() + res.float(), (cols,), gam, bet, 1e-6)
    if sdt == torch.float32:
        torch.testing.assert_close(y32.view(rows, cols), ref, atol=2e-5, rtol=1e-5)
    else:
        torch.testing.assert_close(y32.view(rows, cols).float(), ref, atol=1e-3, rtol=1e-3)          # <= 1 ulp of fp16
    assert _flat_intact(b32, rows * cols, sdt) and _flat_intact(b16, rows * cols, torch.bfloat16)


def test_gather_patchify_assemble_topk_guard_bands(ops):
    from candidate_reranking_cir_amd import lib
    c = lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    # gather_rows: 7 rows of 5 x 64 elements, fp32 -> bf16
    bank = torch.randn((9, 5, 64), device="cuda")
    idx = torch.tensor([3, 3, 0, 8, 1, 7, 2], device="cuda")
    buf, dst = _flat_guard(7 * 320, torch.bfloat16)
    lib.check(c.cir_gather_rows(bank.data_ptr(), lib.CIR_F32, idx.data_ptr(), dst.data_ptr(), lib.CIR_BF16, 7, 320, 9, stream), "gather")
    torch.cuda.synchronize()
    assert torch.equal(dst.view(7, 5, 64), bank[idx].bfloat16()) and _flat_intact(buf, 7 * 320, torch.bfloat16)
    # patchify: 3 images 64 x 64, patch 16 -> (48, 768)
    img = torch.randn((3, 3, 64, 64), device="cuda")
    buf, dst = _flat_guard(48 * 768, torch.bfloat16)
    lib.check(c.cir_patchify(img.data_ptr(), lib.CIR_F32, dst.data_ptr(), lib.CIR_BF16, 3, 3, 64, 64, 16, stream), "patchify")
    torch.cuda.synchronize()
    ref = F.unfold(img, kernel_size=16, stride=16).transpose(1, 2).reshape(-1, 768).bfloat16()
    assert torch.equal(dst.view(48, 768), ref) and _flat_intact(buf, 48 * 768, torch.bfloat16)
    # vit_assemble: (3, 17, 128)
    proj, cls, pos = torch.randn((48, 128), device="cuda"), torch.randn((128,), device="cuda"), torch.randn((17, 128), device="cuda")
    buf, dst = _flat_guard(3 * 17 * 128, torch.float32)
    lib.check(c.cir_vit_assemble(proj.data_ptr(), cls.data_ptr(), pos.data_ptr(), dst.data_ptr(), lib.CIR_F32, 3, 16, 128, stream), "assemble")
    torch.cuda.synchronize()
    assert torch.equal(dst.view(3, 17, 128), torch.cat([cls.expand(3, 1, 128), proj.view(3, 16, 128)], 1) + pos[None])
    assert _flat_intact(buf, 3 * 17 * 128, torch.float32)
    # topk_desc: (5, 205) int64 indices
    logits = torch.randn((5, 205), device="cuda")
    buf, dst = _flat_guard(5 * 205, torch.int64, slack=512)
    lib.check(c.cir_topk_desc(logits.data_ptr(), dst.data_ptr(), 5, 205, stream), "topk")
    torch.cuda.synchronize()
    assert torch.equal(dst.view(5, 205), torch.argsort(logits, dim=-1, descending=True, stable=True))
    assert _flat_intact(buf, 5 * 205, torch.int64, slack=512)
    # small_linear: (101, 2) fp32
    x, w = torch.randn((101, 768), device="cuda").bfloat16(), torch.randn((2, 768), device="cuda").bfloat16()
    buf, dst = _flat_guard(202, torch.float32, slack=256)
    lib.check(c.cir_small_linear(x.data_ptr(), 768, w.data_ptr(), None, dst.data_ptr(), 101, 2, 768, lib.CIR_BF16, stream), "small_linear")
    torch.cuda.synchronize()
    torch.testing.assert_close(dst.view(101, 2), x.float() @ w.float().T, atol=1e-3, rtol=1e-4)
    assert _flat_intact(buf, 202, torch.float32, slack=256)


# ------------------------------------------------------------------------------------------------ attention, many workgroups
@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("masked", [False, True], ids=["nomask", "mask"])
@pytest.mark.parametrize("lq,lk", [(197, 197), (40, 33), (224, 256), (500, 200), (33, 70), (64, 8), (577, 577), (640, 600), (545, 321)])
def test_attention_many_workgroups(ops, dtype, masked, lq, lk):
    """More workgroups than fit the chip at once (720 > 2 x 256 slots), ragged Lq / Lk, with and without a key mask:
    against fp32 torch, output between canaries, and bit-identical to the streamed kernel's tile arithmetic
    (cir_set_tuning forces it: both kernels share the tile update, only the K/V delivery differs).  More than 16 query tiles
    (577 / 640 / 545 queries): the staged kernel deals them to the wave count that balances the four SIMDs (16 / 16 / 12 waves here)
    and stages 577+ keys in one batch of 5 chunks per thread."""
    from candidate_reranking_cir_amd import lib
    b1, b0, h = (30, 2, 12) if lq < 512 else (11, 2, 12)
    d = h * 64
    g = torch.Generator(device="cpu").manual_seed(lq * 131 + lk)
    q = torch.randn((b1, b0, lq, d), generator=g).to(dtype).cuda()
    k = torch.randn((b1, b0, lk, d), generator=g).to(dtype).cuda()
    v = torch.randn((b1, b0, lk, d), generator=g).to(dtype).cuda()
    mask = None
    if masked:
        valid = torch.randint(1, lk + 1, (b1, b0), generator=g)
        mask = ((torch.arange(lk)[None, None] >= valid[..., None]).float() * -10000.0).cuda()
    guard = Guarded(lq, d, dtype, pr=3, pc=32, batch=b1 * b0)
    out = guard.big.view(b1, b0, lq + 6, d + 64)[:, :, 3:3 + lq, 32:32 + d]
    ops.attention(q, k, v, out, 0.125, mask)
    torch.cuda.synchronize()
    guard.assert_intact(f"attention {lq}x{lk}")
    s = torch.einsum("abqhd,abkhd->abhqk", q.float().view(b1, b0, lq, h, 64), k.float().view(b1, b0, lk, h, 64)) * 0.125
    if mask is not None:
        s = s + mask[:, :, None, None, :]
    ref = torch.einsum("abhqk,abkhd->abqhd", torch.softmax(s, -1), v.float().view(b1, b0, lk, h, 64)).reshape(b1, b0, lq, d)
    torch.testing.assert_close(out.float(), ref, atol=2e-2 if dtype == torch.bfloat16 else 4e-3, rtol=0)
    lib.set_tuning(lib.TUNE_ATTN_SHARED_MAX, -1)
    try:
        out2 = torch.empty((b1, b0, lq, d), dtype=dtype, device="cuda")
        ops.attention(q, k, v, out2, 0.125, mask)
        torch.cuda.synchronize()
    finally:
        lib.set_tuning(lib.TUNE_ATTN_SHARED_MAX, 0)
    assert torch.equal(out2, out)


def test_attention_kv_bank_many_workgroups(ops):
    """kv_index with more workgroups than CU slots."""
    t, l, nk, h, rows = 300, 40, 37, 2, 5
    d = h * 64
    g = torch.Generator(device="cpu").manual_seed(5)
    q = torch.randn((t, 2, l, d), generator=g).bfloat16().cuda()
    bank = torch.randn((rows, nk, 4, d), generator=g).bfloat16().cuda()
    idx = torch.randint(0, rows, (t,), generator=g).cuda()
    k4, v4 = bank[:, :, 0::2].permute(0, 2, 1, 3), bank[:, :, 1::2].permute(0, 2, 1, 3)
    out_a, out_b = torch.empty_like(q), torch.empty_like(q)
    ops.attention(q, k4, v4, out_a, 0.125, kv_index=idx)           # 1200 workgroups, 2 query tiles, 64 padded keys
    gth = bank[idx]
    ops.attention(q, gth[:, :, 0::2].permute(0, 2, 1, 3), gth[:, :, 1::2].permute(0, 2, 1, 3), out_b, 0.125)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)


# ------------------------------------------------------------------------------------------------ round 5 kernels
@pytest.mark.parametrize("m,n,k", [(333, 400, 224), (1, 16, 32), (257, 768, 768), (5000, 2304, 64)])
@pytest.mark.parametrize("variant", ["plain", "gelu", "res_alias"])
def test_gemm_f32_guard_bands(ops, m, n, k, variant):
    """The fp32-operand instantiation of the 128-tile GEMM (exact mode): ragged M / N edges, output view inside canaries."""
    a = _ints((m, k), -3, 3, torch.float32, seed=m + k)
    w = _ints((n, k), -3, 3, torch.float32, seed=n + k + 1)
    bias = _ints((n,), -5, 5, torch.float32, seed=3)
    guard = Guarded(m, n, torch.float32)
    ref = a @ w.T + bias
    if variant == "res_alias":
        res = _ints((m, n), -4, 4, torch.float32, seed=9)
        out = guard.fill_view(res)
        ops.gemm(a, w, bias, residual=out, out=out)
        ref = ref + res
    else:
        ops.gemm(a, w, bias, act=1 if variant == "gelu" else 0, out=guard.view)
        if variant == "gelu":
            ref = F.gelu(ref)
    torch.cuda.synchronize()
    guard.assert_intact(f"gemm f32 {variant} {m}x{n}x{k}")
    assert torch.equal(guard.view, ref) if variant != "gelu" else (guard.view - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("b1,h,lq,lk", [(3, 2, 33, 65), (2, 12, 197, 197), (5, 1, 1, 9), (2, 3, 32, 577)])
def test_attention_f32_guard_bands(ops, b1, h, lq, lk):
    """attn_f32_kernel stores through per-lane row pointers with a ragged last query tile: the (B, Lq, H 64) output sits inside canaries."""
    d = 64 * h
    g = torch.Generator(device="cpu").manual_seed(lq * 7 + lk)
    q, k, v = (torch.randn((b1, 1, n_, d), generator=g).cuda() for n_ in (lq, lk, lk))
    guard = Guarded(b1 * lq, d, torch.float32)
    out = guard.view.unflatten(0, (b1, 1, lq))
    ops.attention(q, k, v, out, 0.125)
    torch.cuda.synchronize()
    guard.assert_intact(f"attention f32 {b1}x{h} {lq}x{lk}")
    ref = F.scaled_dot_product_attention(q.view(b1, lq, h, 64).transpose(1, 2), k.view(b1, lk, h, 64).transpose(1, 2), v.view(b1, lk, h, 64).transpose(1, 2))
    assert (out.view(b1, lq, h, 64).transpose(1, 2) - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("t_n,l,n", [(3, 32, 197), (5, 7, 197), (2, 31, 224), (9, 1, 33), (2, 32, 577), (5, 13, 401)])
def test_folded_cross_attention_guard_bands(ops, dtype, t_n, l, n):
    """xattn_fold_kernel writes token rows < L of a (T, L, 2, 768) tensor from 48-row wave tiles (rows beyond L are computed on zero queries and
    must not be stored), reads X rows clamped to N - 1 and weight fragments through buffer descriptors: output inside canaries, and the
    tokens / weights sit at the END of their allocations (a read past them would hit the next allocation's canary NaNs and poison the result)."""
    D = 768
    g = torch.Generator(device="cpu").manual_seed(t_n * 31 + l)
    r = lambda shape, s: (torch.randn(shape, generator=g) * s).to(dtype).cuda()
    q, x = r((2, t_n * l, D), 1.0), r((t_n, n, D), 1.0)
    wk, wv, bv = r((2, D, D), 0.03), r((2, D, D), 0.03), torch.randn((2, D), generator=g).cuda()
    guard = Guarded(t_n * l, 2 * D, dtype)
    out = guard.view.unflatten(0, (t_n, l)).unflatten(2, (2, D))
    ops.cross_attention_folded(q, x, ops.fold_pack_key(wk), ops.fold_pack_value(wv), bv, out, l, 0.125)
    torch.cuda.synchronize()
    guard.assert_intact(f"folded cross-attention T {t_n} L {l} N {n}")
    assert torch.isfinite(out.float()).all()
    # against the projected path of the library on the same inputs
    wkv, bkv = torch.cat([wk[0], wv[0], wk[1], wv[1]]), torch.cat([torch.zeros(D, device="cuda"), bv[0], torch.zeros(D, device="cuda"), bv[1]])
    kv = ops.gemm(x.view(t_n * n, D), wkv, bkv).view(t_n, n, 4, D)
    o2 = torch.empty((t_n, l, 2, D), dtype=dtype, device="cuda")
    ops.attention(q.view(2, t_n, l, D).permute(1, 0, 2, 3), kv[:, :, 0::2].permute(0, 2, 1, 3), kv[:, :, 1::2].permute(0, 2, 1, 3), o2.permute(0, 2, 1, 3), 0.125)
    assert (out.float() - o2.float()).abs().max().item() < (6.5e-2 if dtype == torch.bfloat16 else 4e-3)     # (two roundings: up to two bf16 ulps of 2^-5 at |ctx| ~ 4)
