"""The "split8" path of the text32 precision mode on a real MI355X (round 6): cir_split8 rows bit for bit against their definition,
cir_gemm_split8 (fp16 MFMA on the leading terms + block-scaled fp8 MFMA on the two correction products, one fp32 accumulator) against an
fp64 evaluation of the SAME three products, against the exact fp32 Linear it stands for (nlvr_encoder.py / med.py nn.Linear in the
reference's fp32: validate_stage2.py:140-141), tile-choice bit invariance, the split8 output epilogue, and bounds."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from candidate_reranking_cir_amd import ops as _ops
    return _ops


def _e4m3(x):
    return x.clamp(-448.0, 448.0).cpu().to(torch.float8_e4m3fn)


def _terms(x):
    """(hi fp16, lo8 e4m3 of (x - hi) 2^12, hi8 e4m3 of hi) of the definition in include/cirrank.h, on the CPU"""
    x = x.float().cpu()
    hi = x.half()
    return hi, _e4m3((x - hi.float()) * 4096.0), _e4m3(hi.float())


def _activations(m, k, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((m, k), generator=g) * 1.3
    x[:, 3] = x[:, 3] * 6.0 + 24.0                      # LayerNorm outputs of checkpoints: a few channels with a gain and an offset
    x[:, k - 5] = x[:, k - 5] * 6.0 - 24.0
    x[::7, 11] *= 1e-3                                  # and tiny values
    return x


def test_split8_rows_bit_for_bit(ops):
    m, k = 1031, 768
    x = _activations(m, k, 1)
    x[5, 17], x[6, 18], x[7, 19] = 470.0, -9000.0, 60000.0         # beyond e4m3's 448: the fp8 terms clamp, the fp16 term does not
    s = ops.split8(x.cuda())
    rows = s.rows.cpu()
    hi, lo8, hi8 = _terms(x)
    assert torch.equal(rows[:, :2 * k].contiguous().view(torch.float16), hi)
    assert torch.equal(rows[:, 2 * k:3 * k], lo8.view(torch.uint8))
    assert torch.equal(rows[:, 3 * k:], hi8.view(torch.uint8))
    assert int((hi8.view(torch.uint8) & 0x7f).max()) <= 0x7e       # never the NaN encoding
    # strided rows (the CLS rows of a (T, L, D) tensor) and an activation
    x3 = _activations(64 * 4, k, 2).view(64, 4, k).cuda()
    s1 = ops.split8(x3[:, 0, :].unsqueeze(0))
    assert torch.equal(s1.rows.view(64, 4 * k).cpu(), ops.split8(x3[:, 0, :].contiguous()).rows.cpu())
    x = _activations(m, k, 4)                                               # (no element beyond the lo term's clamp: |y| < 224)
    y = ops.split8(x.cuda(), ops.ACT_GELU).float().cpu().double()
    ref = F.gelu(x.double())
    # hi + lo carries 11 + 4 bits; below |y| ~ 2^-7 the lo term runs into e4m3's subnormals (its absolute size there: < 2^-19)
    assert bool(((y - ref).abs() <= 3.2e-5 * ref.abs() + 6e-7).all()), ((y - ref).abs() / (ref.abs() + 1e-3)).max().item()


def _emulate(a, w, e1, e2, bias, residual):
    """fp64 evaluation of the three products the kernel forms (same roundings of every factor)"""
    a_hi, a_lo8, a_hi8 = _terms(a)
    w = w.float().cpu()
    w_hi = w.half()
    w_hi8 = _e4m3(w_hi.float() * 2.0 ** e1).double() * 2.0 ** -e1
    w_lo8 = _e4m3((w - w_hi.float()) * 2.0 ** e2).double() * 2.0 ** -e2
    y = a_hi.double() @ w_hi.double().transpose(-1, -2) + (a_lo8.double() / 4096.0) @ w_hi8.transpose(-1, -2) + a_hi8.double() @ w_lo8.transpose(-1, -2)
    if bias is not None:
        y = y + bias.double().cpu().unsqueeze(-2) if bias.dim() == 2 else y + bias.double().cpu()
    if residual is not None:
        y = y + residual.double().cpu()
    return y


@pytest.mark.parametrize("m,n,k,tile,res", [(300, 768, 768, 128, False), (4099, 2304, 768, 256, False), (5000, 768, 3072, 256, True),
                                            (777, 3072, 768, 128, True), (6720, 768, 1536, 0, False), (70000, 768, 768, 0, True),
                                            (4099, 320, 256, 256, True), (515, 48, 512, 128, False)])      # ragged column tiles, the smallest depth
def test_gemm_split8_products_and_accuracy(ops, m, n, k, tile, res):
    from candidate_reranking_cir_amd import lib
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    a = _activations(m, k, m)
    w = torch.randn((n, k), generator=g) * 0.03
    w[7] *= 20.0                                                    # output projections of checkpoints: a few 20x rows
    b = torch.randn((n,), generator=g) * 0.1
    r = torch.randn((m, n), generator=g) if res else None
    w_d = ops.split_weight8(w.cuda())
    _, e1, e2 = w_d._split8
    lib.set_tuning(lib.TUNE_GEMM_TILE, tile)
    try:
        got = ops.gemm(a.cuda(), w_d, b.cuda(), residual=None if r is None else r.cuda()).double().cpu()
    finally:
        lib.set_tuning(lib.TUNE_GEMM_TILE, 0)
    emu = _emulate(a, w, e1, e2, b, r)
    exact = a.double() @ w.double().t() + b.double() + (r.double() if res else 0.0)
    scale = (a.double().abs() @ w.double().abs().t()).mean(0)                # per output feature: size of a row's sum of |terms|
    assert bool(((got - emu).abs().max(0).values < 3e-6 * scale * (k / 768) ** 0.5).all()), (got - emu).abs().max().item()   # fp32 accumulation order only
    rms = lambda e: e.pow(2).mean().sqrt().item()
    one = (a.half().double() @ w.half().double().t() + b.double() + (r.double() if res else 0.0))
    assert rms(got - exact) < rms(one - exact) / 12.0, (rms(got - exact), rms(one - exact))                            # >= 3.5 bits better than one fp16 product (measured ~2^5)
    assert rms(got - exact) < 4e-5 * rms(exact - (r.double() if res else 0.0)), rms(got - exact)


def test_gemm_split8_tile_choice_and_batching_bit_invariant(ops):
    from candidate_reranking_cir_amd import lib
    g = torch.Generator(device="cpu").manual_seed(3)
    nb, m, n, k = 2, 20000, 768, 768
    a = torch.stack([_activations(m, k, 5), _activations(m, k, 6)]).cuda()
    w = ops.split_weight8((torch.randn((nb, n, k), generator=g) * 0.03).cuda())
    b = (torch.randn((nb, n), generator=g) * 0.1).cuda()
    r = torch.randn((nb, m, n), generator=g).cuda()
    sa = ops.split8(a)
    outs = []
    for tile in (128, 256):
        lib.set_tuning(lib.TUNE_GEMM_TILE, tile)
        try:
            outs.append(ops.gemm(sa, w, b, residual=r))
            outs.append(ops.gemm(sa, w, b, act=ops.ACT_GELU).rows)
        finally:
            lib.set_tuning(lib.TUNE_GEMM_TILE, 0)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    # a row block scored alone = the same rows inside the batch
    part = ops.gemm(ops.split8(a[1, 300:900].contiguous()), ops.split_weight8(w[1].clone()), b[1], residual=r[1, 300:900].contiguous())
    w1 = ops.split_weight8(w[1].clone())
    if w1._split8[1:] == w._split8[1:]:                      # same per-tensor exponents (they are chosen over the whole batch of weights)
        assert torch.equal(part, outs[0][1, 300:900])


@pytest.mark.parametrize("m,n,k,tile", [(1000, 3072, 768, 128), (9001, 3072, 768, 256), (70000, 3072, 768, 0), (5000, 320, 768, 256), (700, 64, 256, 128)])
def test_gemm_split8_gelu_rows_out(ops, m, n, k, tile):
    """fc1 -> fc2: the GELU output leaves the GEMM as split8 rows; they equal the split of the same GEMM's fp32 output through the
    stand-alone pass (same accumulators, same GELU), and feed the next GEMM."""
    from candidate_reranking_cir_amd import lib
    g = torch.Generator(device="cpu").manual_seed(m)
    a = _activations(m, k, m + 1).cuda()
    w1 = ops.split_weight8((torch.randn((n, k), generator=g) * 0.03).cuda())
    b1 = (torch.randn((n,), generator=g) * 0.1).cuda()
    lib.set_tuning(lib.TUNE_GEMM_TILE, tile)
    try:
        f = ops.gemm(a, w1, b1, act=ops.ACT_GELU)
        pre = ops.gemm(a, w1, b1)
    finally:
        lib.set_tuning(lib.TUNE_GEMM_TILE, 0)
    assert isinstance(f, ops.Split8Operand) and f.rows.shape == (m, 4 * n)
    two = ops.split8(pre, ops.ACT_GELU)
    same = (f.rows == two.rows).float().mean().item()
    assert same > 0.9999, same                                          # (a contraction may differ between the two kernels' GELU code)
    ref = F.gelu(pre.double())
    assert bool(((f.float().double() - ref).abs() <= 3.2e-5 * ref.abs() + 6e-7).all())
    if n % 256 == 0:                                                    # (the next GEMM's depth: whole K-tile pairs per segment)
        w2 = ops.split_weight8((torch.randn((768, n), generator=g) * 0.02).cuda())
        y = ops.gemm(f, w2, None, residual=a)
        exact = F.gelu(pre.double()) @ w2.double().t() + a.double()
        assert (y.double() - exact).pow(2).mean().sqrt().item() < 4e-5 * (exact - a.double()).pow(2).mean().sqrt().item()


def test_split8_writes_stay_inside_their_rows(ops):
    """canary-filled allocations around ragged outputs (GPU AddressSanitizer is not available on this pool)"""
    from candidate_reranking_cir_amd import lib
    m, n, k = 2 * 256 + 37, 3072, 768
    g = torch.Generator(device="cpu").manual_seed(9)
    a = _activations(m, k, 9).cuda()
    w = ops.split_weight8((torch.randn((n, k), generator=g) * 0.03).cuda())
    for tile in (128, 256):
        lib.set_tuning(lib.TUNE_GEMM_TILE, tile)
        try:
            big = torch.full((m + 64, n), 7.25, dtype=torch.float32, device="cuda")
            ops.gemm(a, w, None, out=big[32:32 + m])
            assert bool((big[:32] == 7.25).all()) and bool((big[32 + m:] == 7.25).all())
            sa = ops.split8(a)
            canary = torch.full((m + 64, 4 * k), 0xA5, dtype=torch.uint8, device="cuda")
            ops.split8(a, out=canary[32:32 + m])
            assert bool((canary[:32] == 0xA5).all()) and bool((canary[32 + m:] == 0xA5).all()) and torch.equal(canary[32:32 + m], sa.rows)
        finally:
            lib.set_tuning(lib.TUNE_GEMM_TILE, 0)


def test_layernorm_split8_stream_and_rows(ops):
    """the LayerNorm that writes the next GEMM's split8 operand itself: stream copy = cir_layernorm's bit for bit, rows = cir_split8 of it"""
    g = torch.Generator(device="cpu").manual_seed(11)
    rows, cols = 5003, 768
    m = (torch.randn((rows, cols), generator=g) * 2.0).cuda()                       # shared (broadcast) input, per-branch residual and affine
    res = (torch.randn((2, rows, cols), generator=g) * 3.0 + 0.5).cuda()
    gamma = (1.0 + 0.3 * torch.randn((2, cols), generator=g)).cuda()
    beta = (0.2 * torch.randn((2, cols), generator=g)).cuda()
    gamma[:, 5] = 6.0; beta[:, 5] = 4.0
    ys, sp = ops.layernorm_split8(m, gamma, beta, 1e-12, residual=res)
    y0, _ = ops.layernorm(m, gamma, beta, 1e-12, residual=res, want32=True, dtype16=None, stream_dtype=torch.float32)
    assert ys.shape == (2, rows, cols) and torch.equal(ys, y0)
    assert torch.equal(sp.rows, ops.split8(y0).rows)
    ref = F.layer_norm((m.double() + res.double()), (cols,)) * gamma.double()[:, None, :] + beta.double()[:, None, :]
    assert (ys.double() - ref).abs().max().item() < 2e-5
    y1, sp1 = ops.layernorm_split8(res[0], gamma[0], beta[0], 1e-12, want_stream=False)       # 2-D form, no stream copy
    assert y1 is None and sp1.rows.shape == (rows, 4 * cols)
    assert torch.equal(sp1.rows, ops.split8(ops.layernorm(res[0], gamma[0], beta[0], 1e-12, want32=True, dtype16=None, stream_dtype=torch.float32)[0]).rows)


@pytest.mark.parametrize("items,lq,lk,masked", [(300, 32, 32, True), (37, 1, 32, True), (5, 40, 77, False)])
def test_attention_split8_rows(ops, items, lq, lk, masked):
    """cir_attention_split8: fp32 q / k / v -> context as split8 rows.  Its kernel forms both products as three-term sums of (hi, lo) fp16 pairs
    on the fp16 MFMA (attn_split32_kernel): within the rows' own 15 bits of the f32-input MFMA form and of fp64; the f32-input form of the same
    entry point (tuning -2) writes exactly cir_split8 of cir_attention's fp32 output."""
    from candidate_reranking_cir_amd import lib
    g = torch.Generator(device="cpu").manual_seed(items)
    d = 768
    qkv = torch.randn((2, items, max(lq, lk), 3 * d), generator=g).cuda()
    q, k, v = qkv[:, :, :lq, :d], qkv[:, :, :lk, d:2 * d], qkv[:, :, :lk, 2 * d:]
    mask = None
    if masked:
        mask = torch.zeros((2, items, lk)).cuda()
        mask[:, ::3, lk - 5:] = -10000.0
    out = torch.empty((2, items, lq, d), dtype=torch.float32, device="cuda")
    ops.attention(q, k, v, out, 0.125, mask)
    sp = ops.attention_split8(q, k, v, 0.125, mask)
    assert sp.rows.shape == (2, items, lq, 4 * d)
    lib.set_tuning(lib.TUNE_ATTN_SHARED_MAX, -2)
    try:
        sp32 = ops.attention_split8(q, k, v, 0.125, mask)
    finally:
        lib.set_tuning(lib.TUNE_ATTN_SHARED_MAX, 0)
    assert torch.equal(sp32.rows, ops.split8(out).rows)
    h = d // 64
    qh, kh, vh = (t.double().reshape(2, items, -1, h, 64).permute(0, 1, 3, 2, 4) for t in (q, k, v))
    sc = qh @ kh.transpose(-1, -2) * 0.125
    if mask is not None:
        sc = sc + mask.double()[:, :, None, None, :]
    ref = (torch.softmax(sc, -1) @ vh).permute(0, 1, 3, 2, 4).reshape(2, items, lq, d)
    got = sp.float().double()
    assert bool(((got - ref).abs() <= 3.2e-5 * ref.abs() + 2e-6).all()), (got - ref).abs().max().item()
    hi = sp.rows[..., :2 * d].contiguous().view(torch.float16).double()         # the fp16 terms alone: three-term products are far inside their rounding
    assert (hi - ref).abs().max().item() <= (out.double() - ref).abs().max().item() + 1.1 * 2.0 ** -11 * ref.abs().max().item()
    assert (got - out.double()).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())
