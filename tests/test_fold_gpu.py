"""cir_cross_attention_folded (round 5): the two-branch cross-attention with the key / value projections folded out of the image-token side,
against an fp64 restatement of the reference's arithmetic (nlvr_encoder.py:150-168, 183-217: K = X W_k^T + b_k, V = X W_v^T + b_v, softmax(q K^T / 8) V)
and against the projected path of this library (cir_gemm_bias_act K|V + cir_attention) on the same 16-bit inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D, H = 768, 12


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from candidate_reranking_cir_amd import ops as _ops
    return _ops


def _rand(shape, scale, seed, dtype):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def _reference(q, x, wk, bk, wv, bv, l):
    """fp64, as the reference writes it: per branch b and candidate t, heads of 64."""
    t_n, n, _ = x.shape
    out = torch.empty((t_n, l, 2, D), dtype=torch.float64)
    for b in (0, 1):
        k = (x.double() @ wk[b].double().T + bk[b].double()).view(t_n, n, H, 64).permute(0, 2, 1, 3)
        v = (x.double() @ wv[b].double().T + bv[b].double()).view(t_n, n, H, 64).permute(0, 2, 1, 3)
        qq = q[b].double().view(t_n, l, H, 64).permute(0, 2, 1, 3)
        p = torch.softmax(qq @ k.transpose(-1, -2) / 8.0, -1)
        out[:, :, b] = (p @ v).permute(0, 2, 1, 3).reshape(t_n, l, D)
    return out


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("t_n,l,n", [(3, 32, 197), (5, 11, 197), (2, 32, 224), (4, 1, 50), (2, 17, 17), (2, 32, 577), (3, 9, 300), (1, 32, 608), (5, 30, 225)])
def test_folded_cross_attention_against_fp64_and_projected_path(ops, dtype, t_n, l, n):
    q = _rand((2, t_n * l, D), 1.0, 1, dtype)
    x = _rand((t_n, n, D), 1.0, 2, dtype)
    wk, wv = _rand((2, D, D), 0.03, 3, dtype), _rand((2, D, D), 0.03, 4, dtype)
    bk, bv = _rand((2, D), 0.5, 5, torch.float32), _rand((2, D), 0.5, 6, torch.float32)
    ref = _reference(q, x, wk, bk, wv, bv, l)
    out = torch.empty((t_n, l, 2, D), dtype=dtype, device="cuda")
    ops.cross_attention_folded(q.cuda(), x.cuda(), ops.fold_pack_key(wk).cuda(), ops.fold_pack_value(wv).cuda(), bv.cuda(), out, l, 0.125)
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    # the projected path on the same inputs: [K0 V0 K1 V1] GEMM + attention
    wkv = torch.cat([wk[0], wv[0], wk[1], wv[1]]).cuda()
    bkv = torch.cat([bk[0], bv[0], bk[1], bv[1]]).cuda()
    kv = ops.gemm(x.cuda().view(t_n * n, D), wkv, bkv).view(t_n, n, 4, D)
    o2 = torch.empty((t_n, l, 2, D), dtype=dtype, device="cuda")
    qc = q.cuda().view(2, t_n, l, D).permute(1, 0, 2, 3)
    ops.attention(qc, kv[:, :, 0::2].permute(0, 2, 1, 3), kv[:, :, 1::2].permute(0, 2, 1, 3), o2.permute(0, 2, 1, 3), 0.125)
    err2 = (o2.cpu().double() - ref).abs().max().item()
    print(f"\n[folded cross-attention {dtype} T {t_n} L {l} N {n}] max|err| vs fp64: folded {err:.2e}, projected {err2:.2e} (|ctx| max {ref.abs().max():.2f})")
    tol = 4e-2 if dtype == torch.bfloat16 else 6e-3
    assert err < tol and err < 2.5 * err2 + 1e-3


def test_folded_cross_attention_exact_small_integers(ops):
    """Attention over identical keys is the mean of the values whatever the scores: with integer X (all rows equal) and integer W_v the folded
    chain (P X, then W_v) is exact in fp16 - any k-slot / permutation slip in G3 / G4 shows up as a wrong integer."""
    t_n, l, n = 2, 32, 197
    g = torch.Generator().manual_seed(3)
    row = torch.randint(-2, 3, (t_n, 1, D), generator=g).float()
    x = row.expand(t_n, n, D).contiguous().half()
    q = _rand((2, t_n * l, D), 1.0, 1, torch.float16)
    wk = _rand((2, D, D), 0.03, 2, torch.float16)
    wv = torch.randint(-1, 2, (2, D, D), generator=g).half()
    bv = torch.randint(-3, 4, (2, D), generator=g).float()
    out = torch.empty((t_n, l, 2, D), dtype=torch.float16, device="cuda")
    ops.cross_attention_folded(q.cuda(), x.cuda(), ops.fold_pack_key(wk).cuda(), ops.fold_pack_value(wv).cuda(), bv.cuda(), out, l, 0.125)
    want = torch.stack([row[:, 0].double() @ wv[b].double().T + bv[b].double() for b in (0, 1)], dim=1)      # (T, 2, D)
    err = (out.cpu().double() - want[:, None].expand(t_n, l, 2, D)).abs().max().item()
    print(f"\n[folded cross-attention, constant keys] max|err| {err:.2e} (values up to {want.abs().max():.0f})")
    assert err < 0.13          # sums of ~50 terms up to ~100: one fp16 ulp there is 0.06 (P X is exact, the row sum of P rounds)


@pytest.mark.parametrize("n", [197, 577])
def test_folded_cross_attention_scores_follow_the_keys(ops, n):
    """One-hot attention: with a huge score on one key the output must be THAT key's projected value - exercises G1 / G2's k-slot maps and the
    key-block layout of the softmax, per head (each head is steered to a different key)."""
    t_n, l = 1, 32
    x = _rand((t_n, n, D), 1.0, 7, torch.float16)
    wk = _rand((2, D, D), 0.05, 8, torch.float16)
    wv = _rand((2, D, D), 0.03, 9, torch.float16)
    bv = _rand((2, D), 0.5, 10, torch.float32)
    # q of (branch b, token tok, head h) = 40 x the key vector of key j(b, tok, h): its score with that key dominates
    k_all = [(x[0].double() @ wk[b].double().T).view(n, H, 64) for b in (0, 1)]
    q = torch.zeros((2, t_n * l, D), dtype=torch.float16)
    pick = np.zeros((2, l, H), dtype=np.int64)
    for b in (0, 1):
        for tok in range(l):
            for h in range(H):
                j = (37 * tok + 11 * h + 5 * b) % n
                pick[b, tok, h] = j
                kv = k_all[b][j, h]
                q[b, tok, h * 64:(h + 1) * 64] = (kv * (300.0 / (kv @ kv))).half()          # q . k_j = 300 -> score 37.5 after the 1/8
    ref = _reference(q, x, wk, torch.zeros((2, D)), wv, bv, l)
    out = torch.empty((t_n, l, 2, D), dtype=torch.float16, device="cuda")
    ops.cross_attention_folded(q.cuda(), x.cuda(), ops.fold_pack_key(wk).cuda(), ops.fold_pack_value(wv).cuda(), bv.cuda(), out, l, 0.125)
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"\n[folded cross-attention, steered heads] max|err| vs fp64 {err:.2e}")
    assert err < 2e-2


def test_folded_cross_attention_rejects_other_geometries(ops):
    from candidate_reranking_cir_amd.lib import CirrankError
    q, x = torch.zeros((2, 2 * 40, D), dtype=torch.float16, device="cuda"), torch.zeros((2, 197, D), dtype=torch.float16, device="cuda")
    w, bv = torch.zeros((2, D, D), dtype=torch.float16, device="cuda"), torch.zeros((2, D), device="cuda")
    with pytest.raises(CirrankError):       # L > 32
        ops.cross_attention_folded(q, x, w, w, bv, torch.empty((2, 40, 2, D), dtype=torch.float16, device="cuda"), 40, 0.125)
    x2 = torch.zeros((2, 609, D), dtype=torch.float16, device="cuda")
    with pytest.raises(CirrankError):       # N > 608 keeps the projected path (577 tokens - the 384-px geometry - run on the 16-rows-per-wave kernel)
        ops.cross_attention_folded(q[:, :64], x2, w, w, bv, torch.empty((2, 32, 2, D), dtype=torch.float16, device="cuda"), 32, 0.125)


@pytest.mark.parametrize("t_n,l,n", [(4, 32, 197), (3, 13, 577), (2, 32, 224)])
def test_folded_cross_attention_with_a_key_mask(ops, t_n, l, n):
    """Round 6: an additive key mask per candidate (padded candidate token sets, nlvr_encoder.py:863-868: (1 - m) * finfo.min) in both folded
    kernels - against fp64 with the same mask, an all-zero mask = the unmasked kernel bit for bit, and a fully masked candidate stays finite
    (uniform attention, like the reference's softmax over equal finfo.min logits)."""
    dtype = torch.float16
    q = _rand((2, t_n * l, D), 1.0, 11, dtype)
    x = _rand((t_n, n, D), 1.0, 12, dtype)
    wk, wv = _rand((2, D, D), 0.03, 13, dtype), _rand((2, D, D), 0.03, 14, dtype)
    bk, bv = _rand((2, D), 0.5, 15, torch.float32), _rand((2, D), 0.5, 16, torch.float32)
    keep = torch.rand((t_n, n), generator=torch.Generator().manual_seed(17)) > 0.3
    keep[:, 0] = True
    keep[0, n // 2:] = False                                             # a candidate padded to half its tokens
    mask = ((1.0 - keep.float()) * torch.finfo(torch.float32).min)
    args = (q.cuda(), x.cuda(), ops.fold_pack_key(wk).cuda(), ops.fold_pack_value(wv).cuda(), bv.cuda())
    out = torch.empty((t_n, l, 2, D), dtype=dtype, device="cuda")
    ops.cross_attention_folded(*args, out, l, 0.125, mask=mask.cuda())
    # fp64 reference with the mask
    ref = torch.empty((t_n, l, 2, D), dtype=torch.float64)
    for b in (0, 1):
        k = (x.double() @ wk[b].double().T + bk[b].double()).view(t_n, n, H, 64).permute(0, 2, 1, 3)
        v = (x.double() @ wv[b].double().T + bv[b].double()).view(t_n, n, H, 64).permute(0, 2, 1, 3)
        qq = q[b].double().view(t_n, l, H, 64).permute(0, 2, 1, 3)
        s = qq @ k.transpose(-1, -2) / 8.0 + mask.double().clamp(min=-1e30)[:, None, None, :]
        ref[:, :, b] = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(t_n, l, D)
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"\n[folded cross-attention, key mask, T {t_n} L {l} N {n}] max|err| vs fp64 {err:.2e}")
    assert err < 6e-3
    plain = torch.empty_like(out)
    zero = torch.empty_like(out)
    ops.cross_attention_folded(*args, plain, l, 0.125)
    ops.cross_attention_folded(*args, zero, l, 0.125, mask=torch.zeros((t_n, n), device="cuda"))
    assert (plain.float() - zero.float()).abs().max().item() < 2e-3      # (the masked form scales the scores before the maximum: one rounding apart)
    allm = torch.empty_like(out)
    ops.cross_attention_folded(*args, allm, l, 0.125, mask=torch.full((t_n, n), torch.finfo(torch.float32).min, device="cuda"))
    assert bool(torch.isfinite(allm.float()).all())


def test_engine_takes_the_fold_with_a_candidate_mask_and_counts_long_captions(ops):
    """NlvrEngine: a padded candidate set (cand_mask) keeps the query-side fold (round 6) and agrees with the projected path; captions of more
    than 32 tokens take the projected path - visibly (`fold_fallbacks`, one warning)."""
    import warnings
    from candidate_reranking_cir_amd import synthetic
    from candidate_reranking_cir_amd.config import BertGeometry, VitGeometry
    from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR
    dev = torch.device("cuda")
    vit = VitGeometry(image_size=64, patch_size=16, width=768, depth=1, num_heads=12)
    torch.manual_seed(0)
    m = BLIP_NLVR(BertGeometry(num_hidden_layers=3), vit_geometry=vit, tokenizer=synthetic.HashTokenizer()).to(dev).eval()
    eng = m.engines()[1]
    g = torch.Generator(device="cpu").manual_seed(5)
    q_n, k, l, n = 2, 6, 12, vit.num_tokens
    z = torch.randn((q_n, l, 768), generator=g).to(dev)
    ids = torch.randint(1000, 20000, (q_n, l), generator=g).to(dev)
    mask = torch.ones_like(ids)
    cand = (torch.randn((q_n * k, n, 768), generator=g) * 0.5).to(dev).half()
    qidx = torch.arange(q_n, device=dev).repeat_interleave(k)
    cmask = torch.ones((q_n * k, n), dtype=torch.int64, device=dev)
    cmask[::2, n - 4:] = 0
    folded = eng.forward(ids, mask, z, cand, qidx, cand_mask=cmask)
    eng.fold_cross_kv = False
    projected = eng.forward(ids, mask, z, cand, qidx, cand_mask=cmask)
    eng.fold_cross_kv = True
    unmasked = eng.forward(ids, mask, z, cand, qidx)
    assert (folded - projected).abs().max().item() < 3e-3 and (folded - unmasked).abs().max().item() > 1e-4
    assert eng.fold_fallbacks == 0
    l2 = 40
    z2 = torch.randn((q_n, l2, 768), generator=g).to(dev)
    ids2 = torch.randint(1000, 20000, (q_n, l2), generator=g).to(dev)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng.forward(ids2, torch.ones_like(ids2), z2, cand, qidx)
        eng.forward(ids2, torch.ones_like(ids2), z2, cand, qidx)
    assert eng.fold_fallbacks == 2 and sum("projected" in str(x.message) for x in w) == 1
