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
