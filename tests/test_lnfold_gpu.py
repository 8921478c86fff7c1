"""cir_gemm_ln_bias_act on a real MI355X: LayerNorm folded into the GEMM behind it (vit.py:107-109 + :72 / :36-37) against the fp64
composition LayerNorm -> Linear (-> GELU), against the LayerNorm pass + GEMM it replaces, bounds, and independence of the row count."""
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


def _case(m, n, k, seed, outliers=True):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((m, k), generator=g) * 1.5 + 0.3
    if outliers:                                   # ViT residual streams carry a few channels far from the rest
        x[:, 5] += 40.0
        x[:, k - 3] -= 25.0
    w = torch.randn((n, k), generator=g) * 0.03
    b = torch.randn((n,), generator=g) * 0.1
    gamma = 1.0 + 0.2 * torch.randn((k,), generator=g)
    beta = 0.1 * torch.randn((k,), generator=g)
    return [t.cuda() for t in (x.half(), w, b, gamma, beta)]


def _ref64(x, w, b, gamma, beta, eps, gelu):
    y = F.layer_norm(x.double(), (x.shape[1],), gamma.double(), beta.double(), eps)
    o = y @ w.double().t() + b.double()
    return F.gelu(o) if gelu else o


@pytest.mark.parametrize("m,n,k,gelu", [(197, 2304, 768, False), (1000, 320, 768, True), (4099, 3072, 768, True), (2600, 64, 128, False),
                                        (70001, 784, 256, False)])
def test_folded_layernorm_gemm_matches_fp64_and_beats_the_two_pass_form(ops, m, n, k, gelu):
    x, w, b, gamma, beta = _case(m, n, k, seed=m + n)
    eps = 1e-6
    wg, cs, bb = ops.ln_fold_pack(w, b, gamma, beta)
    act = ops.ACT_GELU if gelu else ops.ACT_NONE
    got = ops.gemm_ln(x, wg, cs, bb, eps, act).double()
    ref = _ref64(x, w, b, gamma, beta, eps, gelu)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item() / scale
    assert err < 1.5e-3, err                                      # fp16 output: half an ulp of the largest value is 4.9e-4
    _, xb = ops.layernorm(x, gamma, beta, eps, want32=False, dtype16=torch.float16, stream_dtype=torch.float16)
    two = ops.gemm(xb, w.half(), b, act=act).double()
    rms_f, rms_t = (got - ref).pow(2).mean().sqrt().item(), (two - ref).pow(2).mean().sqrt().item()
    assert rms_f <= rms_t * 1.02, (rms_f, rms_t)                  # one rounding fewer (no fp16 copy of the normalised rows)


def test_row_statistics_are_exact_on_integer_rows(ops):
    """Small-integer rows: sums and sums of squares are exact in fp32, so mean / rstd equal the fp64 values to fp32 rounding and the
    result must agree with the formula evaluated in fp64 on the SAME packed operands to the fp16 output rounding alone."""
    m, n, k = 3000, 512, 768
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randint(-8, 9, (m, k), generator=g).half().cuda()
    w = (torch.randint(-4, 5, (n, k), generator=g).float() / 64.0).cuda()
    b = torch.randn((n,), generator=g).cuda()
    wg, cs, bb = ops.ln_fold_pack(w, b, torch.ones(k, device="cuda"), torch.zeros(k, device="cuda"))
    got = ops.gemm_ln(x, wg, cs, bb, 1e-6).double()
    x64 = x.double()
    mean, var = x64.mean(1, keepdim=True), x64.var(1, unbiased=False, keepdim=True)
    ref = (var + 1e-6).rsqrt() * (x64 @ wg.double().t() - mean * cs.double()) + bb.double()
    assert (got - ref).abs().max().item() <= ref.abs().max().item() * 2.0 ** -11 * 1.01


def test_result_does_not_depend_on_the_row_count(ops):
    x, w, b, gamma, beta = _case(5000, 2304, 768, seed=3)
    wg, cs, bb = ops.ln_fold_pack(w, b, gamma, beta)
    big = ops.gemm_ln(x, wg, cs, bb, 1e-6)
    for lo, hi in ((0, 197), (300, 497), (4803, 5000), (255, 258)):
        assert torch.equal(big[lo:hi], ops.gemm_ln(x[lo:hi], wg, cs, bb, 1e-6)), (lo, hi)
    assert torch.equal(big, ops.gemm_ln(x, wg, cs, bb, 1e-6))     # and not on the run


@pytest.mark.parametrize("m,n", [(79588, 2304), (66049, 2304 + 16), (300, 3072)])
def test_folded_gemm_guard_bands(ops, m, n):
    """> 256 tiles (every workgroup walks several: statistics registers are reset per tile, the exchange pages are re-used), a ragged
    last n-tile, strided rows in and out; canaries around the output stay intact."""
    from tests.test_guard_gpu import Guarded
    k = 768
    x, w, b, gamma, beta = _case(m, n, k, seed=11, outliers=False)
    xg = Guarded(m, k, torch.float16, pr=8, pc=64)
    xv = xg.fill_view(x)
    wg, cs, bb = ops.ln_fold_pack(w, b, gamma, beta)
    out = Guarded(m, n, torch.float16)
    ops.gemm_ln(xv, wg, cs, bb, 1e-6, ops.ACT_GELU, out=out.view)
    torch.cuda.synchronize()
    out.assert_intact("gemm_ln")
    rows = torch.cat([torch.arange(0, 600), torch.arange(m - 600, m)]).cuda() if m > 1200 else torch.arange(m).cuda()
    ref = _ref64(x[rows], w, b, gamma, beta, 1e-6, True)
    assert (out.view[rows].double() - ref).abs().max().item() < 1.5e-3 * ref.abs().max().item()


def test_vit_engine_with_and_without_the_fold(ops):
    """The ViT engine with norm1 (and norm2) folded against the same engine with LayerNorm passes: both are fp16 roundings of one
    function - the token difference stays at the fp16 noise of a 2-block encoder, and the folded engine is at least as close to fp64."""
    from candidate_reranking_cir_amd.config import VitGeometry
    from candidate_reranking_cir_amd.engine import VitEngine
    geo = VitGeometry(image_size=64, patch_size=16, width=256, depth=2, num_heads=4, mlp_ratio=4)
    g = torch.Generator(device="cpu").manual_seed(5)
    d, n_tok = geo.width, geo.num_tokens
    sd = {"visual_encoder.patch_embed.proj.weight": torch.randn((d, 3, 16, 16), generator=g) * 0.02, "visual_encoder.patch_embed.proj.bias": torch.zeros(d),
          "visual_encoder.cls_token": torch.randn((1, 1, d), generator=g) * 0.02, "visual_encoder.pos_embed": torch.randn((1, n_tok, d), generator=g) * 0.02,
          "visual_encoder.norm.weight": torch.ones(d), "visual_encoder.norm.bias": torch.zeros(d)}
    for i in range(geo.depth):
        p = f"visual_encoder.blocks.{i}."
        sd.update({p + "norm1.weight": 1 + 0.1 * torch.randn(d, generator=g), p + "norm1.bias": 0.1 * torch.randn(d, generator=g),
                   p + "norm2.weight": 1 + 0.1 * torch.randn(d, generator=g), p + "norm2.bias": 0.1 * torch.randn(d, generator=g),
                   p + "attn.qkv.weight": torch.randn((3 * d, d), generator=g) * 0.05, p + "attn.qkv.bias": torch.randn(3 * d, generator=g) * 0.02,
                   p + "attn.proj.weight": torch.randn((d, d), generator=g) * 0.05, p + "attn.proj.bias": torch.zeros(d),
                   p + "mlp.fc1.weight": torch.randn((4 * d, d), generator=g) * 0.05, p + "mlp.fc1.bias": torch.zeros(4 * d),
                   p + "mlp.fc2.weight": torch.randn((d, 4 * d), generator=g) * 0.05, p + "mlp.fc2.bias": torch.zeros(d)})
    img = torch.randn((6, 3, 64, 64), generator=g).cuda()
    eng = VitEngine(sd, geo, torch.float16, torch.device("cuda"))
    assert "qkv_f" in eng.blocks[0] and eng.ln_fold == 1
    exact = VitEngine(sd, geo, torch.float32, torch.device("cuda"), stream_dtype=torch.float32)
    ref = exact.forward(img, want32=True)[0].double()
    outs = {}
    for mode in (0, 1, 2):
        eng.ln_fold = mode
        outs[mode] = eng.forward(img, want32=True)[0].double()
    err = {mode: (o - ref).abs().max().item() for mode, o in outs.items()}
    assert err[1] <= err[0] * 1.5 + 1e-4 and err[2] <= err[0] * 1.5 + 1e-4, err
    assert (outs[1] - outs[0]).abs().max().item() < 3e-2 and (outs[2] - outs[0]).abs().max().item() < 3e-2


@pytest.mark.parametrize("mean,std,bound", [(3.0, 1.0, 1.0e-3), (50.0, 0.5, 4e-3)])
def test_folded_layernorm_on_rows_with_a_large_common_offset(ops, mean, std, bound):
    """Round-5 advisor finding: the folded form takes the row variance in ONE pass (E[x^2] - mean^2 over fp32 sums of the fp16 row), where the
    LayerNorm pass subtracts the mean first.  Rows whose COMMON offset dwarfs their spread cancel: at |mean| / std = 100 the fp32 sum of
    squares (1.9e6 per 768-wide row, ulp 0.125) carries the variance (0.25) less exactly - measured 1.5e-3 of the output's scale against 4.6e-4
    for the LayerNorm pass + GEMM (bounded at 4e-3 here); for |mean| / std = 3 both forms sit at the fp16 output rounding (4.3e-4).  The ViT residual streams this kernel serves have row means within
    a few sigma (their outlier CHANNELS raise the row's spread, not its mean: reference stream peak 344, row mean O(1)); a checkpoint whose rows
    sit on a common offset of hundreds of sigma should run with `VitEngine.ln_fold = 0` (the LayerNorm pass)."""
    m, n, k = 3000, 768, 768
    g = torch.Generator(device="cpu").manual_seed(int(mean))
    x = (torch.randn((m, k), generator=g) * std + mean).half().cuda()
    w = (torch.randn((n, k), generator=g) * 0.03).cuda()
    b = (torch.randn((n,), generator=g) * 0.1).cuda()
    gamma = (1.0 + 0.2 * torch.randn((k,), generator=g)).cuda()
    beta = (0.1 * torch.randn((k,), generator=g)).cuda()
    wg, cs, bb = ops.ln_fold_pack(w, b, gamma, beta)
    got = ops.gemm_ln(x, wg, cs, bb, 1e-6).double()
    ref = _ref64(x, w, b, gamma, beta, 1e-6, False)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    _, xb = ops.layernorm(x, gamma, beta, 1e-6, want32=False, dtype16=torch.float16, stream_dtype=torch.float16)
    two = (ops.gemm(xb, w.half(), b).double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"\n[LayerNorm fold, rows N({mean}, {std})] relative max error: folded {err:.2e}, LayerNorm pass + GEMM {two:.2e}")
    assert err < bound
