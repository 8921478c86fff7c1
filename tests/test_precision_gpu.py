"""Rank identity and the residual-stream precision choice, pinned on reference outputs (MI355X).

north_star asks for "identical top-K rank order".  A 16-bit path cannot promise that for candidates whose fp32 logits
differ by less than its own rounding drift, so the claim is split into what IS asserted:
  * every pair / sorted position that the reference decides by a margin holds (tests/test_model_gpu.py);
  * on top of that (tests/test_model_gpu.py::test_rank_identity_floors, next to the shared rank224 fixture): the fraction of
    sorted positions holding exactly the reference's candidate, Kendall's tau of the two orders and the top-10 overlap have
    FLOORS (measured on MI355X, minus 10 %), per operand dtype;
  * HERE: the precision table: operand modes (bf16 / mixed / fp16) x residual-stream storage against a fixture whose weights
    carry OUTLIER CHANNELS (tests/golden/outlier224.npz: the reference's ViT residual stream peaks at ~1e2..1e3 in three
    channels, as pretrained checkpoints do).  Round 3 ran bf16 operands by default and lost the reference's order there (tau
    0.62); since round 4 the library default is fp16 operands + fp16 streams, and its floors here are tau >= 0.90, top-10 >= 0.9.
"""
import numpy as np
import pytest
import torch
from scipy.stats import kendalltau

from candidate_reranking_cir_amd import synthetic
from tests import helpers as H
from tests.test_model_gpu import build_models

pytestmark = pytest.mark.gpu
BF, HF, F32 = torch.bfloat16, torch.float16, torch.float32


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def order_stats(ours: np.ndarray, ref: np.ndarray):
    """(exact-position fraction, Kendall tau, top-10 overlap fraction) of two logit rows."""
    o, r = np.argsort(-ours, kind="stable"), np.argsort(-ref, kind="stable")
    tau = kendalltau(ours, ref).statistic
    return float((o == r).mean()), float(tau), len(set(o[:10]) & set(r[:10])) / 10.0


# ------------------------------------------------------------------------------------------------ outlier-channel weights
# mode -> (text operands, image operands or None, text stream, ViT stream), bound on max|dlogit| (the fixture's logit sigma is only
# 0.027: the outlier channels dominate every LayerNorm's statistics and compress the informative signal), floor on Kendall tau,
# floor on the top-10 overlap.  Measured on MI355X (round 4, tools/precision_modes.py -> profiles/r4_precision_modes.json):
#   bf16, fp16 streams (round 1-3 headline)            4.2e-2 / tau 0.647 / top-10 0.50     16.8 k triplets/s
#   mixed (ViT + cross block bf16, text fp16), fp16     8.9e-3 / tau 0.909 / top-10 0.90     16.9 k
#   fp16, fp16 streams  (LIBRARY DEFAULT)               1.1e-2 / tau 0.906 / top-10 0.85     16.6 k
#   fp16, text stream fp32, ViT stream fp16 ("split")   3.5e-3 / tau 0.942 / top-10 0.95     15.9 k
#   fp16, fp32 streams                                  3.4e-3 / tau 0.942 / top-10 0.95     15.0 k
# The per-site attribution (oracle/attribute_rounding.py, profiles/r4_precision_attribution_outlier.json) says where the bf16
# error enters on THIS fixture: the text-side self-attention / FFN / cls_head activations (ViT and cross block in bf16: tau
# unchanged) - which is what "mixed" keeps in fp16; on well-conditioned weights (rank224) the bf16 WEIGHT rounding of the ViT
# and of the cross K|V projection costs exact positions too (0.93 -> 0.82), so the default is fp16 everywhere.
# The default's tau floor is the acceptance bar of the round-3 review (tau >= 0.90; measured 0.906 - 0.910 across the round's GEMM
# epilogue variants).  Its top-10 overlap is 0.85 - 0.90 in EVERY fp16 mode, strictest included (two scored queries: one candidate
# = 0.05; the reference's 10th / 11th candidates are closer than any 16-bit path can resolve), so that floor is 0.85.
OUTLIER_MODES = {
    "bf16 | fp16 streams": ((BF, None, HF, HF), (5.5e-2, 0.52, 0.3)),
    "mixed | fp16 streams": ((HF, BF, HF, HF), (2.0e-2, 0.88, 0.8)),
    "fp16 | fp16 streams (default)": ((HF, None, HF, HF), (1.5e-2, 0.90, 0.85)),
    "fp16 | text fp32, ViT fp16": ((HF, None, F32, HF), (9.0e-3, 0.92, 0.8)),
    "fp16 | fp32 streams": ((HF, None, F32, F32), (8.0e-3, 0.90, 0.8)),
}


def test_outlier_weights_precision_table(cuda):
    from candidate_reranking_cir_amd import validate_stage2 as V
    z = H.load("outlier224.npz")
    g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
    ref, gref = z["logits"], z["group_logits"]
    skipped = ~z["labels"].any(1)
    ref_out = z["bank_outlier_slice"]
    imgs = synthetic.scene_images(range(int(z["n_index"])), 224)
    ds = V.RelativeValSet(ref_index=z["refs"], cand_index=z["cand"], labels=z["labels"], captions=[str(c) for c in z["caps"]],
                          group_index=z["groups"], target_index=z["targets"])
    errs, taus = {}, {}
    print(f"\n[outlier224] reference residual-stream peaks: ViT {float(z['vit_stream_peak']):.0f}, BERT {float(z['bert_stream_peak']):.0f}; "
          f"logit sigma {ref[~skipped].std():.3f}")
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), HF, cuda)
    assert m2.set_precision("f16").set_stream_dtype(None, vit=None).stream_dtype == HF and m2.compute_dtype == HF   # what "default" means
    for mode, ((dt, idt, sdt, vsdt), (tol, tau_min, top_min)) in OUTLIER_MODES.items():
        for m in (m2, m1):
            m.set_compute_dtype(dt, idt).set_stream_dtype(sdt, vit=vsdt)
        bank = V.extract_index_features(imgs, m2, batch_size=64, dtype=torch.float32)
        assert torch.isfinite(bank).all()
        e_tok = np.abs(bank[:, :3, :8].cpu().numpy() - z["bank_slice"]).max()
        e_out = np.abs(bank[:8, :3][:, :, [17, 300, 555]].cpu().numpy() - ref_out).max() / np.abs(ref_out).max()
        lt, gt = V.generate_cirr_val_predictions(m2, m1, ds, bank.to(m2.token_dtype), query_batch=3)
        logits, gl = lt.cpu().numpy(), gt.cpu().numpy()
        assert np.array_equal(logits[skipped], ref[skipped]) and np.isfinite(logits).all() and np.isfinite(gl).all()
        err = max(np.abs(logits[~skipped] - ref[~skipped]).max(), np.abs(gl - gref).max())
        stats = np.array([order_stats(logits[q], ref[q]) for q in np.where(~skipped)[0]])
        exact, tau, top10 = stats.mean(0)
        errs[mode], taus[mode] = err, tau
        print(f"   {mode:32s} tokens {e_tok:.3e} (outlier channels rel {e_out:.2e})  max|dlogit| {err:.3e}  exact positions {exact:.2f}  "
              f"tau {tau:.3f}  top-10 overlap {top10:.2f}")
        assert err < tol and tau >= tau_min and top10 >= top_min - 1e-9, mode
        assert e_out < (2e-2 if m2.token_dtype == BF else 3e-3)
        del bank
    # the operand format decides (7x), the storage of the residual stream refines
    assert errs["fp16 | fp16 streams (default)"] < 0.5 * errs["bf16 | fp16 streams"]
    assert taus["fp16 | fp16 streams (default)"] > taus["bf16 | fp16 streams"] + 0.2
    assert errs["fp16 | fp32 streams"] < errs["fp16 | fp16 streams (default)"]


# ------------------------------------------------------------------------------------------------ ViT-large against the reference
@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
def test_vit_large_reference_golden(cuda, dtype):
    """`vit='large'` (blip.py:203-209: depth 24, width 1024, 16 heads) against the reference's own VisionTransformer
    (tests/golden/vitl_tiny.npz, 64 px): final tokens and the CLS row after every one of the 24 blocks."""
    from candidate_reranking_cir_amd import config, weights
    from candidate_reranking_cir_amd.engine import VitEngine
    z = H.load("vitl_tiny.npz")
    v = config.VitGeometry(image_size=64, width=1024, depth=24, num_heads=16)
    sd = weights.synth_state_dict(weights._vit_spec(v), int(z["seed"]), str(z["profile"]))
    eng = VitEngine({k: t.cuda() for k, t in sd.items()}, v, dtype, cuda, stream_dtype=torch.float16 if dtype == torch.bfloat16 else torch.float32)
    y32, y16 = eng.forward(synthetic.images(z["image_ids"].tolist(), 64).cuda(), want32=True)
    assert y32.shape == (4, 17, 1024)
    err = np.abs(y32[:, :, :16].cpu().numpy() - z["tokens_slice"]).max()
    rel_sum = abs(y32.double().sum().item() - float(z["tokens_sum"])) / (float(z["tokens_abs_mean"]) * y32.numel())
    print(f"\n[vit-large golden {dtype}] tokens max|err| {err:.3e}  |sum error| / (n * mean|x|) {rel_sum:.2e}")
    assert err < (5e-2 if dtype == BF else 6e-3) and rel_sum < 1e-4        # measured 2.2e-2 / 2.6e-3 after 24 blocks
