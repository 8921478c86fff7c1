"""End-to-end parity of the HIP path (through the C ABI) against the golden vectors of the real
reference and against the CPU oracle, on a real MI355X.

Tolerances (stated per north_star): the reference computes in fp32; the HIP path rounds GEMM and
attention operands to bf16 (8-bit mantissa, ~0.4 % per rounding) or fp16 (11-bit) and accumulates
in fp32.  Measured drift on these fixtures is printed by each test (`-s`); every asserted bound is
about 2x the drift measured on MI355X (table in DESIGN.md section 2):
  ViT tokens / z_t (O(1) LayerNorm outputs): TOK_TOL, absolute
  logits: LOGIT_TOL per fixture (the fixtures' logit spreads differ by 100x), absolute; tiny fixtures: a fraction
          of the row's own spread (REL_TOL * sigma)
  rank order: every candidate pair whose reference gap exceeds 4x the bound keeps its order, the NUMBER of such
          decided pairs is asserted (>= half of all pairs on the rank fixtures), sorted positions whose two
          neighbour gaps exceed the margin hold the same candidate, and Recall@k / Recall_subset@k computed from
          our logits equal the reference's recall tuples on labels whose rank is decided by that margin.
"""
import json

import numpy as np
import pytest
import torch

from candidate_reranking_cir_amd import synthetic
from tests import helpers as H

pytestmark = pytest.mark.gpu

BF, HF = torch.bfloat16, torch.float16
# Precision modes under test.  BF = bf16 operands + fp16 residual streams (BASELINE's named dtype; round 1-3 headline mode);
# HF = fp16 operands + fp32 residual streams (the strictest mode: what the fp16 tolerances below were measured on);
# DEF = the LIBRARY DEFAULT since round 4: fp16 operands + fp16 residual streams (the mode that holds the
# reference's rank order at >= 16 k triplets/s) - checked on the rank fixtures and the outlier-weight table.
DEF = "fp16-default"
TOK_TOL = {BF: 4e-2, HF: 6e-3, DEF: 2.5e-2}
LOGIT_TOL = {                       # absolute, per fixture (logit sigma over candidates in brackets)
    "full224": {BF: 6e-3, HF: 1.5e-3},          # [0.026]
    "full224_spread": {BF: 4e-2, HF: 4.5e-3},   # [0.345]  (bf16: 1.4e-2 .. 2.7e-2 across the epilogue variants of round 2)
    "full384": {BF: 8e-3, HF: 2e-3},
    "rank224": {BF: 8e-3, HF: 2e-3, DEF: 3e-3},  # [0.14]  (BF = tol_unit the fixture's label margins were cut with; DEF measured 1.5e-3)
    "bxb224": {BF: 6e-3, HF: 1.5e-3},           # [0.135]
}
REL_TOL = {BF: 0.10, HF: 0.015}     # tiny geometry: |error| <= REL_TOL * sigma(reference logits of that row) [sigma 0.08..0.2; measured 0.039 / 0.006]


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def build_models(g, v, seed, profile, dtype, device, fold_merge=True):
    from candidate_reranking_cir_amd.blip_stage1 import BLIP_Retrieval
    from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR
    sd2, sd1 = H.state_dicts(g, v, seed, profile)
    m2 = BLIP_NLVR(med_config=g, vit_geometry=v, fold_merge=fold_merge, tokenizer=synthetic.HashTokenizer())
    m1 = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    assert m2.load_state_dict(sd2, strict=True) is not None and m1.load_state_dict(sd1, strict=True) is not None
    m2, m1 = m2.to(device).float().eval(), m1.to(device).float().eval()
    for m in (m2, m1):
        if dtype == DEF:
            m.set_precision("f16").set_stream_dtype(None, vit=None)          # the library's defaults, spelled out
        else:
            m.set_compute_dtype(dtype).set_stream_dtype(torch.float16 if dtype == BF else torch.float32)
    return m2, m1


def pair_order(ours: np.ndarray, ref: np.ndarray, tol: float):
    """(decided, total, ok): candidate pairs whose reference gap exceeds 4*tol, all pairs, and whether every decided
    pair keeps its order in `ours`."""
    iu = np.triu_indices(len(ref), 1)
    d_ref = (ref[:, None] - ref[None, :])[iu]
    d_our = (ours[:, None] - ours[None, :])[iu]
    decided = np.abs(d_ref) > 4 * tol
    return int(decided.sum()), len(d_ref), bool(np.all(np.sign(d_ref[decided]) == np.sign(d_our[decided])))


def margin_order_ok(ours: np.ndarray, ref: np.ndarray, tol: float) -> bool:
    return pair_order(ours, ref, tol)[2]


def fixed_positions(ours: np.ndarray, ref: np.ndarray, tol: float):
    """Sorted positions whose gaps to BOTH neighbours exceed 4*tol in the reference: (count, all hold the same candidate)."""
    o_ref, o_our = np.argsort(-ref, kind="stable"), np.argsort(-ours, kind="stable")
    s = ref[o_ref]
    up = np.concatenate([[np.inf], s[:-1] - s[1:]])
    dn = np.concatenate([s[:-1] - s[1:], [np.inf]])
    pos = np.where((up > 4 * tol) & (dn > 4 * tol))[0]
    return len(pos), bool(np.all(o_ref[pos] == o_our[pos]))


# ------------------------------------------------------------------------------------------------ tiny geometry, full loop
@pytest.fixture(scope="module", params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def tiny(request, cuda):
    z = H.load("tiny_loop.npz")
    g, v = H.geometry(json.loads(str(z["bert_cfg"])), json.loads(str(z["vit_cfg"])))
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), request.param, cuda)
    return z, g, v, m2, m1, request.param


def test_img_embed_tiny(tiny):
    z, g, v, m2, m1, dt = tiny
    feats = m2.img_embed(H.fixture_images(z, range(14), v.image_size).cuda())
    assert feats.dtype == torch.float32 and feats.shape == (14, v.num_tokens, v.width)
    err = np.abs(feats[:, :3, :8].cpu().numpy() - z["index_features_slice"]).max()
    print(f"\n[tiny vit {dt}] max|err| = {err:.3e}")
    assert err < TOK_TOL[dt]
    f2, atts = m2.img_embed(H.fixture_images(z, range(2), v.image_size).cuda(), atts=True)
    assert atts.dtype == torch.long and atts.shape == f2.shape[:2] and bool((atts == 1).all())


@pytest.mark.parametrize("flavour", ["cirr", "fiq"])
@pytest.mark.parametrize("query_batch", [1, 3, 8])
def test_scoring_loop_tiny(tiny, flavour, query_batch):
    from candidate_reranking_cir_amd import validate_stage2 as V
    z, g, v, m2, m1, dt = tiny
    bank = V.extract_index_features(H.fixture_images(z, range(14), v.image_size), m2)
    caps = [str(c) for c in z["cirr_caps"]] if flavour == "cirr" else [V.fiq_caption(str(p[0]), str(p[1])) for p in z["fiq_caps"]]
    ds = V.RelativeValSet(ref_index=z["refs"], cand_index=z["cand_idx"], labels=z["labels"], captions=caps,
                          group_index=z["groups"] if flavour == "cirr" else None, target_index=z["targets"])
    out = V.generate_val_predictions(m2, m1, ds, bank, query_batch=query_batch)
    logits = (out[0] if flavour == "cirr" else out).cpu().numpy()
    ref = z[f"{flavour}_logits"]
    skipped = ~z["labels"].any(1)
    assert skipped.any() and np.all(logits[skipped] == np.float32(-99999.99)) and np.all(ref[skipped] == np.float32(-99999.99))
    # the tiny model's logits cluster (sigma ~ 2e-3 per row): the bound is a fraction of each row's own spread, so a model
    # that ignored the candidates (row mean everywhere, error ~ 1 sigma) cannot pass
    sig = ref[~skipped].std(axis=1, keepdims=True)
    rel = (np.abs(logits[~skipped] - ref[~skipped]) / sig).max()
    print(f"\n[tiny {flavour} {dt} qb={query_batch}] max|dlogit| = {np.abs(logits[~skipped] - ref[~skipped]).max():.3e} "
          f"= {rel:.3f} sigma (row sigma {sig.min():.2e}..{sig.max():.2e})")
    assert rel < REL_TOL[dt]
    if flavour == "cirr":
        gref = z["cirr_group_logits"]
        grel = (np.abs(out[1].cpu().numpy() - gref) / gref.std(axis=1, keepdims=True)).max()
        assert grel < 2 * REL_TOL[dt]                      # 5 members: the row sigma itself is a noisy estimate
        for q in np.where(~skipped)[0]:
            assert margin_order_ok(logits[q], ref[q], REL_TOL[dt] * float(sig.min()))
        metrics = V.compute_cirr_val_metrics(out[0], out[1], ds)
        ref_metrics = V.compute_cirr_val_metrics(torch.tensor(ref), torch.tensor(z["cirr_group_logits"]), ds)
        np.testing.assert_allclose(ref_metrics, z["cirr_metrics"], atol=1e-4)     # host metric code == reference's
        print("   recall ours", np.round(metrics, 2), "reference", np.round(z["cirr_metrics"], 2))
    else:
        ref_metrics = V.compute_fiq_val_metrics(torch.tensor(ref), ds)
        np.testing.assert_allclose(ref_metrics, z["fiq_metrics"], atol=1e-4)


def test_reference_signatures(tiny):
    """The reference's own call forms (stage2_train.py:270, :513; validate_stage2.py:69-71, 153-156, 209-211): duck-typed
    FashionIQ / CIRR datasets with NAMES, fp32 index features as extract_index_features returns them -> the reference's
    tuples, equal to the native form bit for bit and to the reference's outputs (tiny_loop.npz) within the bound."""
    from candidate_reranking_cir_amd import validate_stage2 as V
    z, g, v, m2, m1, dt = tiny
    names = ["n%03d" % (7 * i % 100) for i in range(14)]
    feats32 = m2.img_embed(H.fixture_images(z, range(14), v.image_size).cuda())          # (14, N, D) fp32, utils.py:51
    fiq = H.DuckFIQ(names, z["refs"], z["targets"], z["fiq_caps"], z["cand_idx"], z["labels"])
    cirr = H.DuckCIRR(names, z["refs"], z["targets"], z["cirr_caps"], z["cand_idx"], z["labels"], z["groups"], ref_slot=2)
    skipped = ~z["labels"].any(1)
    lg, tn = V.generate_fiq_val_predictions(m2, m1, fiq, names, feats32)
    assert tn == [names[i] for i in z["targets"]] and lg.shape == (8, 6) and lg.dtype == torch.float32
    clg, cgl, rn, ctn, mem = V.generate_cirr_val_predictions(m2, m1, cirr, names, feats32)
    assert rn == [names[i] for i in z["refs"]] and mem == [[names[j] for j in row] for row in z["groups"]]
    bank = V.extract_index_features(H.fixture_images(z, range(14), v.image_size), m2)
    caps = [V.fiq_caption(str(p[0]), str(p[1])) for p in z["fiq_caps"]]
    nat = V.generate_val_predictions(m2, m1, V.RelativeValSet(ref_index=z["refs"], cand_index=z["cand_idx"], labels=z["labels"], captions=caps), bank)
    assert torch.equal(nat, lg)                                                           # same path underneath
    for ours, ref in ((lg.cpu().numpy(), z["fiq_logits"]), (clg.cpu().numpy(), z["cirr_logits"])):
        assert np.all(ours[skipped] == np.float32(-99999.99))
        sig = ref[~skipped].std(axis=1, keepdims=True)
        assert (np.abs(ours[~skipped] - ref[~skipped]) / sig).max() < REL_TOL[dt]
    gref = z["cirr_group_logits"]
    assert (np.abs(cgl.cpu().numpy() - gref) / gref.std(axis=1, keepdims=True)).max() < 2 * REL_TOL[dt]
    m_f = V.compute_fiq_val_metrics(fiq, m2, m1, feats32, names)
    m_c = V.compute_cirr_val_metrics(cirr, m2, m1, feats32, names)
    assert m_f == V.compute_fiq_val_metrics(lg, V.relative_val_set_from_dataset(fiq, names)[0])
    assert len(m_f) == 2 and len(m_c) == 7 and all(0.0 <= x <= 100.0 for x in m_f + m_c)
    print(f"\n[reference signatures {dt}] fiq {np.round(m_f, 2)} (reference {np.round(z['fiq_metrics'], 2)})  cirr {np.round(m_c, 2)} "
          f"(reference {np.round(z['cirr_metrics'], 2)})")


def test_padded_masks(tiny):
    """Padded captions (attention_mask zeros) through both text encoders, batch of 3 ragged rows."""
    z, g, v, m2, m1, dt = tiny
    m = H.load("masks.npz")
    ids, mask = torch.tensor(m["input_ids"]).cuda(), torch.tensor(m["attention_mask"]).cuda()
    feats16 = m2.img_embed16(H.fixture_images(z, range(6), v.image_size).cuda())
    zt = m1.z_t(feats16[:3], ids, mask)
    valid = mask.bool().cpu().numpy()
    err1 = np.abs(zt.last_hidden_state.cpu().numpy() - m["stage1_hidden"])[valid].max()
    # stage II on the reference's own z_t so that the two encoders are checked independently
    eng = m2.engines()[1]
    taps = []
    eng.forward(ids, mask, torch.tensor(m["stage1_hidden"]).cuda(), feats16[3:6], torch.arange(3).cuda(), taps=taps)
    ours = torch.cat([taps[-1][0], taps[-1][1]], dim=0).cpu().numpy()
    d = g.hidden_size
    ref = np.concatenate([m["stage2_hidden"][:, :8], m["stage2_hidden"][:, d:d + 8]], axis=0)
    err2 = np.abs(ours - ref).max()
    print(f"\n[masks {dt}] stage-I max|err| = {err1:.3e}, stage-II CLS max|err| = {err2:.3e}")
    assert err1 < TOK_TOL[dt] and err2 < TOK_TOL[dt]


def test_batch_invariance_and_api(tiny):
    """A candidate scored alone equals the same candidate scored inside a batch (the reference's
    expand-to-K semantics), through the drop-in img_txt_fusion_val(text=[str]) surface."""
    z, g, v, m2, m1, dt = tiny
    feats = m2.img_embed(H.fixture_images(z, range(8), v.image_size).cuda())
    cap = [str(z["cirr_caps"][2])]
    zt = m1.img_txt_fusion(feats[:1], None, cap, train=False, return_raw=True)
    assert zt.last_hidden_state.shape[0] == 1
    full = m2.img_txt_fusion_val(zt, feats[1:], cap)
    assert full.shape == (7,) and full.dtype == torch.float32
    one = m2.img_txt_fusion_val(zt, feats[3:4], cap)
    assert torch.equal(one, full[2:3])
    bb = m2.img_txt_fusion(type(zt)(zt.last_hidden_state.expand(2, -1, -1).contiguous()), feats[1:3], cap * 2)
    assert bb.shape == (2, 2) and torch.allclose(bb[0], full[:2], atol=1e-6) and torch.allclose(bb[1], full[:2], atol=1e-6)


def test_kv_bank_reuse_is_bit_identical(tiny):
    """Cross-query reuse: scoring out of the per-image K/V bank gives exactly the logits of the
    per-candidate projection (same kernel, same rows), including skip rows and the CIRR subset."""
    from candidate_reranking_cir_amd import validate_stage2 as V
    z, g, v, m2, m1, dt = tiny
    bank = V.extract_index_features(H.fixture_images(z, range(14), v.image_size), m2)
    ds = V.RelativeValSet(ref_index=z["refs"], cand_index=z["cand_idx"], labels=z["labels"], captions=[str(c) for c in z["cirr_caps"]],
                          group_index=z["groups"], target_index=z["targets"])
    plain = V.generate_val_predictions(m2, m1, ds, bank, query_batch=3)
    kvb = m2.build_kv_bank(bank)
    assert len(kvb) == g.num_hidden_layers and kvb[0].shape == (14, v.num_tokens, 4 * g.hidden_size)
    reuse = V.generate_val_predictions(m2, m1, ds, bank, query_batch=3, kv_bank=kvb)
    assert torch.equal(plain[0], reuse[0]) and torch.equal(plain[1], reuse[1])
    # ... and the bank path itself against the REFERENCE's logits (not only against the other HIP path)
    ref, gref, skipped = z["cirr_logits"], z["cirr_group_logits"], ~z["labels"].any(1)
    ours, gours = reuse[0].cpu().numpy(), reuse[1].cpu().numpy()
    assert np.all(ours[skipped] == np.float32(-99999.99))
    assert (np.abs(ours[~skipped] - ref[~skipped]) / ref[~skipped].std(axis=1, keepdims=True)).max() < REL_TOL[dt]
    assert (np.abs(gours - gref) / gref.std(axis=1, keepdims=True)).max() < 2 * REL_TOL[dt]


def test_last_layer_cls_trimming_is_equivalent(tiny):
    """The last layer's per-token work on CLS rows only gives the logits of the untrimmed schedule (same rows, same
    kernels up to the tile variant; bias enters the accumulator first or last)."""
    z, g, v, m2, m1, dt = tiny
    feats = m2.img_embed(H.fixture_images(z, range(9), v.image_size).cuda())
    cap = [str(z["cirr_caps"][3])]
    zt = m1.img_txt_fusion(feats[:1], None, cap, train=False, return_raw=True)
    eng = m2.engines()[1]
    assert eng.trim_last
    eng.fold_cls_kv = False          # the trimming alone (the K / V fold of the trimmed layer is a different arithmetic: its own test)
    try:
        a = m2.img_txt_fusion_val(zt, feats[1:], cap)
        eng.trim_last = False
        b = m2.img_txt_fusion_val(zt, feats[1:], cap)
    finally:
        eng.trim_last, eng.fold_cls_kv = True, True
    c = m2.img_txt_fusion_val(zt, feats[1:], cap)      # trimmed + folded: within the operand rounding of the projected path
    print(f"\n[cls trim] max|d| {(a - b).abs().max().item():.3e}; folded vs projected {(c - a).abs().max().item():.3e}")
    assert torch.allclose(a, b, atol=1e-4, rtol=0)
    assert (c - a).abs().max().item() < REL_TOL[dt] * max(float(a.std()), 1e-3) + 1e-3


def test_unfolded_merge_matches_folded(cuda):
    z = H.load("tiny_loop.npz")
    g, v = H.geometry(json.loads(str(z["bert_cfg"])), json.loads(str(z["vit_cfg"])))
    outs = []
    for fold in (True, False):
        m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), torch.bfloat16, cuda, fold_merge=fold)
        feats = m2.img_embed(H.fixture_images(z, range(8), v.image_size).cuda())
        cap = [str(z["cirr_caps"][2])]
        zt = m1.img_txt_fusion(feats[:1], None, cap, train=False, return_raw=True)
        outs.append(m2.img_txt_fusion_val(zt, feats[1:], cap).cpu().numpy())
    d = np.abs(outs[0] - outs[1]).max()
    print(f"\n[fold vs unfold] max|d| {d:.3e} (logit sigma {outs[1].std():.3f})")
    assert d < 0.05 * outs[1].std()                       # the two schedules round Wm*W0 differently (bf16); measured 0.016 sigma


# ------------------------------------------------------------------------------------------------ reference geometry
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("tag", ["full224", "full224_spread"])
def test_full224(cuda, tag, dtype):
    z = H.load(tag + ".npz")
    g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), dtype, cuda)
    k = int(z["k"])
    feats = m2.img_embed(synthetic.images(range(k + 1), 224).cuda())
    e_vit = np.abs(feats[:, :4, :16].cpu().numpy() - z["vit_slice"]).max()
    cap = [synthetic.caption_text(0, 30)]
    zt = m1.img_txt_fusion(feats[:1], None, cap, train=False, return_raw=True)
    e_zt = np.abs(zt.last_hidden_state[0, 0].cpu().numpy() - z["z_t_cls"]).max()
    ids, mask = __import__("candidate_reranking_cir_amd.blip_stage2", fromlist=["encode_text"]).encode_text(m2.tokenizer, cap, cuda)
    taps = []
    logits = m2.score(zt.last_hidden_state, ids, mask, feats[1:], torch.zeros(k, dtype=torch.int64), taps=taps).cpu().numpy()
    e_tap = max(np.abs(torch.stack([t[b] for t in taps]).cpu().numpy() - z[f"taps{b}"]).max() for b in (0, 1))
    e_log = np.abs(logits - z["logits"]).max()
    tol = LOGIT_TOL[tag][dtype]
    exact = float((np.argsort(-logits, kind="stable") == z["order"]).mean())
    decided, total, ok = pair_order(logits, z["logits"], tol)
    print(f"\n[{tag} {dtype}] vit {e_vit:.3e}  z_t {e_zt:.3e}  taps {e_tap:.3e}  logits {e_log:.3e} "
          f"(std {z['logits'].std():.3f}, tol {tol:.1e})  exact-rank match {exact:.2f}  decided pairs {decided}/{total}")
    assert e_vit < TOK_TOL[dtype] and e_zt < TOK_TOL[dtype] and e_tap < TOK_TOL[dtype] * 1.5
    assert e_log < tol
    assert ok and decided >= (30 if dtype == HF else 9)     # of 45 pairs (bf16 on these clustered logits: few, see rank224)
    # exact sorted order of the K = 10 candidates: fp16 operands reproduce it; bf16 measured 0.80 / 0.60 (test / spread weights:
    # one resp. two swaps of neighbours whose reference gap is below the bf16 drift)
    # (10 clustered logits, sigma 0.026: bf16 0.4 - 0.6 across the rounds' rounding changes; fp16 1.0 through round 5 and 0.8 - ONE swap of
    #  neighbours - since round 6's lazy softmax rescale re-rolled the ViT's roundings: what fp16 must hold is every pair the reference separates
    #  by more than the logit tolerance itself, i.e. any position that differs is a neighbour swap inside the tolerance)
    assert exact >= (0.8 if dtype == HF else 0.4)
    if dtype == HF:
        iu = np.triu_indices(len(logits), 1)
        d_ref, d_our = (z["logits"][:, None] - z["logits"][None, :])[iu], (logits[:, None] - logits[None, :])[iu]
        sep = np.abs(d_ref) > tol
        assert np.all(np.sign(d_ref[sep]) == np.sign(d_our[sep])) and int(sep.sum()) >= 38


def test_full384_cirr_loop(cuda):
    """The reference's real geometry (384 px, 577 tokens) through its own extract_index_features +
    generate_cirr_val_predictions (golden), against our batched loop."""
    from candidate_reranking_cir_amd import validate_stage2 as V
    z = H.load("full384.npz")
    g, v = H.geometry(H.FULL_BERT, dict(image_size=384))
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), torch.bfloat16, cuda)
    bank = V.extract_index_features(synthetic.images(range(7), 384), m2, batch_size=4)
    assert bank.shape == (7, 577, 768)
    e_bank = np.abs(bank[:, :3, :8].float().cpu().numpy() - z["index_features_slice"]).max()
    ds = V.RelativeValSet(ref_index=z["refs"], cand_index=z["cand_idx"], labels=z["labels"], captions=[str(c) for c in z["cirr_caps"]],
                          group_index=z["groups"], target_index=z["targets"])
    logits, glogits = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=2)
    skipped = ~z["labels"].any(1)
    assert np.all(logits.cpu().numpy()[skipped] == np.float32(-99999.99))
    e1 = np.abs(logits.cpu().numpy()[~skipped] - z["cirr_logits"][~skipped]).max()
    e2 = np.abs(glogits.cpu().numpy() - z["cirr_group_logits"]).max()
    print(f"\n[full384] bank {e_bank:.3e} logits {e1:.3e} group logits {e2:.3e}")
    assert e_bank < TOK_TOL[BF] and e1 < LOGIT_TOL["full384"][BF] and e2 < LOGIT_TOL["full384"][BF]


def test_full_size_properties(cuda):
    """BASELINE-size batch (queries x K=100 candidates from pixels, 224 px, the shapes bench.py runs: every large Linear
    goes through the persistent 256 x 256 GEMM) checked through size-independent properties instead of an oracle run:
      * rows are independent: permuting the candidates permutes the logits BIT FOR BIT (no kernel reduces across rows,
        and a row's reduction order does not depend on where the row sits);
      * batch invariance: a candidate scored in a small batch (every Linear on the 128 x 128 GEMM instead of the persistent
        256 x 256 one) gets the SAME BITS - tokens and logits - with the fp16 and with the fp32 residual stream (round 4: the two
        GEMM kernels share their accumulation order and epilogue arithmetic), so a query block may be split over ranks in
        any sizes without changing a result;
      * the order returned by cir_topk_desc is a permutation that sorts the logits (ties only among equal values)."""
    from candidate_reranking_cir_amd import ops
    g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
    m2, m1 = build_models(g, v, 0, "test", torch.bfloat16, cuda)
    q_n, k = 4, 100
    gen = torch.Generator(device="cuda").manual_seed(99)
    images = torch.randn((q_n + q_n * k, 3, 224, 224), generator=gen, device="cuda").bfloat16()
    ids = torch.stack([synthetic.caption_ids(q, 32) for q in range(q_n)]).cuda()
    mask = torch.ones_like(ids)
    qidx = torch.arange(q_n, device="cuda").repeat_interleave(k)
    toks = m2.img_embed16(images)
    z = m1.z_t(toks[:q_n], ids, mask).last_hidden_state
    logits = m2.score(z, ids, mask, toks[q_n:], qidx)
    assert logits.shape == (q_n * k,) and torch.isfinite(logits).all()
    # permutation of the candidates (and of their query assignment with them)
    perm = torch.randperm(q_n * k, generator=torch.Generator().manual_seed(5)).cuda()
    logits_p = m2.score(z, ids, mask, toks[q_n:][perm], qidx[perm])
    assert torch.equal(logits_p, logits[perm])
    # a permutation of the images through the ViT as well
    iperm = torch.randperm(q_n * k, generator=torch.Generator().manual_seed(6)).cuda()
    toks_p = m2.img_embed16(images[q_n:][iperm])
    assert torch.equal(toks_p, toks[q_n:][iperm])
    # batch invariance against a small batch (the other GEMM kernel / tiling): bit for bit, both residual-stream formats
    for sdt in (None, torch.float32):
        m2.set_stream_dtype(sdt); m1.set_stream_dtype(sdt)
        big_t = m2.img_embed16(images[q_n:])
        big = m2.score(z, ids, mask, big_t, qidx)
        small_t = m2.img_embed16(images[q_n:q_n + 6])
        small = m2.score(z[:1], ids[:1], mask[:1], small_t, torch.zeros(6, dtype=torch.int64, device="cuda"))
        e_t, e = (small_t.float() - big_t[:6].float()).abs().max().item(), (small - big[:6]).abs().max().item()
        print(f"\n[full size, stream {m2.stream_dtype}] small batch vs large batch: tokens {e_t:.3e} logits {e:.3e}")
        assert torch.equal(small_t, big_t[:6]) and torch.equal(small, big[:6])
        # ... and a split of the candidates into two unequal parts (one on each side of the tile heuristic)
        cut = q_n * k - 37
        part = torch.cat([m2.score(z, ids, mask, big_t[:cut], qidx[:cut]), m2.score(z, ids, mask, big_t[cut:], qidx[cut:])])
        assert torch.equal(part, big)
    m2.set_stream_dtype(None); m1.set_stream_dtype(None)
    # PARITY at this size (round 5): the exact mode (fp32; pinned to the reference by tests/test_exact_gpu.py) referees the bf16 logits of the
    # same 400 candidates from the same pixels - i.i.d. noise images on "test" weights: clustered logits, the hardest case for a rank statistic
    from scipy.stats import kendalltau
    for m in (m2, m1):
        m.set_precision("exact")
    toks_x = m2.img_embed16(images.float())
    logits_x = m2.score(m1.z_t(toks_x[:q_n], ids, mask).last_hidden_state, ids, mask, toks_x[q_n:], qidx)
    for m in (m2, m1):
        m.set_precision("bf16").set_stream_dtype(torch.float16)
    a, b = logits.view(q_n, k).cpu().numpy(), logits_x.view(q_n, k).cpu().numpy()
    e_x = np.abs(a - b).max()
    taus = [kendalltau(x, y).statistic for x, y in zip(a, b)]
    print(f"\n[full size] bf16 vs the exact mode: max|dlogit| {e_x:.2e} (sigma per query {b.std(axis=1).mean():.3f})  tau {np.mean(taus):.4f} (worst {np.min(taus):.4f})  "
          f"top-1 agree {np.mean([x.argmax() == y.argmax() for x, y in zip(a, b)]):.2f}")
    assert e_x < LOGIT_TOL["full224"][BF] and np.mean(taus) >= 0.90
    del toks_x
    # ranking
    lv = logits.view(q_n, k)
    order = ops.argsort_desc(lv)
    assert torch.equal(torch.sort(order, dim=1).values, torch.arange(k, device="cuda").expand(q_n, k))
    sorted_l = torch.gather(lv, 1, order)
    assert (sorted_l[:, 1:] <= sorted_l[:, :-1]).all()



# ------------------------------------------------------------------------------------------------ rank order at K = 100 / 50 / 200
@pytest.fixture(scope="module", params=[BF, HF, DEF], ids=["bf16", "fp16", "default"])
def rank(request, cuda):
    from candidate_reranking_cir_amd import validate_stage2 as V
    z = H.load("rank224.npz")
    g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), request.param, cuda)
    imgs = synthetic.scene_images(range(int(z["n_index"])), 224)
    bank = V.extract_index_features(imgs, m2, batch_size=128)
    return z, m2, m1, bank, request.param


def _rank_asserts(tag, logits, ref, tol, min_decided_frac=0.5):
    e = np.abs(logits - ref)
    ec = np.abs((logits - ref) - (logits - ref).mean(axis=1, keepdims=True))
    n_dec = n_tot = n_fix = 0
    for q in range(len(ref)):
        d, t, ok = pair_order(logits[q], ref[q], tol)
        assert ok, f"{tag} query {q}: a decided pair changed order"
        nf, okf = fixed_positions(logits[q], ref[q], tol)
        assert okf, f"{tag} query {q}: a margin-separated sorted position holds another candidate"
        n_dec, n_tot, n_fix = n_dec + d, n_tot + t, n_fix + nf
    print(f"   [{tag}] max|dlogit| {e.max():.3e} (centred {ec.max():.3e}, sigma {ref.std(axis=1).mean():.3f}, tol {tol:.1e})  "
          f"decided pairs {n_dec}/{n_tot} = {n_dec / n_tot:.2f}  fixed positions {n_fix}")
    assert e.max() < tol
    assert n_dec >= min_decided_frac * n_tot, "rank check went vacuous"
    return n_fix


def test_rank_bank_tokens(rank):
    z, m2, m1, bank, dt = rank
    assert bank.shape == (int(z["n_index"]), 197, 768)
    err = np.abs(bank[:, :3, :8].float().cpu().numpy() - z["bank_slice"]).max()
    print(f"\n[rank224 bank {dt}] max|err| {err:.3e}")
    assert err < TOK_TOL[dt]


@pytest.mark.parametrize("tag", ["c100", "c200"])
def test_rank_order_cirr_k100_k200(rank, tag):
    """BASELINE configs[2] / the K=200 half of configs[4]: the reference's generate_cirr_val_predictions +
    compute_cirr_val_metrics at K=100 (+5 subset, one skipped row) and K=200 (+5) on separated logits."""
    from candidate_reranking_cir_amd import validate_stage2 as V
    z, m2, m1, bank, dt = rank
    ds = V.RelativeValSet(ref_index=z[f"{tag}_refs"], cand_index=z[f"{tag}_cand"], labels=z[f"{tag}_labels"],
                          captions=[str(c) for c in z[f"{tag}_caps"]], group_index=z[f"{tag}_groups"], target_index=z[f"{tag}_targets"])
    logits_t, glogits_t = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=4)
    logits, glogits = logits_t.cpu().numpy(), glogits_t.cpu().numpy()
    ref, gref = z[f"{tag}_logits"], z[f"{tag}_group_logits"]
    skipped = ~z[f"{tag}_labels"].any(1)
    assert np.array_equal(logits[skipped], ref[skipped])                         # -99999.99 rows, bit for bit
    tol = LOGIT_TOL["rank224"][dt]
    print(f"\n[rank224 {tag} {dt}]")
    n_fix = _rank_asserts(tag, logits[~skipped], ref[~skipped], tol)
    _rank_asserts(tag + " subset", glogits, gref, tol, min_decided_frac=0.5)
    assert n_fix >= {"c100": {BF: 2, HF: 20, DEF: 8}, "c200": {BF: 0, HF: 8, DEF: 2}}[tag][dt]      # (measured 4 / 31 and 0 / 12)
    # the target scored as a top-K candidate and as a subset member is the same image through the same z_t
    for q in np.where(~skipped)[0]:
        ci, gi = int(z[f"{tag}_labels"][q].argmax()), int(np.where(z[f"{tag}_groups"][q] == z[f"{tag}_targets"][q])[0][0])
        assert abs(logits[q, ci] - glogits[q, gi]) < 1e-5
    # Recall@1/5/10/50 and Recall_subset@1/2/3 from OUR logits == the reference's tuple (labels sit on margin-decided ranks)
    ours = V.compute_cirr_val_metrics(logits_t, glogits_t, ds)
    print("   recall ours", np.round(ours, 2), "reference", np.round(z[f"{tag}_metrics"], 2))
    np.testing.assert_allclose(ours, z[f"{tag}_metrics"], atol=1e-4)


def test_rank_kv_bank_path_against_reference(rank):
    """SURVEY 8(f)-1 at the benchmark geometry: K = 100 (+5) scored straight out of the per-image K/V bank of the 256-image
    index (224 px, 197 tokens) - against the reference's own logits, and bit for bit against the per-candidate path."""
    from candidate_reranking_cir_amd import validate_stage2 as V
    z, m2, m1, bank, dt = rank
    ds = V.RelativeValSet(ref_index=z["c100_refs"], cand_index=z["c100_cand"], labels=z["c100_labels"],
                          captions=[str(c) for c in z["c100_caps"]], group_index=z["c100_groups"], target_index=z["c100_targets"])
    kvb = m2.build_kv_bank(bank)
    lt, gt = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=4, kv_bank=kvb)
    eng = m2.engines()[1]
    fold_was = eng.fold_cross_kv          # the bank holds PROJECTED keys / values: bit identity is with the projected per-candidate path
    eng.fold_cross_kv = False
    plain = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=4)
    eng.fold_cross_kv = fold_was
    assert torch.equal(lt, plain[0]) and torch.equal(gt, plain[1])
    if fold_was:                          # ... and the folded path (round 5's default at this geometry) agrees within the operand rounding
        folded = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=4)
        act_t = torch.as_tensor(z["c100_labels"].any(1))
        assert (folded[0] - lt)[act_t].abs().max().item() < LOGIT_TOL["rank224"][dt] and (folded[1] - gt).abs().max().item() < LOGIT_TOL["rank224"][dt]
    del kvb
    ref, gref, skipped = z["c100_logits"], z["c100_group_logits"], ~z["c100_labels"].any(1)
    logits = lt.cpu().numpy()
    assert np.array_equal(logits[skipped], ref[skipped])
    e = max(np.abs(logits[~skipped] - ref[~skipped]).max(), np.abs(gt.cpu().numpy() - gref).max())
    print(f"\n[rank224 c100 K/V bank {dt}] max|dlogit| vs reference {e:.3e}")
    assert e < LOGIT_TOL["rank224"][dt]
    np.testing.assert_allclose(V.compute_cirr_val_metrics(lt, gt, ds), z["c100_metrics"], atol=1e-4)


def test_rank_order_fiq_k50(rank):
    """BASELINE configs[1]: FashionIQ-style K=50 (two captions joined as validate_stage2.py:97-100) through the reference's
    generate_fiq_val_predictions / compute_fiq_val_metrics."""
    from candidate_reranking_cir_amd import validate_stage2 as V
    z, m2, m1, bank, dt = rank
    caps = [V.fiq_caption(str(p[0]), str(p[1])) for p in z["f50_caps"]]
    ds = V.RelativeValSet(ref_index=z["f50_refs"], cand_index=z["f50_cand"], labels=z["f50_labels"], captions=caps)
    logits_t = V.generate_fiq_val_predictions(m2, m1, ds, bank, query_batch=2)
    print(f"\n[rank224 f50 {dt}]")
    _rank_asserts("f50", logits_t.cpu().numpy(), z["f50_logits"], LOGIT_TOL["rank224"][dt])
    ours = V.compute_fiq_val_metrics(logits_t, ds)
    np.testing.assert_allclose(ours, z["f50_metrics"], atol=1e-4)


def order_stats(ours: np.ndarray, ref: np.ndarray):
    """(exact-position fraction, Kendall tau, top-10 overlap fraction) of two logit rows."""
    from scipy.stats import kendalltau
    o, r = np.argsort(-ours, kind="stable"), np.argsort(-ref, kind="stable")
    return float((o == r).mean()), float(kendalltau(ours, ref).statistic), len(set(o[:10]) & set(r[:10])) / 10.0


# ------------------------------------------------------------------------------------------------ rank identity floors
RANK_FLOORS = {          # (exact-position fraction, Kendall tau, top-10 overlap) floors = measured on MI355X minus ~10 %:
    # measured       bf16: c100 0.752 / 0.9933 / 0.97   c200 0.510 / 0.9915 / 0.95   f50 0.827 / 0.9924 / 1.00
    #                fp16: c100 0.945 / 0.9989 / 1.00   c200 0.927 / 0.9992 / 1.00   f50 0.987 / 0.9995 / 1.00
    #             default (fp16 operands + fp16 streams, round 4): c100 0.922-0.932 / 0.9984 / 1.00   c200 0.775 / 0.9974 / 1.00
    #             f50 0.940 / 0.9962 / 1.00 - the c100 floor is the acceptance bar of the round-3 review: >= 0.90 of the sorted positions
    #             hold exactly the reference's candidate at K = 100
    "c100": {BF: (0.67, 0.989, 0.87), HF: (0.85, 0.997, 0.9), DEF: (0.90, 0.997, 0.9)},
    "c200": {BF: (0.40, 0.987, 0.85), HF: (0.83, 0.997, 0.9), DEF: (0.72, 0.996, 0.9)},   # (bf16: 0.51 projected, 0.44 with round 5's folded cross-attention - 2 scored queries)
    "f50": {BF: (0.74, 0.988, 0.9), HF: (0.88, 0.996, 0.9), DEF: (0.88, 0.995, 0.9)},   # (fp16 strict: 0.920 / 0.9967 since round 4's epilogue unification)
}


@pytest.mark.parametrize("tag", ["c100", "c200", "f50"])
def test_rank_identity_floors(rank, tag):
    from candidate_reranking_cir_amd import validate_stage2 as V
    z, m2, m1, bank, dt = rank
    if tag == "f50":
        caps = [V.fiq_caption(str(p[0]), str(p[1])) for p in z["f50_caps"]]
        ds = V.RelativeValSet(ref_index=z["f50_refs"], cand_index=z["f50_cand"], labels=z["f50_labels"], captions=caps)
        logits = V.generate_fiq_val_predictions(m2, m1, ds, bank, query_batch=3).cpu().numpy()
    else:
        ds = V.RelativeValSet(ref_index=z[f"{tag}_refs"], cand_index=z[f"{tag}_cand"], labels=z[f"{tag}_labels"],
                              captions=[str(c) for c in z[f"{tag}_caps"]], group_index=z[f"{tag}_groups"], target_index=z[f"{tag}_targets"])
        logits = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=4)[0].cpu().numpy()
    ref = z[f"{tag}_logits"]
    active = z[f"{tag}_labels"].any(1)
    stats = np.array([order_stats(logits[q], ref[q]) for q in np.where(active)[0]])
    exact, tau, top10 = stats.mean(0)
    f_exact, f_tau, f_top = RANK_FLOORS[tag][dt]
    print(f"\n[rank224 {tag} {dt}] exact positions {exact:.3f} (floor {f_exact})  Kendall tau {tau:.4f} (floor {f_tau})  "
          f"top-10 overlap {top10:.2f} (floor {f_top})  worst query tau {stats[:, 1].min():.4f}")
    assert exact >= f_exact and tau >= f_tau and top10 >= f_top



# Round 5: the fixture regrown to 16 SCORED queries per case (tests/golden/rank224_wide.npz: the same weights and bank, so this shares the
# models / index features of `rank`): 1600 / 3200 / 800 sorted positions instead of 400 / 400 / 150 - one candidate is 0.006 of a top-10
# statistic instead of 0.05 - and the top-10 floor is back at >= 0.9 in every mode.  Measured on MI355X (profiles/r5_precision_modes.json):
#   bf16 + fp16 streams      c100 0.722 / 0.9927 / 0.988   c200 0.560 / 0.9929 / 0.981   f50 0.875 / 0.9946 / 0.994
#   fp16 + fp32 streams      c100 0.956 / 0.9991 / 0.994   c200 0.914 / 0.9991 / 1.000   f50 0.980 / 0.9992 / 1.000
#   default (fp16 + fp16)    c100 0.893 / 0.9977 / 0.994   c200 0.815 / 0.9978 / 0.994   f50 0.968 / 0.9987 / 1.000
#   exact (fp32, tests/test_exact_gpu.py)  1.000 / 1.0000 / 1.000   0.999 / 1.0000 / 1.000   1.000 / 1.0000 / 1.000
WIDE_FLOORS = {
    "c100": {BF: (0.65, 0.991, 0.93), HF: (0.92, 0.9985, 0.95), DEF: (0.85, 0.997, 0.95)},
    "c200": {BF: (0.50, 0.991, 0.93), HF: (0.87, 0.9985, 0.95), DEF: (0.77, 0.997, 0.95)},
    "f50": {BF: (0.80, 0.993, 0.93), HF: (0.94, 0.9985, 0.95), DEF: (0.93, 0.9975, 0.95)},   # (default tau 0.9980 with norm1 folded into the qkv GEMM, 0.9983 before: one rounding realisation against another)
}


@pytest.mark.parametrize("tag", ["c100", "c200", "f50"])
def test_rank_identity_floors_wide(rank, tag):
    from candidate_reranking_cir_amd import validate_stage2 as V
    _, m2, m1, bank, dt = rank
    z = H.load("rank224_wide.npz")
    if tag == "f50":
        caps = [V.fiq_caption(str(p[0]), str(p[1])) for p in z["f50_caps"]]
        ds = V.RelativeValSet(ref_index=z["f50_refs"], cand_index=z["f50_cand"], labels=z["f50_labels"], captions=caps)
        lt = V.generate_fiq_val_predictions(m2, m1, ds, bank, query_batch=4)
        ours = V.compute_fiq_val_metrics(lt, ds)
    else:
        ds = V.RelativeValSet(ref_index=z[f"{tag}_refs"], cand_index=z[f"{tag}_cand"], labels=z[f"{tag}_labels"],
                              captions=[str(c) for c in z[f"{tag}_caps"]], group_index=z[f"{tag}_groups"], target_index=z[f"{tag}_targets"])
        lt, gt = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=4)
        ours = V.compute_cirr_val_metrics(lt, gt, ds)
    logits, ref = lt.cpu().numpy(), z[f"{tag}_logits"]
    active = z[f"{tag}_labels"].any(1)
    assert int(active.sum()) == 16 and np.array_equal(logits[~active], ref[~active])
    stats = np.array([order_stats(logits[q], ref[q]) for q in np.where(active)[0]])
    exact, tau, top10 = stats.mean(0)
    f_exact, f_tau, f_top = WIDE_FLOORS[tag][dt]
    err = np.abs(logits[active] - ref[active]).max()
    print(f"\n[rank224_wide {tag} {dt}] max|dlogit| {err:.2e}  exact positions {exact:.3f} (floor {f_exact})  tau {tau:.4f} (floor {f_tau}, worst query "
          f"{stats[:, 1].min():.4f})  top-10 {top10:.3f} (floor {f_top})")
    assert err < LOGIT_TOL["rank224"][dt] * 1.25 and exact >= f_exact and tau >= f_tau and top10 >= f_top
    # labels sit on margin-decided ranks (4 x the bf16 bound): every mode reproduces the reference's recall tuple
    np.testing.assert_allclose(ours, z[f"{tag}_metrics"], atol=1e-4)


@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
def test_img_txt_fusion_bxb_matches_reference(cuda, dtype):
    """SURVEY 8(f)-4, forward half: the training-mode surface img_txt_fusion (blip_stage2.py:65-99; row i's caption and
    z_t against all B candidates, ragged captions -> padding='longest' masks inside the batch) in eval mode against the
    reference's (B, B) logits, z_t from our stage I on the padded batch as stage2_train.py:202-207 does."""
    z = H.load("bxb224.npz")
    g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), dtype, cuda)
    bank = m2.img_embed(synthetic.scene_images(range(8), 224).cuda())
    caps = [str(c) for c in z["caps"]]
    zt = m1.img_txt_fusion(bank[:4], bank[:4], caps, train=False, return_raw=True)
    e_z = np.abs(zt.last_hidden_state[:, 0].cpu().numpy() - z["z_t_cls"]).max()
    out = m2.img_txt_fusion(zt, bank[4:8], caps, train=True)
    assert out.shape == (4, 4) and out.dtype == torch.float32
    e = np.abs(out.cpu().numpy() - z["logits"]).max()
    print(f"\n[bxb224 {dtype}] z_t cls {e_z:.3e}  logits {e:.3e} (sigma {z['logits'].std():.3f})")
    assert e_z < TOK_TOL[dtype] and e < LOGIT_TOL["bxb224"][dtype]
    m2.train()                                          # training mode: the differentiable surface (tests/test_train_gpu.py); with
    with torch.no_grad():                               # autograd off it is the same inference path
        assert torch.equal(m2.img_txt_fusion(zt, bank[4:8], caps), out)
    m2.eval()


@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
def test_vit_large_width(cuda, dtype):
    """`vit='large'` (blip.py:203-209: width 1024, 16 heads, mlp 4096) is accepted by the factories: its widths go
    through every kernel on the ViT path (1024-column LayerNorm, N = 3072 / 1024 / 4096 GEMMs, 16-head attention).
    Checked at 3 of the 24 blocks against the fp32 oracle (no reference golden: the oracle is pinned on ViT-B)."""
    from candidate_reranking_cir_amd import config, weights
    from candidate_reranking_cir_amd.engine import VitEngine
    from oracle import cir_oracle as O
    v = config.VitGeometry(image_size=224, width=1024, depth=3, num_heads=16)
    spec = {k: sh for k, sh in weights.nlvr_param_spec(config.BertGeometry(encoder_width=1024), v).items() if k.startswith("visual_encoder.")}
    sd = weights.synth_state_dict(spec, 5, "test")
    imgs = synthetic.scene_images(range(3), 224)
    eng = VitEngine({k: t.cuda() for k, t in sd.items()}, v, dtype, cuda, stream_dtype=torch.float16 if dtype == torch.bfloat16 else torch.float32)
    y32, y16 = eng.forward(imgs.cuda(), want32=True)
    with torch.no_grad():
        ref = O.vit_forward(sd, imgs, n_heads=16)
    err = (y32.cpu() - ref).abs().max().item()
    print(f"\n[vit-large width {dtype}] max|err| {err:.3e}")
    assert y32.shape == (3, 197, 1024) and err < TOK_TOL[dtype]


@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
def test_last_layer_kv_fold_matches_projected_path(cuda, dtype):
    """The last fusion layer with the cross K / V projections folded out of the token side (cir_cls_cross_attention + two
    small GEMMs per branch) against the same layer through the K|V GEMM + attention (`fold_cls_kv = False`): same logits
    within the operand rounding - and both within the fixture's bound of the reference."""
    z = H.load("full224.npz")
    g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), dtype, cuda)
    k = int(z["k"])
    feats = m2.img_embed(synthetic.images(range(k + 1), 224).cuda())
    cap = [synthetic.caption_text(0, 30)]
    zt = m1.img_txt_fusion(feats[:1], None, cap, train=False, return_raw=True)
    eng = m2.engines()[1]
    assert eng.cls_fold is not None and eng.fold_cls_kv
    a = m2.img_txt_fusion_val(zt, feats[1:], cap).cpu().numpy()
    eng.fold_cls_kv = False
    b = m2.img_txt_fusion_val(zt, feats[1:], cap).cpu().numpy()
    eng.fold_cls_kv = True
    tol = LOGIT_TOL["full224"][dtype]
    print(f"\n[last-layer K/V fold {dtype}] folded vs projected {np.abs(a - b).max():.3e}; vs reference {np.abs(a - z['logits']).max():.3e} / {np.abs(b - z['logits']).max():.3e}")
    assert np.abs(a - b).max() < tol and np.abs(a - z["logits"]).max() < tol and np.abs(b - z["logits"]).max() < tol


def test_edge_cases_k1_all_skipped_empty_and_long_caption(tiny):
    """Edges the reference's loop handles explicitly: K == 1 (its `unsqueeze` special case, validate_stage2.py:247-250),
    a dataset whose every row is skipped, an empty shard, B = 1 in the B x B surface, and a caption longer than the
    512-row position table (the reference raises on the position_ids slice; here an IndexError before any launch)."""
    from candidate_reranking_cir_amd import validate_stage2 as V
    from oracle import cir_oracle as O
    z, g, v, m2, m1, dt = tiny
    imgs = H.fixture_images(z, range(14), v.image_size)
    bank = V.extract_index_features(imgs, m2)
    caps = [str(c) for c in z["cirr_caps"]][:3]
    # K = 1 against the oracle
    ds = V.RelativeValSet(ref_index=np.array([0, 5, 9]), cand_index=np.array([[3], [7], [1]]), labels=np.ones((3, 1), dtype=bool), captions=caps)
    out = V.generate_val_predictions(m2, m1, ds, bank, query_batch=2).cpu()
    assert out.shape == (3, 1)
    sd2, sd1 = H.state_dicts(g, v, int(z["seed"]), str(z["profile"]))
    with torch.no_grad():
        feats = O.img_embed(sd2, imgs)
        for q in range(3):
            ids, mask = H.tokenize([caps[q]])
            zt = O.stage1_z_t(sd1, feats[ds.ref_index[q]][None], ids, mask)
            ref = O.img_txt_fusion_val(sd2, zt, feats[ds.cand_index[q]], ids, mask)
            assert abs(out[q, 0].item() - ref[0].item()) < (3e-2 if dt == BF else 4e-3)
    # every row skipped, no subset: nothing is launched, the matrix is the fill value
    ds0 = V.RelativeValSet(ref_index=np.array([0, 1]), cand_index=np.array([[2, 3], [4, 5]]), labels=np.zeros((2, 2), dtype=bool), captions=caps[:2])
    out0 = V.generate_val_predictions(m2, m1, ds0, bank)
    assert out0.shape == (2, 2) and bool((out0 == np.float32(-99999.99)).all())
    # an empty shard of a real dataset
    oute = V.generate_val_predictions(m2, m1, ds, bank, rows=[])
    assert oute.shape == (0, 1)
    # B = 1 through the B x B surface == the K = 1 validation call
    feats32 = m2.img_embed(imgs[:2].cuda())
    zt = m1.img_txt_fusion(feats32[:1], None, caps[:1], train=False, return_raw=True)
    bb = m2.img_txt_fusion(zt, feats32[1:2], caps[:1])
    assert bb.shape == (1, 1) and torch.equal(bb[0], m2.img_txt_fusion_val(zt, feats32[1:2], caps[:1]))
    # caption longer than the position table
    from candidate_reranking_cir_amd.blip_stage2 import encode_text
    long_ids = torch.full((1, g.max_position_embeddings + 1), 1000, dtype=torch.int64)
    with pytest.raises(IndexError, match="position-embedding table"):
        m1.z_t(feats32[:1], *encode_text(m1.tokenizer, {"input_ids": long_ids, "attention_mask": torch.ones_like(long_ids)}, "cuda"))


@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
def test_residual_stream_fp32_mode(cuda, dtype):
    """The residual stream is stored in fp16 with bf16 operands and in fp32 with fp16 operands by default (sum in fp32, one
    rounding per sublayer); `set_stream_dtype` overrides.  With bf16 operands both modes meet the same bounds against the
    reference; with fp16 operands the fp16 stream costs about 2x on the logits and 4x on the tokens (bounds scaled here)."""
    z = H.load("full224.npz")
    g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), dtype, cuda)
    assert m2.stream_dtype == (torch.float16 if dtype == BF else torch.float32)
    k = int(z["k"])
    cap = [synthetic.caption_text(0, 30)]
    outs = {}
    for sdt in (torch.float16, torch.float32):
        m2.set_stream_dtype(sdt); m1.set_stream_dtype(sdt)
        feats = m2.img_embed(synthetic.images(range(k + 1), 224).cuda())
        zt = m1.img_txt_fusion(feats[:1], None, cap, train=False, return_raw=True)
        outs[sdt] = m2.img_txt_fusion_val(zt, feats[1:], cap).cpu().numpy()
        e_vit = np.abs(feats[:, :4, :16].cpu().numpy() - z["vit_slice"]).max()
        e_log = np.abs(outs[sdt] - z["logits"]).max()
        print(f"\n[stream {sdt} / operands {dtype}] vit {e_vit:.3e} logits {e_log:.3e}")
        loose = 2.5 if (dtype == HF and sdt == torch.float16) else 1.0
        assert e_vit < TOK_TOL[dtype] * loose and e_log < LOGIT_TOL["full224"][dtype] * loose
    assert np.abs(outs[torch.float16] - outs[torch.float32]).max() < LOGIT_TOL["full224"][dtype]
    m2.set_stream_dtype(None); m1.set_stream_dtype(None)


def test_state_dict_roundtrip_and_cpu_refusal(cuda):
    from candidate_reranking_cir_amd.blip_stage2 import blip_stage2
    z = H.load("tiny_loop.npz")
    g, v = H.geometry(json.loads(str(z["bert_cfg"])), json.loads(str(z["vit_cfg"])))
    m = blip_stage2(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    sd = m.state_dict()
    assert "text_encoder.encoder.layer.6.crossattention.output.merge_layer.weight" in sd
    assert "text_encoder.encoder.layer.5.crossattention.output.merge_layer.weight" not in sd
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.img_embed(torch.zeros(1, 3, v.image_size, v.image_size))
