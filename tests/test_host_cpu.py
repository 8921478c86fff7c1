"""Host-side logic that needs no GPU: metric code against the reference's own metric arithmetic
(tests/golden/metrics.npz), caption joining, tokenizer double, state-dict layout, config loader."""
import json
import os

import numpy as np
import pytest
import torch

from candidate_reranking_cir_amd import config, synthetic, weights
from candidate_reranking_cir_amd import validate_stage2 as V
from tests import helpers as H


def test_recall_metrics_match_reference_arithmetic():
    m = H.load("metrics.npz")
    ds = V.RelativeValSet(ref_index=np.zeros(len(m["labels"]), dtype=int), cand_index=np.zeros_like(m["labels"], dtype=int),
                          labels=m["labels"], group_index=m["group_members"], target_index=m["targets"])
    logits, glogits = torch.tensor(m["logits"]), torch.tensor(m["group_logits"])
    np.testing.assert_allclose(V.compute_fiq_val_metrics(logits, ds), m["fiq_metrics"], atol=1e-4)
    np.testing.assert_allclose(V.compute_cirr_val_metrics(logits, glogits, ds), m["cirr_metrics"], atol=1e-4)


def test_fiq_caption_rule():
    assert V.fiq_caption("is red.", "  has long sleeves?") == "Is red and has long sleeves"
    z = H.load("tiny_loop.npz")
    for a, b in z["fiq_caps"]:
        assert V.fiq_caption(str(a), str(b)) == H.fiq_caption((a, b))


def test_hash_tokenizer_padding_and_enc_token():
    tok = synthetic.HashTokenizer()
    enc = tok(["a b c", "a"], padding="longest", return_tensors="pt")
    assert enc.input_ids.shape == (2, 5) and enc.attention_mask.tolist() == [[1, 1, 1, 1, 1], [1, 1, 1, 0, 0]]
    assert enc.input_ids[0, 0] == 101 and enc.input_ids[0, -1] == 102 and enc.input_ids[1, 3] == 0
    assert enc.input_ids[0, 1] == enc.input_ids[1, 1]                 # same word -> same id
    assert tok.enc_token_id == 30523


def test_state_dict_layout_matches_reference_counts():
    g, v = config.BertGeometry(), config.VitGeometry.named("base", 384)
    s2, s1 = weights.nlvr_param_spec(g, v), weights.retrieval_param_spec(g, v)
    assert len(s2) == 723 and len(s1) == 472                           # SURVEY.md section 8(b) [probe]
    n2 = sum(int(np.prod(sh)) for k, (sh, kind) in s2.items() if kind != "position_ids")
    n1 = sum(int(np.prod(sh)) for k, (sh, kind) in s1.items() if kind != "position_ids")
    assert abs(n2 / 1e6 - 288.06) < 0.4 and abs(n1 / 1e6 - 223.45) < 0.4
    assert s2["visual_encoder.pos_embed"][0] == (1, 577, 768)
    assert "text_encoder.encoder.layer.6.crossattention.output.merge_layer.weight" in s2
    assert "text_encoder.encoder.layer.5.crossattention.output.merge_layer.weight" not in s2


def test_med_config_loader_and_factories(tmp_path):
    from candidate_reranking_cir_amd.blip_stage2 import load_bert_geometry
    g = load_bert_geometry("configs/med_config.json")                   # the reference's default path
    assert (g.hidden_size, g.num_attention_heads, g.num_hidden_layers, g.intermediate_size, g.vocab_size) == (768, 12, 12, 3072, 30524)
    assert g.layer_norm_eps == 1e-12
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(dict(hidden_size=768, num_attention_heads=8)))
    with pytest.raises(ValueError, match="head_dim"):
        load_bert_geometry(str(bad))
    with pytest.raises(AssertionError):
        config.VitGeometry.named("huge", 224)


def test_synth_weights_are_deterministic_by_name():
    a = weights.synth_tensor("cls_head.0.weight", (4, 8), "weight", seed=3)
    b = weights.synth_tensor("cls_head.0.weight", (4, 8), "weight", seed=3)
    c = weights.synth_tensor("cls_head.2.weight", (4, 8), "weight", seed=3)
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_length_buckets_never_pad():
    caps = ["a b c", "d e", "f g h", "i"]
    ds = V.RelativeValSet(ref_index=np.zeros(4, dtype=int), cand_index=np.zeros((4, 2), dtype=int), labels=np.ones((4, 2), dtype=bool), captions=caps)
    buckets = V._length_buckets(ds, synthetic.HashTokenizer(), range(4))
    assert buckets == {5: [0, 2], 4: [1], 3: [3]}



def test_init_tokenizer_offline_behaviour():
    """No WordPiece vocabulary offline: init_tokenizer raises unless the synthetic fallback is allowed."""
    from candidate_reranking_cir_amd.blip import init_tokenizer, create_vit
    tok = init_tokenizer(allow_fallback=True)
    assert tok.enc_token_id == 30523
    assert create_vit("base", 384)[1] == 768 and create_vit("large", 224)[0].depth == 24


def test_bank_rows_are_validated_on_the_host():
    """A candidate / reference / subset row outside the index bank raises before anything is launched (the
    gather kernel would clamp it to a silently wrong image)."""
    ds = V.RelativeValSet(ref_index=np.array([0, 3]), cand_index=np.array([[1, 2], [2, 9]]), labels=np.ones((2, 2), dtype=bool), captions=["a", "b"])
    V._check_bank_rows(ds, 10)
    with pytest.raises(IndexError, match="cand_index holds row 9"):
        V._check_bank_rows(ds, 9)
    ds.group_index = np.array([[0, 1, 2, 3, -1]] * 2)
    with pytest.raises(IndexError, match="group_index holds row -1"):
        V._check_bank_rows(ds, 10)


def test_layernorm_fold_and_operand_split_packings():
    """Host-side packings of round 5 (pure torch, no GPU): `ops.ln_fold_pack` - LayerNorm(x) W^T + b equals
    rstd (x Wg^T - mean colsum) + b' exactly on the packed (fp16-rounded) Wg - and `ops.split_weight` - rows [W_hi | W_hi | W_lo] whose
    hi + lo reconstructs the fp32 weight to 2^-21 (fp16 subnormals included)."""
    import torch
    from candidate_reranking_cir_amd import ops
    g = torch.Generator().manual_seed(0)
    k, n, m = 96, 40, 17
    w, b = torch.randn((n, k), generator=g) * 0.05, torch.randn((n,), generator=g)
    gamma, beta = 1 + 0.3 * torch.randn((k,), generator=g), 0.2 * torch.randn((k,), generator=g)
    x = (torch.randn((m, k), generator=g) * 2 + 0.5).half().double()
    wg, colsum, bias = ops.ln_fold_pack(w, b, gamma, beta)
    assert wg.dtype == torch.float16 and colsum.dtype == bias.dtype == torch.float32
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-6).rsqrt()
    folded = rstd * (x @ wg.double().t() - mean * colsum.double()) + bias.double()
    direct = ((x - mean) * rstd) @ wg.double().t() + (b.double() + w.double() @ beta.double())     # LayerNorm with gamma inside Wg
    assert (folded - direct).abs().max().item() < 1e-5                     # colsum / bias are stored in fp32
    ref = torch.nn.functional.layer_norm(x, (k,), gamma.double(), beta.double(), 1e-6) @ w.double().t() + b.double()
    assert (folded - ref).abs().max().item() < 2e-3 * ref.abs().max().item()                         # Wg's fp16 rounding
    w32 = torch.cat([w, w * 1e-3], dim=0).contiguous()                                               # second half: lo terms land in fp16's subnormals
    cat = ops.split_weight(w32)._split3
    assert cat.shape == (2 * n, 3 * k) and cat.dtype == torch.float16 and torch.equal(cat[:, :k], cat[:, k:2 * k])
    rec = cat[:, :k].double() + cat[:, 2 * k:].double()
    assert ((rec - w32.double()).abs() <= w32.double().abs() * 2.0 ** -21 + 2.0 ** -25).all()
