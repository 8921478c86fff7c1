"""The factories' `pretrained=` path against the reference's own loaders (blip_stage2.load_checkpoint,
blip.load_checkpoint): tests/golden/ckpt_tiny.npz holds what THEY made of a BLIP-base style file whose ViT had a larger
position grid (oracle/make_golden.py ckpt) - per-tensor checksums of the loaded models, the resized position embedding
and the reported missing / unexpected keys."""
import json
import os

import numpy as np
import pytest
import torch

from candidate_reranking_cir_amd import config as cfgmod, synthetic, weights
from candidate_reranking_cir_amd.blip_stage1 import blip_stage1
from candidate_reranking_cir_amd.blip_stage2 import blip_stage2
from tests import helpers as H


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    z = H.load("ckpt_tiny.npz")
    bert, vit, big = json.loads(str(z["bert_cfg"])), json.loads(str(z["vit_cfg"])), json.loads(str(z["file_vit_cfg"]))
    g = cfgmod.BertGeometry.from_dict(bert)
    base = weights.synth_state_dict(weights.retrieval_param_spec(g, cfgmod.VitGeometry(**big)), int(z["seed"]), str(z["profile"]))
    base["text_encoder.embeddings.position_ids"] = torch.arange(g.max_position_embeddings).unsqueeze(0)
    path = str(tmp_path_factory.mktemp("ckpt") / "blip_base_like.pth")
    torch.save({"model": base}, path)
    return z, bert, vit, path


def _checksums(model, names):
    sd = model.state_dict()
    return (np.array([sd[k].double().sum().item() for k in names]), np.array([sd[k].double().abs().sum().item() for k in names]))


def _seeded(model, spec_fn, g, v, seed):
    model.load_state_dict(weights.synth_state_dict(spec_fn(g, v), seed, "test"))


def test_stage2_pretrained_matches_reference_loader(case, capsys):
    z, bert, vit, path = case
    g, v = H.geometry(bert, vit)
    from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR
    from candidate_reranking_cir_amd.checkpoint import load_stage2_checkpoint
    model = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    _seeded(model, weights.nlvr_param_spec, g, v, int(z["model_seed"]))         # what the reference model held before loading
    model, msg = load_stage2_checkpoint(model, path)
    assert sorted(msg.missing_keys) == list(z["s2_missing"]) and sorted(msg.unexpected_keys) == list(z["s2_unexpected"])
    names = list(z["s2_names"])
    s, a = _checksums(model, names)
    np.testing.assert_allclose(s, z["s2_sum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(a, z["s2_abs"], rtol=0, atol=1e-9)
    assert np.array_equal(model.state_dict()["visual_encoder.pos_embed"].numpy(), z["s2_pos_embed"])
    # both branches start from the same single-branch weights
    sd = model.state_dict()
    k0 = "text_encoder.encoder.layer.0.crossattention.self0.key.weight"
    assert torch.equal(sd[k0], sd[k0.replace("self0", "self1")])
    # the factory prints what the reference's prints
    m = blip_stage2(pretrained=path, med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    assert (m.precision, m.compute_dtype, m.token_dtype, m.stream_dtype, m.vit_stream_dtype) == \
        ("text32", torch.float32, torch.float16, torch.float32, torch.float16)   # real weights: the text side at ~20 bits (round 5)
    out = capsys.readouterr().out
    assert "reshape position embedding from 36 to 16" in out and "missing keys:" in out and isinstance(m, BLIP_NLVR)


def test_stage1_pretrained_matches_reference_loader(case):
    z, bert, vit, path = case
    g, v = H.geometry(bert, vit)
    from candidate_reranking_cir_amd.blip_stage1 import BLIP_Retrieval
    from candidate_reranking_cir_amd.checkpoint import load_stage1_checkpoint
    model = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    _seeded(model, weights.retrieval_param_spec, g, v, int(z["model_seed"]) + 1)
    model, msg = load_stage1_checkpoint(model, path)
    assert sorted(msg.missing_keys) == list(z["s1_missing"]) and sorted(msg.unexpected_keys) == list(z["s1_unexpected"])
    s, a = _checksums(model, list(z["s1_names"]))
    np.testing.assert_allclose(s, z["s1_sum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(a, z["s1_abs"], rtol=0, atol=1e-9)


def test_bad_path_and_trained_checkpoint_formats(case, tmp_path):
    z, bert, vit, _ = case
    g, v = H.geometry(bert, vit)
    with pytest.raises(RuntimeError, match="checkpoint url or path is invalid"):      # no network on this path
        blip_stage2(pretrained="https://example.invalid/model_base.pth", med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    # a file written by the reference's training scripts ({'BLIP_NLVR': state_dict}, utils.py:145-150) loads as it is
    sd = weights.synth_state_dict(weights.nlvr_param_spec(g, v), 5, "test")
    path = str(tmp_path / "tuned.pth")
    torch.save({"BLIP_NLVR": sd}, path)
    m = blip_stage2(pretrained=path, med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    assert all(torch.equal(m.state_dict()[k], t) for k, t in sd.items())


def test_pretrained_without_a_real_vocabulary_is_refused(case):
    """Real weights scored on hashed token ids would rank garbage silently: with no WordPiece vocabulary available (offline)
    and no explicit tokenizer, the factories raise instead of falling back (reference: blip_stage2.py:38-44 fails too)."""
    z, bert, vit, path = case
    g, v = H.geometry(bert, vit)
    with pytest.raises(RuntimeError, match="WordPiece"):
        blip_stage2(pretrained=path, med_config=g, vit_geometry=v)
    with pytest.raises(RuntimeError, match="WordPiece"):
        blip_stage1(pretrained=path, med_config=g, vit_geometry=v)
