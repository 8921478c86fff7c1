"""z_t in `.train()` mode: the reference's loop leaves model_stage1.train() on while it forms z_t under no_grad
(stage2_train.py:166, 200-203), so the stage-I BERT's nn.Dropout sites (med.py:108, 225, 252, 330) are active.  train_med.py is that
forward on the fused training kernels; here it meets the CPU oracle's MED forward with EXACTLY the masks the kernels drew (regenerated on
the host from the counters: tests/helpers.pair_keep / splitmix_keep), plus the mode switch (`.eval()` = the inference engine, bit for bit)."""

import numpy as np
import pytest
import torch

from candidate_reranking_cir_amd import synthetic
from tests import helpers as H
from tests.test_model_gpu import build_models

pytestmark = pytest.mark.gpu
BF, HF = torch.bfloat16, torch.float16


def _setup(dtype):
    z, g, v, sd2, sd1 = H.tiny_setup()
    m2, m1 = build_models(g, v, int(z["seed"]), str(z["profile"]), dtype, torch.device("cuda"))
    caps = [synthetic.caption_text(300 + i, n) for i, n in enumerate((5, 9, 7, 9))]
    ids, mask = H.tokenize(caps)
    ids = ids.clone()
    ids[:, 0] = getattr(m1.tokenizer, "enc_token_id", 30523)
    gen = torch.Generator().manual_seed(17)
    n_tok = (v.image_size // v.patch_size) ** 2 + 1
    toks = torch.randn((len(caps), n_tok, g.encoder_width), generator=gen)
    return z, g, v, sd1, m1, ids, mask, toks


def _hooks(seed, q_n, l, n, d, heads, ph, pa):
    from candidate_reranking_cir_amd import train_med as M
    r = q_n * l
    kind_site = {"self_attn": M.SITE_SELF_ATTN, "self_out": M.SITE_SELF_OUT, "cross_attn": M.SITE_CROSS_ATTN, "cross_out": M.SITE_CROSS_OUT, "ffn_out": M.SITE_FFN_OUT}
    kept = []

    def drop(kind, layer, x):
        if kind == "emb":                                                       # cir_eltwise's generator: element index of the launch
            mk, p = H.splitmix_keep(M.site_seed(seed, 0, M.SITE_EMB), r * d, ph).view(q_n, l, d), ph
        elif kind.endswith("_out"):                                             # fused dropout + residual + LayerNorm: (row, column)
            mk, p = H.pair_keep(M.site_seed(seed, layer, kind_site[kind]), r, d, ph).view(q_n, l, d), ph
        else:                                                                   # fused attention: row = (group * H + head) * Lq + query, column = key
            lk = l if kind == "self_attn" else n
            mk, p = H.pair_keep(M.site_seed(seed, layer, kind_site[kind]), q_n * heads * l, lk, pa).view(q_n, heads, l, lk), pa
        assert mk.shape == x.shape, (kind, mk.shape, x.shape)
        kept.append(mk.float().mean().item())
        return x * mk.to(x.dtype) / (1.0 - p)

    return drop, kept


@pytest.mark.parametrize("dtype", [BF, HF], ids=["bf16", "fp16"])
def test_z_t_train_mode_against_oracle_with_the_same_masks(dtype):
    from oracle import cir_oracle as O
    z, g, v, sd1, m1, ids, mask, toks = _setup(dtype)
    assert g.hidden_dropout_prob == 0.1 and g.attention_probs_dropout_prob == 0.1
    m1.train()
    torch.manual_seed(5)
    out = m1.z_t(toks.cuda(), ids.cuda(), mask.cuda())
    fwd = m1._med_dropout[1]
    seed = fwd.last_seed
    q_n, l = ids.shape
    drop, kept = _hooks(seed, q_n, l, toks.shape[1], g.hidden_size, g.num_attention_heads, 0.1, 0.1)
    w1 = {k: t.float() for k, t in sd1.items()}
    t16 = toks.to(dtype).float()                                                # the tokens as the 16-bit forward saw them
    ref = O.med_forward(w1, ids, mask, t16, drop=drop)
    ref_eval = O.med_forward(w1, ids, mask, t16)
    valid = mask.bool()
    e = (out.last_hidden_state.cpu() - ref)[valid].abs().max().item()
    moved = (ref - ref_eval)[valid].abs().max().item()
    print(f"\n[z_t train mode, {dtype}] vs oracle with the same masks {e:.3e}; dropout moves z_t by up to {moved:.3e}; kept {np.mean(kept):.3f} over {len(kept)} sites")
    assert len(kept) == 1 + 5 * g.num_hidden_layers and abs(np.mean(kept) - 0.9) < 0.03
    assert moved > 0.3 and e < (4e-2 if dtype == BF else 6e-3)          # measured on MI355X: 1.7e-2 / 1.9e-3 where the dropout itself moves z_t by 4.6
    assert (out.last_hidden_state16.float() - out.last_hidden_state).abs().max().item() < (4e-2 if dtype == BF else 5e-3)
    m1.eval()


def test_mode_switch_and_draws():
    """.eval() -> the inference engine (same bits as before train()); .train() -> a new draw per call from torch's global generator,
    reproducible under torch.manual_seed; zero probabilities -> no train-mode path at all; a text32 model (fp32 text side) runs the
    16-bit training plan on a twin engine."""
    z, g, v, sd1, m1, ids, mask, toks = _setup(HF)
    t, i, m = toks.cuda(), ids.cuda(), mask.cuda()
    e0 = m1.z_t(t, i, m).last_hidden_state.clone()
    m1.train()
    torch.manual_seed(11)
    a = m1.z_t(t, i, m).last_hidden_state.clone()
    b = m1.z_t(t, i, m).last_hidden_state.clone()
    torch.manual_seed(11)
    a2 = m1.z_t(t, i, m).last_hidden_state.clone()
    assert torch.equal(a, a2) and not torch.equal(a, b)
    assert (a - e0).abs().max().item() > 0.3 and torch.isfinite(a).all()
    m1.eval()
    assert torch.equal(m1.z_t(t, i, m).last_hidden_state, e0)
    # fp32 text side (what the factories set for real weights): the train-mode forward packs a 16-bit twin of the encoder
    m1.set_precision("text32")
    e32 = m1.z_t(t, i, m).last_hidden_state.clone()
    m1.train()
    torch.manual_seed(11)
    c = m1.z_t(t, i, m).last_hidden_state
    assert m1._med_dropout[1].eng is not m1.engines()[0] and m1._med_dropout[1].eng.dtype == HF
    assert torch.isfinite(c).all() and (c - a).abs().max().item() < 2e-2          # same draw, same 16-bit arithmetic as the fp16 model's
    m1.eval()
    assert torch.equal(m1.z_t(t, i, m).last_hidden_state, e32)
    # probabilities zero: train() changes nothing
    g0 = type(g)(**{**g.__dict__, "hidden_dropout_prob": 0.0, "attention_probs_dropout_prob": 0.0})
    m1.bert_geometry = g0
    m1.train()
    assert m1._dropout_forward() is None and torch.equal(m1.z_t(t, i, m).last_hidden_state, e32)
    m1.eval()
