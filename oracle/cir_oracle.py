"""CPU oracle for the stage-II candidate re-ranking forward path (fp32, plain torch ops).

TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product package
(`candidate_reranking_cir_amd`) never does and fails loudly when its HIP library is missing.

This is an independent restatement of the reference's arithmetic as flat functions over a
`state_dict`-style weight table (no module tree).  Every function cites the reference
file:line it follows (paths relative to /root/reference/src).  It is pinned against golden
vectors produced by running the *real* reference in the build container
(`oracle/make_golden.py` -> `tests/golden/*.npz`, checked by `tests/test_oracle_golden.py`).

Third-party arithmetic that is not under /root/reference:
  * timm==0.4.12 `PatchEmbed` (Conv2d k=16 s=16 -> flatten(2).transpose(1,2)) - restated from the
    published definition; pinned through the goldens (the generator's shim uses the same
    published definition, the reference's own tests hold nothing for it).
  * transformers==4.25.0 `invert_attention_mask` / `ACT2FN['gelu']` - exact-erf GELU,
    additive masks `(1-m)*-10000` (self) and `(1-m)*finfo.min` (encoder side).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Weights = Dict[str, torch.Tensor]
SKIP_FILL = -99999.99  # validate_stage2.py:123, 258


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
def _lin(w: Weights, key: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, w[key + ".weight"], w[key + ".bias"])


def _ln(w: Weights, key: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), w[key + ".weight"], w[key + ".bias"], eps)


def _heads(x: torch.Tensor, n_heads: int) -> torch.Tensor:
    b, t, d = x.shape
    return x.view(b, t, n_heads, d // n_heads).transpose(1, 2)  # (b, h, t, dh)


def _sdpa(q, k, v, add_mask: Optional[torch.Tensor], n_heads: int, pdrop=None) -> torch.Tensor:
    """softmax(q k^T / sqrt(dh) + mask) v with heads merged back.

    nlvr_encoder.py:175 (scores), :193 (scale), :194-196 (mask), :199 (softmax), :213-217
    (context + head merge); identical text in med.py:193-235.  `pdrop` (optional callable): the dropout on the attention probabilities
    (nlvr_encoder.py:207) with a GIVEN mask - see `nlvr_forward`'s `drop`.
    """
    qh, kh, vh = _heads(q, n_heads), _heads(k, n_heads), _heads(v, n_heads)
    s = qh @ kh.transpose(-1, -2)
    s = s / math.sqrt(qh.shape[-1])
    if add_mask is not None:
        s = s + add_mask
    p = torch.softmax(s, dim=-1)
    if pdrop is not None:
        p = pdrop(p)
    ctx = p @ vh
    b, h, t, dh = ctx.shape
    return ctx.transpose(1, 2).reshape(b, t, h * dh)


def self_mask_additive(attention_mask: torch.Tensor) -> torch.Tensor:
    """(B, L) ones/zeros -> (B,1,1,L) additive mask, nlvr_encoder.py:746,773-774 / med.py:683."""
    return (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0


def encoder_mask_additive(attention_mask: torch.Tensor) -> torch.Tensor:
    """transformers `invert_attention_mask` (called nlvr_encoder.py:863-868, med.py:766-772)."""
    m = attention_mask[:, None, None, :].to(torch.float32)
    return (1.0 - m) * torch.finfo(torch.float32).min


# ----------------------------------------------------------------------------------------------
# ViT-B/16 (vit.py) + timm PatchEmbed
# ----------------------------------------------------------------------------------------------
def vit_forward(w: Weights, image: torch.Tensor, prefix: str = "visual_encoder.", n_heads: Optional[int] = None,
                patch: int = 16, eps: float = 1e-6, branch_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """VisionTransformer.forward, vit.py:180-194 (eval mode: dropout / DropPath are identity).  `branch_scale` ((depth, 2, B), optional):
    DropPath with GIVEN per-sample factors (0 = dropped, 1 / keep = kept; vit.py:98-109 with timm's DropPath) on the attention [.., 0, ..]
    and MLP [.., 1, ..] branch of every block - the train-mode function for one fixed draw, differentiable like the rest."""
    pw = w[prefix + "patch_embed.proj.weight"]
    d = pw.shape[0]
    if n_heads is None:
        n_heads = d // 64
    depth = 1 + max(int(k[len(prefix) + 7:].split(".")[0]) for k in w if k.startswith(prefix + "blocks."))
    # timm PatchEmbed: conv stride=kernel -> (B, D, gh, gw) -> flatten(2).transpose(1, 2)
    x = F.conv2d(image, pw, w[prefix + "patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2)
    b = x.shape[0]
    x = torch.cat([w[prefix + "cls_token"].expand(b, -1, -1), x], dim=1)       # vit.py:184-185
    x = x + w[prefix + "pos_embed"][:, : x.shape[1], :]                        # vit.py:187
    for i in range(depth):
        p = f"{prefix}blocks.{i}."
        y = _ln(w, p + "norm1", x, eps)                                        # vit.py:108
        qkv = _lin(w, p + "attn.qkv", y)                                       # vit.py:72
        n = qkv.shape[1]
        qkv = qkv.reshape(b, n, 3, n_heads, d // n_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)) * ((d // n_heads) ** -0.5)               # vit.py:75
        a = a.softmax(dim=-1)                                                  # vit.py:76
        y = (a @ v).transpose(1, 2).reshape(b, n, d)                           # vit.py:83
        y = _lin(w, p + "attn.proj", y)                                        # vit.py:84
        x = x + (y if branch_scale is None else y * branch_scale[i, 0].view(b, 1, 1))      # vit.py:108
        y = _ln(w, p + "norm2", x, eps)
        y = _lin(w, p + "mlp.fc2", F.gelu(_lin(w, p + "mlp.fc1", y)))          # vit.py:35-41
        x = x + (y if branch_scale is None else y * branch_scale[i, 1].view(b, 1, 1))      # vit.py:109
    return _ln(w, prefix + "norm", x, eps)                                     # vit.py:192


# ----------------------------------------------------------------------------------------------
# BERT embeddings (shared by med.py and nlvr_encoder.py)
# ----------------------------------------------------------------------------------------------
def bert_embeddings(w: Weights, input_ids: torch.Tensor, prefix: str = "text_encoder.", eps: float = 1e-12):
    """BertEmbeddings.forward, nlvr_encoder.py:68-91 / med.py:87-110 (absolute positions)."""
    e = w[prefix + "embeddings.word_embeddings.weight"][input_ids]
    e = e + w[prefix + "embeddings.position_embeddings.weight"][: input_ids.shape[1]][None]
    return _ln(w, prefix + "embeddings.LayerNorm", e, eps)


def _n_layers(w: Weights, prefix: str) -> int:
    tag = prefix + "encoder.layer."
    return 1 + max(int(k[len(tag):].split(".")[0]) for k in w if k.startswith(tag))


# ----------------------------------------------------------------------------------------------
# stage-I BERT/MED encoder (med.py) -> z_t
# ----------------------------------------------------------------------------------------------
def med_forward(w: Weights, input_ids: torch.Tensor, attention_mask: torch.Tensor, enc: torch.Tensor,
                enc_mask: Optional[torch.Tensor] = None, prefix: str = "text_encoder.",
                n_heads: Optional[int] = None, eps: float = 1e-12, drop=None) -> torch.Tensor:
    """med.BertModel.forward in 'multimodal' mode, med.py:685-821; per layer med.py:348-398:
    self-attn -> cross-attn(image tokens) -> FFN, each followed by residual + LayerNorm.
    `drop(kind, layer, x)` (optional): the module's train-mode nn.Dropout sites with GIVEN masks - 'emb' (med.py:108), 'self_attn' /
    'cross_attn' (the attention probabilities, med.py:225), 'self_out' / 'cross_out' (med.py:252), 'ffn_out' (med.py:330) - what the
    reference's loop leaves on while it forms z_t (stage2_train.py:166, 200-203)."""
    dr = (lambda kind, layer, x: x) if drop is None else drop
    pd = (lambda kind, layer: None) if drop is None else (lambda kind, layer: (lambda p: drop(kind, layer, p)))
    h = dr("emb", 0, bert_embeddings(w, input_ids, prefix, eps))
    d = h.shape[-1]
    n_heads = n_heads or d // 64
    smask = self_mask_additive(attention_mask)
    if enc_mask is None:
        enc_mask = torch.ones(enc.shape[:2], dtype=torch.long)
    emask = encoder_mask_additive(enc_mask)
    for i in range(_n_layers(w, prefix)):
        p = f"{prefix}encoder.layer.{i}."
        a = p + "attention.self."
        ctx = _sdpa(_lin(w, a + "query", h), _lin(w, a + "key", h), _lin(w, a + "value", h), smask, n_heads, pdrop=pd("self_attn", i))
        h = _ln(w, p + "attention.output.LayerNorm", dr("self_out", i, _lin(w, p + "attention.output.dense", ctx)) + h, eps)  # med.py:250-253
        c = p + "crossattention.self."
        ctx = _sdpa(_lin(w, c + "query", h), _lin(w, c + "key", enc), _lin(w, c + "value", enc), emask, n_heads, pdrop=pd("cross_attn", i))
        h = _ln(w, p + "crossattention.output.LayerNorm", dr("cross_out", i, _lin(w, p + "crossattention.output.dense", ctx)) + h, eps)
        f = F.gelu(_lin(w, p + "intermediate.dense", h))                       # med.py:319-322
        h = _ln(w, p + "output.LayerNorm", dr("ffn_out", i, _lin(w, p + "output.dense", f)) + h, eps)  # med.py:329-335
    return h


def stage1_z_t(w: Weights, ref_tokens: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor,
               enc_token_id: int = 30523, drop=None) -> torch.Tensor:
    """BLIP_Retrieval.img_txt_fusion(..., train=False, return_raw=True).last_hidden_state,
    blip_stage1.py:67-86: ids[:,0] <- [ENC]; med encoder with the reference-image tokens."""
    ids = input_ids.clone()
    ids[:, 0] = enc_token_id                                                   # blip_stage1.py:73
    return med_forward(w, ids, attention_mask, ref_tokens, drop=drop)


# ----------------------------------------------------------------------------------------------
# stage-II two-branch BERT (nlvr_encoder.py)
# ----------------------------------------------------------------------------------------------
def nlvr_forward(w: Weights, input_ids: torch.Tensor, attention_mask: torch.Tensor, z_t: torch.Tensor,
                 cand: torch.Tensor, cand_mask: Optional[torch.Tensor] = None, prefix: str = "text_encoder.",
                 n_heads: Optional[int] = None, eps: float = 1e-12, taps: Optional[list] = None, drop=None) -> torch.Tensor:
    """nlvr_encoder.BertModel.forward, nlvr_encoder.py:777-908 -> (K, 2*D).

    Branch 0 starts from z_t, branch 1 from the caption embeddings (:891-892).  Per layer
    (:414-476): twin self-attention with separate weights and LayerNormA/B (:262-264); twin
    cross-attention to the same candidate tokens, merged by average for layers < 6 (:257-260)
    or by `merge_layer` on the concatenation for layers >= 6 (:252-256, no activation); the
    merged tensor is added to each branch's residual and normalised by LayerNormA / LayerNormB;
    the FFN uses shared weights on both branches (:469-476).

    `drop` (optional callable (kind, layer, branch, tensor) -> tensor): the train-mode dropouts with GIVEN masks, at the reference's sites -
    "emb" (BertEmbeddings, :90), "self_attn" / "cross_attn" (attention probabilities, :207), "self_out" (BertSelfOutput after the dense,
    :250-264), "cross_out" (the same module behind the merge / average: ONE mask, branch index 2), "ffn_out" (BertOutput, :397-409).  With it
    this is the training-mode function for one fixed draw - tests hand it the masks the HIP kernels regenerate from their counters.
    """
    D = (lambda kind, layer, b, x: x) if drop is None else drop
    emb = D("emb", -1, 1, bert_embeddings(w, input_ids, prefix, eps))
    assert z_t.shape == emb.shape, "left and right inputs shall be the same shape"  # :891
    d = emb.shape[-1]
    n_heads = n_heads or d // 64
    smask = self_mask_additive(attention_mask)
    if cand_mask is None:
        cand_mask = torch.ones(cand.shape[:2], dtype=torch.long)
    emask = encoder_mask_additive(cand_mask)
    h = [z_t, emb]
    for i in range(_n_layers(w, prefix)):
        p = f"{prefix}encoder.layer.{i}."
        att = []
        for b in (0, 1):
            s = f"{p}attention.self{b}."
            ctx = _sdpa(_lin(w, s + "query", h[b]), _lin(w, s + "key", h[b]), _lin(w, s + "value", h[b]), smask, n_heads,
                        pdrop=None if drop is None else (lambda pr, i=i, b=b: D("self_attn", i, b, pr)))
            o = D("self_out", i, b, _lin(w, f"{p}attention.output.dense{b}", ctx)) + h[b]
            att.append(_ln(w, p + "attention.output.LayerNorm" + "AB"[b], o, eps))
        dd = []
        for b in (0, 1):
            s = f"{p}crossattention.self{b}."
            ctx = _sdpa(_lin(w, s + "query", att[b]), _lin(w, s + "key", cand), _lin(w, s + "value", cand), emask, n_heads,
                        pdrop=None if drop is None else (lambda pr, i=i, b=b: D("cross_attn", i, b, pr)))
            dd.append(_lin(w, f"{p}crossattention.output.dense{b}", ctx))
        mkey = p + "crossattention.output.merge_layer"
        if mkey + ".weight" in w:                                              # layers >= 6
            m = _lin(w, mkey, torch.cat(dd, dim=-1))
        else:                                                                  # layers < 6
            m = (dd[0] + dd[1]) / 2
        m = D("cross_out", i, 2, m)
        x = [_ln(w, p + "crossattention.output.LayerNorm" + "AB"[b], m + att[b], eps) for b in (0, 1)]
        for b in (0, 1):
            f = F.gelu(_lin(w, p + "intermediate.dense", x[b]))
            h[b] = _ln(w, p + "output.LayerNorm", D("ffn_out", i, b, _lin(w, p + "output.dense", f)) + x[b], eps)
        if taps is not None:
            taps.append((h[0][:, 0, :8].clone(), h[1][:, 0, :8].clone()))
    return torch.cat([h[0][:, 0, :], h[1][:, 0, :]], dim=-1)                   # :906-908


def img_txt_fusion_val(w: Weights, z_t: torch.Tensor, cand: torch.Tensor, input_ids: torch.Tensor,
                       attention_mask: torch.Tensor, enc_token_id: int = 30523, taps: Optional[list] = None):
    """BLIP_NLVR.img_txt_fusion_val, blip_stage2.py:101-136: expand z_t / ids to the K
    candidates, two-branch encoder, cls_head = Linear -> ReLU -> Linear, return column 0."""
    assert z_t.shape[0] == 1                                                   # blip_stage2.py:108
    k = cand.shape[0]
    ids = input_ids.clone()
    ids[:, 0] = enc_token_id                                                   # blip_stage2.py:114
    hid = nlvr_forward(w, ids.expand(k, -1), attention_mask.expand(k, -1), z_t.expand(k, -1, -1), cand, taps=taps)
    y = F.relu(F.linear(hid, w["cls_head.0.weight"], w["cls_head.0.bias"]))
    y = F.linear(y, w["cls_head.2.weight"], w["cls_head.2.bias"])
    return y[:, 0]


def img_txt_fusion_train(w: Weights, z_t: torch.Tensor, feats: torch.Tensor, input_ids: torch.Tensor,
                          attention_mask: torch.Tensor, enc_token_id: int = 30523, relu_mask: Optional[torch.Tensor] = None,
                          drop=None) -> torch.Tensor:
    """BLIP_NLVR.img_txt_fusion, blip_stage2.py:65-99 -> (B, B): row i runs caption i / z_t[i] (expanded to B rows, :83-87)
    against all B target images, rows stacked (:94), cls_head, column 0 (:96-99).  Dropout-free (p = 0 or eval mode);
    differentiable: with `requires_grad` weights, torch autograd over this function is the gradient oracle of the training
    step (stage2_train.py:210-216), pinned by tests/golden/train768.npz.

    `relu_mask` ((B*B, hidden) bool, optional) replaces cls_head's ReLU by a multiplication with that mask: the same function on
    the linear piece the mask selects.  The gradient of the head is discontinuous where a pre-activation crosses zero, so a
    16-bit forward whose pre-activations differ by 1e-3 takes a different piece for a handful of the B*B*hidden entries (each
    flip moves that row's whole gradient by ~1/sqrt(hidden/2)); tests that check the BACKWARD arithmetic pass the mask of the
    forward under test, tests that check the forward leave it out.  `drop` (optional callable (kind, layer, branch, query row i, tensor)):
    the train-mode dropouts with given masks (`nlvr_forward`), called once per site and query row - each call sees the B candidates of row i."""
    b = z_t.shape[0]
    ids = input_ids.clone()
    ids[:, 0] = enc_token_id                                                   # blip_stage2.py:71
    rows = []
    for i in range(b):
        hid = nlvr_forward(w, ids[i:i + 1].expand(b, -1), attention_mask[i:i + 1].expand(b, -1), z_t[i:i + 1].expand(b, -1, -1), feats,
                           drop=None if drop is None else (lambda kind, layer, br, x, i=i: drop(kind, layer, br, i, x)))
        y = F.linear(hid, w["cls_head.0.weight"], w["cls_head.0.bias"])
        y = F.relu(y) if relu_mask is None else y * relu_mask[i * b:(i + 1) * b].to(y.dtype)
        rows.append(F.linear(y, w["cls_head.2.weight"], w["cls_head.2.bias"])[:, 0])
    return torch.stack(rows)


def img_embed(w: Weights, image: torch.Tensor) -> torch.Tensor:
    """BLIP_NLVR.img_embed, blip_stage2.py:57-63."""
    return vit_forward(w, image)


# ----------------------------------------------------------------------------------------------
# scoring loop + metrics (validate_stage2.py)
# ----------------------------------------------------------------------------------------------
def score_queries(w2: Weights, w1: Weights, index_features: torch.Tensor, ref_index: Sequence[int],
                  cand_index: np.ndarray, labels: np.ndarray, input_ids: torch.Tensor,
                  attention_mask: torch.Tensor, group_index: Optional[np.ndarray] = None):
    """generate_fiq_val_predictions / generate_cirr_val_predictions, validate_stage2.py:69-129,
    209-278, with names replaced by integer rows of `index_features`.

    Per query q: if the K_labels row holds a positive (:95, :239) compute z_t from the reference
    image tokens and the caption (:106, :244), gather the K candidate token tensors in top-K
    order (:115, :251) and score them; otherwise emit a row of -99999.99 (:123, :258).  CIRR
    additionally scores the 5 non-reference group members for every query (:261-269).
    Stage-I weights use the `text_encoder.` keys of the BLIP_Retrieval state dict.
    """
    q_n, k_n = cand_index.shape
    out = torch.empty((q_n, k_n), dtype=torch.float32)
    gout = torch.empty((q_n, group_index.shape[1]), dtype=torch.float32) if group_index is not None else None
    for q in range(q_n):
        ids, msk = input_ids[q:q + 1], attention_mask[q:q + 1]
        z_t = None
        if labels[q].any():
            z_t = stage1_z_t(w1, index_features[ref_index[q]][None], ids, msk)
            out[q] = img_txt_fusion_val(w2, z_t, index_features[torch.as_tensor(cand_index[q])], ids, msk)
        else:
            out[q] = SKIP_FILL
        if group_index is not None:
            if z_t is None:
                z_t = stage1_z_t(w1, index_features[ref_index[q]][None], ids, msk)
            gout[q] = img_txt_fusion_val(w2, z_t, index_features[torch.as_tensor(group_index[q])], ids, msk)
    return (out, gout) if group_index is not None else out


def recall_at(logits: torch.Tensor, labels: np.ndarray, ks: Sequence[int]):
    """validate_stage2.py:53-62 / :174-200: argsort descending -> take_along_axis(K_labels) ->
    100 * sum(labels[:, :k]) / Q."""
    order = torch.argsort(logits, dim=-1, descending=True).cpu().numpy()
    lab = torch.tensor(np.take_along_axis(labels, order, axis=1))
    return [(torch.sum(lab[:, :k]) / len(lab)).item() * 100 for k in ks]


def group_recall_at(group_logits: torch.Tensor, group_members: np.ndarray, targets: np.ndarray, ks=(1, 2, 3)):
    """validate_stage2.py:187-203: argsort the 5 subset logits, map to member ids, compare with
    the target id, Recall_subset@k."""
    order = torch.argsort(group_logits, dim=-1, descending=True).cpu().numpy()
    names = np.take_along_axis(group_members, order, axis=1)
    lab = torch.tensor(names == np.repeat(np.asarray(targets), group_members.shape[1]).reshape(len(targets), -1))
    return [(torch.sum(lab[:, :k]) / len(lab)).item() * 100 for k in ks]


# ----------------------------------------------------------------------------------------------
# stage-I retrieval + top-K (validate.py) and CIRR test dicts (cirr_test_submission_stage2.py)
# ----------------------------------------------------------------------------------------------
def stage1_img_embed(w1: Weights, image: torch.Tensor):
    """BLIP_Retrieval.img_embed(..., return_pool_and_normalized=True), blip_stage1.py:48-65."""
    tokens = vit_forward(w1, image)
    pooled = F.normalize(F.linear(tokens[:, 0, :], w1["vision_proj.weight"], w1["vision_proj.bias"]), dim=-1)
    return tokens, pooled


def stage1_query_features(w1: Weights, ref_tokens: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor):
    """BLIP_Retrieval.img_txt_fusion(..., train=False), blip_stage1.py:67-88: normalised text_proj of z_t's CLS row."""
    z = stage1_z_t(w1, ref_tokens, input_ids, attention_mask)
    return F.normalize(F.linear(z[:, 0, :], w1["text_proj.weight"], w1["text_proj.bias"]), dim=-1)


def rank_index(predicted: torch.Tensor, index_pooled: torch.Tensor) -> np.ndarray:
    """validate.py:57-58 / 202-203: argsort of `1 - predicted @ index.T` (ascending)."""
    distances = 1 - predicted @ index_pooled.float().T
    return torch.argsort(distances, dim=-1).cpu().numpy()


def cirr_drop_reference(sorted_rows: np.ndarray, ref_index: np.ndarray) -> np.ndarray:
    """validate.py:206-210: remove the reference image from every ranking."""
    keep = sorted_rows != np.asarray(ref_index)[:, None]
    return sorted_rows[keep].reshape(sorted_rows.shape[0], sorted_rows.shape[1] - 1)


def cirr_test_dicts(logits: torch.Tensor, group_logits: torch.Tensor, cand_names: np.ndarray, group_names: np.ndarray, pair_ids):
    """cirr_test_submission_stage2.py:92-108: top-50 names by descending logit, top-3 subset names."""
    order = torch.argsort(logits, dim=-1, descending=True).cpu().numpy()
    names = np.take_along_axis(cand_names, order, axis=1)
    gorder = torch.argsort(group_logits, dim=-1, descending=True).cpu().numpy()
    gnames = np.take_along_axis(group_names, gorder, axis=1)
    rec = {str(int(p)): row[:50].tolist() for p, row in zip(pair_ids, names)}
    sub = {str(int(p)): row[:3].tolist() for p, row in zip(pair_ids, gnames)}
    return rec, sub
