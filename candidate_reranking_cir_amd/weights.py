"""State-dict layout of the two models on the path, and a deterministic weight synthesiser.

Layout contract: the checkpoint entries `["BLIP_NLVR"]` and
`["BLIP_Retrieval"]` written by the reference's `save_model` (utils.py:135-150) and read back
with `load_state_dict` (validate_stage2.py:347-348, 359-360).  `nlvr_param_spec` /
`retrieval_param_spec` enumerate exactly those keys and shapes.

There are no checkpoints or datasets offline, so tests and the benchmark use weights synthesised
from the *key name* (`synth_state_dict`): the same generator runs in the build container (to load
the real reference for golden vectors) and on the GPU box, so no reference file has to travel.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import BertGeometry, VitGeometry

Spec = "OrderedDict[str, Tuple[Tuple[int, ...], str]]"  # key -> (shape, kind)


def _vit_spec(v: VitGeometry, prefix="visual_encoder.") -> OrderedDict:
    d, f = v.width, v.width * v.mlp_ratio
    s = OrderedDict()
    s[prefix + "cls_token"] = ((1, 1, d), "embed")
    s[prefix + "pos_embed"] = ((1, v.num_tokens, d), "embed")
    s[prefix + "patch_embed.proj.weight"] = ((d, v.in_chans, v.patch_size, v.patch_size), "weight")
    s[prefix + "patch_embed.proj.bias"] = ((d,), "bias")
    for i in range(v.depth):
        p = f"{prefix}blocks.{i}."
        s[p + "norm1.weight"] = ((d,), "ln_w")
        s[p + "norm1.bias"] = ((d,), "ln_b")
        s[p + "attn.qkv.weight"] = ((3 * d, d), "weight")
        s[p + "attn.qkv.bias"] = ((3 * d,), "bias")
        s[p + "attn.proj.weight"] = ((d, d), "weight")
        s[p + "attn.proj.bias"] = ((d,), "bias")
        s[p + "norm2.weight"] = ((d,), "ln_w")
        s[p + "norm2.bias"] = ((d,), "ln_b")
        s[p + "mlp.fc1.weight"] = ((f, d), "weight")
        s[p + "mlp.fc1.bias"] = ((f,), "bias")
        s[p + "mlp.fc2.weight"] = ((d, f), "weight")
        s[p + "mlp.fc2.bias"] = ((d,), "bias")
    s[prefix + "norm.weight"] = ((d,), "ln_w")
    s[prefix + "norm.bias"] = ((d,), "ln_b")
    return s


def _linear(s, key, out_f, in_f):
    s[key + ".weight"] = ((out_f, in_f), "weight")
    s[key + ".bias"] = ((out_f,), "bias")


def _ln(s, key, d):
    s[key + ".weight"] = ((d,), "ln_w")
    s[key + ".bias"] = ((d,), "ln_b")


def _embeddings_spec(s, g: BertGeometry, prefix):
    s[prefix + "embeddings.position_ids"] = ((1, g.max_position_embeddings), "position_ids")
    s[prefix + "embeddings.word_embeddings.weight"] = ((g.vocab_size, g.hidden_size), "weight")
    s[prefix + "embeddings.position_embeddings.weight"] = ((g.max_position_embeddings, g.hidden_size), "weight")
    _ln(s, prefix + "embeddings.LayerNorm", g.hidden_size)


def nlvr_param_spec(g: BertGeometry, v: VitGeometry) -> OrderedDict:
    """Keys/shapes of `BLIP_NLVR.state_dict()` (blip_stage2.py:19-54; nlvr_encoder.py:94-397)."""
    h, f, ew = g.hidden_size, g.intermediate_size, g.encoder_width
    s = _vit_spec(v)
    _embeddings_spec(s, g, "text_encoder.")
    for i in range(g.num_hidden_layers):
        p = f"text_encoder.encoder.layer.{i}."
        for blk, kv_in in (("attention", h), ("crossattention", ew)):
            for b in (0, 1):
                _linear(s, f"{p}{blk}.self{b}.query", h, h)
                _linear(s, f"{p}{blk}.self{b}.key", h, kv_in)
                _linear(s, f"{p}{blk}.self{b}.value", h, kv_in)
            _ln(s, f"{p}{blk}.output.LayerNormA", h)
            _ln(s, f"{p}{blk}.output.LayerNormB", h)
            _linear(s, f"{p}{blk}.output.dense0", h, h)
            _linear(s, f"{p}{blk}.output.dense1", h, h)
            if blk == "crossattention" and i >= g.merge_mlp_from_layer:
                _linear(s, f"{p}{blk}.output.merge_layer", h, 2 * h)
        _linear(s, p + "intermediate.dense", f, h)
        _linear(s, p + "output.dense", h, f)
        _ln(s, p + "output.LayerNorm", h)
    _linear(s, "cls_head.0", h, 2 * h)
    _linear(s, "cls_head.2", 2, h)
    return s


def retrieval_param_spec(g: BertGeometry, v: VitGeometry, embed_dim: int = 256) -> OrderedDict:
    """Keys/shapes of `BLIP_Retrieval.state_dict()` (blip_stage1.py:15-45; med.py:112-346)."""
    h, f, ew = g.hidden_size, g.intermediate_size, g.encoder_width
    s = _vit_spec(v)
    _embeddings_spec(s, g, "text_encoder.")
    for i in range(g.num_hidden_layers):
        p = f"text_encoder.encoder.layer.{i}."
        for blk, kv_in in (("attention", h), ("crossattention", ew)):
            _linear(s, f"{p}{blk}.self.query", h, h)
            _linear(s, f"{p}{blk}.self.key", h, kv_in)
            _linear(s, f"{p}{blk}.self.value", h, kv_in)
            _linear(s, f"{p}{blk}.output.dense", h, h)
            _ln(s, f"{p}{blk}.output.LayerNorm", h)
        _linear(s, p + "intermediate.dense", f, h)
        _linear(s, p + "output.dense", h, f)
        _ln(s, p + "output.LayerNorm", h)
    _linear(s, "vision_proj", embed_dim, v.width)
    _linear(s, "text_proj", embed_dim, h)
    s["temp"] = ((), "temp")
    return s


# profile -> (weight std, bias std, LN gamma jitter, LN beta std, embed std)
_PROFILES = {
    # the reference's own init: N(0, .02) weights, zero biases, unit LN (nlvr_encoder.py:663-673, vit.py:167-174)
    "init": (0.02, 0.0, 0.0, 0.0, 0.02),
    # every tensor non-trivial so that a dropped bias / swapped gamma shows up in parity tests
    "test": (0.02, 0.02, 0.1, 0.05, 0.02),
    # larger weights: activations and logits spread out (rank-order tests, SURVEY section 7 hard part 1)
    "spread": (0.05, 0.05, 0.1, 0.05, 0.05),
    # "test" plus OUTLIER CHANNELS (what pretrained ViT / BERT checkpoints carry and N(0, .02) weights do not): in three
    # channels the ViT's fc2 rows and biases are 40x / +-25, so the pre-LN residual stream reaches 1e2..1e3 there while
    # the matching LayerNorm gammas are small; the BERT LayerNorms give three channels a 6x gain and a +-4 offset and
    # their output projections 20x rows.  Pins the fp16 residual stream on a stream that is NOT O(1) (tests/golden/outlier224.npz).
    "outlier": (0.02, 0.02, 0.1, 0.05, 0.02),
}
_OUTLIER_CHANNELS = (17, 300, 555)


def _apply_outliers(key: str, t: torch.Tensor, kind: str) -> torch.Tensor:
    if t.dim() == 0 or kind in ("position_ids", "temp", "embed"):
        return t
    ch = [c % t.shape[0] for c in _OUTLIER_CHANNELS]
    sign = torch.tensor([1.0, -1.0, 1.0])
    vit = key.startswith("visual_encoder.")
    if vit and key.endswith("mlp.fc2.weight"):
        t[ch] *= 40.0
    elif vit and key.endswith("mlp.fc2.bias"):
        t[ch] += 25.0 * sign
    elif vit and kind == "ln_w":
        t[ch] *= 0.1
    elif not vit and key.endswith("output.dense.weight") or ".output.dense0.weight" in key or ".output.dense1.weight" in key:
        t[ch] *= 20.0
    elif not vit and kind == "ln_w" and "LayerNorm" in key:
        t[ch] *= 6.0
    elif not vit and kind == "ln_b" and "LayerNorm" in key:
        t[ch] += 4.0 * sign
    return t


def synth_tensor(key: str, shape, kind: str, seed: int = 0, profile: str = "test") -> torch.Tensor:
    """Deterministic tensor for state-dict entry `key` (fp32 on CPU; int64 for position_ids)."""
    w_std, b_std, g_jit, be_std, e_std = _PROFILES[profile]
    if kind == "position_ids":
        return torch.arange(shape[1], dtype=torch.int64).expand(shape).clone()
    if kind == "temp":
        return torch.tensor(0.07)
    gen = torch.Generator(device="cpu")
    gen.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    t = torch.randn(shape, generator=gen, dtype=torch.float32)
    if kind == "weight":
        t = t * w_std
    elif kind == "embed":
        t = t * e_std
    elif kind == "bias":
        t = t * b_std
    elif kind == "ln_w":
        t = 1.0 + t * g_jit
    elif kind == "ln_b":
        t = t * be_std
    else:
        raise KeyError(kind)
    return _apply_outliers(key, t, kind) if profile == "outlier" else t


def synth_state_dict(spec: OrderedDict, seed: int = 0, profile: str = "test", threads: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Every tensor has its own generator (seeded by its key), so the tensors can be drawn concurrently and the result does not
    depend on the thread count (torch.randn releases the GIL; a full-size pair of models is 0.5 G normals: ~25 s on one core)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    items = list(spec.items())
    n = threads or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    if n <= 1 or len(items) < 8:
        return OrderedDict((k, synth_tensor(k, shape, kind, seed, profile)) for k, (shape, kind) in items)
    saved = torch.get_num_threads()
    torch.set_num_threads(1)                      # one core per tensor instead of all cores on one tensor at a time
    try:
        with ThreadPoolExecutor(n) as pool:
            vals = list(pool.map(lambda kv: synth_tensor(kv[0], kv[1][0], kv[1][1], seed, profile), items))
    finally:
        torch.set_num_threads(saved)
    return OrderedDict((k, v) for (k, _), v in zip(items, vals))
