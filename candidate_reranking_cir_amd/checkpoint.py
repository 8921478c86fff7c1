"""Loading BLIP base checkpoints the way the reference's factories do (host logic only, CPU tensors).

  * stage II - `blip_stage2.load_checkpoint` (blip_stage2.py:148-190): take `checkpoint['model']`, resize the ViT
    position embedding to the model's grid, then give BOTH branches of the two-branch BERT the single-branch weights:
    every `attention.self.*` / `crossattention.self.*` tensor is also stored under `self0` and `self1`, every
    `(cross)attention.output.dense.*` under `dense0` / `dense1`, every `(cross)attention.output.LayerNorm.*` under
    `LayerNormA` / `LayerNormB`; load non-strictly and report the missing keys.
  * stage I - `blip.load_checkpoint` (blip.py:215-237): resize the position embedding, drop tensors whose shape differs
    from the model's, load non-strictly.

URLs are downloaded by the reference (timm's `download_cached_file`); this path has no network and rejects anything that
is not a local file with the reference's own error message.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


def interpolate_pos_embed(pos_embed: torch.Tensor, num_patches: int, num_tokens: int) -> torch.Tensor:
    """vit.py:281-305: bicubic resize of the patch-position grid; class (extra) tokens are kept as they are.
    `num_patches` / `num_tokens`: the MODEL's patch count and position-embedding length."""
    dim = pos_embed.shape[-1]
    extra = num_tokens - num_patches
    old = int((pos_embed.shape[-2] - extra) ** 0.5)
    new = int(num_patches ** 0.5)
    if old == new:
        return pos_embed
    grid = pos_embed[:, extra:].reshape(-1, old, old, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(new, new), mode="bicubic", align_corners=False)
    print("reshape position embedding from %d to %d" % (old ** 2, new ** 2))
    return torch.cat((pos_embed[:, :extra], grid.permute(0, 2, 3, 1).flatten(1, 2)), dim=1)


def duplicate_for_two_branches(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """blip_stage2.py:159-186 (the reference rewrites with str.replace on the whole key; every key it touches holds the
    replaced word exactly once, so the explicit forms below give the same names)."""
    for key in list(state_dict.keys()):
        if "attention.self." in key:                                   # also matches 'crossattention.self.'
            state_dict[key.replace("self", "self0")] = state_dict[key]
            state_dict[key.replace("self", "self1")] = state_dict[key]
        elif "attention.output.dense." in key:
            state_dict[key.replace("dense", "dense0")] = state_dict[key]
            state_dict[key.replace("dense", "dense1")] = state_dict[key]
        if "output.LayerNorm" in key and "attention" in key:
            state_dict[key.replace("LayerNorm", "LayerNormA")] = state_dict[key]
            state_dict[key.replace("LayerNorm", "LayerNormB")] = state_dict[key]
    return state_dict


def _read(path: str) -> dict:
    if not os.path.isfile(path):
        raise RuntimeError("checkpoint url or path is invalid")        # blip_stage2.py:156 / blip.py:222 (no network here)
    return torch.load(path, map_location="cpu")


def _vit_grid(model) -> Tuple[int, int]:
    g = model.vit_geometry
    return g.num_tokens - 1, g.num_tokens


def load_stage2_checkpoint(model, path: str):
    """Returns (model, msg) like blip_stage2.load_checkpoint.  A file saved by the reference's training scripts
    ({'BLIP_NLVR': state_dict}, utils.py:145-150) or a bare state dict is loaded as it is."""
    ckpt = _read(path)
    if "model" in ckpt:
        sd = ckpt["model"]
        sd["visual_encoder.pos_embed"] = interpolate_pos_embed(sd["visual_encoder.pos_embed"], *_vit_grid(model))
        sd = duplicate_for_two_branches(sd)
    else:
        sd = ckpt.get("BLIP_NLVR", ckpt)
    msg = model.load_state_dict(sd, strict=False)
    print("load checkpoint from %s" % path)
    return model, msg


def load_stage1_checkpoint(model, path: str):
    """Returns (model, msg) like blip.load_checkpoint (momentum encoders are not part of this model)."""
    ckpt = _read(path)
    if "model" in ckpt:
        sd = ckpt["model"]
        sd["visual_encoder.pos_embed"] = interpolate_pos_embed(sd["visual_encoder.pos_embed"], *_vit_grid(model))
        own = model.state_dict()
        for key in list(own.keys()):
            if key in sd and sd[key].shape != own[key].shape:
                del sd[key]
    else:
        sd = ckpt.get("BLIP_Retrieval", ckpt)
    msg = model.load_state_dict(sd, strict=False)
    print("load checkpoint from %s" % path)
    return model, msg
