"""Model geometry for the stage-II re-ranking path.

Reads the same JSON keys as the reference's `configs/med_config.json` (hidden_size,
num_attention_heads, num_hidden_layers, intermediate_size, layer_norm_eps, vocab_size,
max_position_embeddings; reference: blip_stage2.py:46-47) and the ViT factory arguments of
`create_vit` (blip.py:194-209: base = 768/12/12, large = 1024/24/16, patch 16).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field, asdict

HEAD_DIM = 64  # every attention on this path is 64-wide per head (768/12, 1024/16)


@dataclass
class BertGeometry:
    hidden_size: int = 768
    num_attention_heads: int = 12
    num_hidden_layers: int = 12
    intermediate_size: int = 3072
    layer_norm_eps: float = 1e-12
    vocab_size: int = 30524
    max_position_embeddings: int = 512
    encoder_width: int = 768
    merge_mlp_from_layer: int = 6  # nlvr_encoder.py:286: mergeMLP for layer_num >= 6, mergeAvg below
    pad_token_id: int = 0
    hidden_dropout_prob: float = 0.1            # training mode only (train.py); eval-mode dropout is the identity
    attention_probs_dropout_prob: float = 0.1
    extra: dict = field(default_factory=dict)

    @classmethod
    def from_json_file(cls, path: str) -> "BertGeometry":
        with open(path) as fh:
            raw = json.load(fh)
        return cls.from_dict(raw)

    @classmethod
    def from_dict(cls, raw: dict) -> "BertGeometry":
        known = {k: raw[k] for k in cls.__dataclass_fields__ if k in raw and k != "extra"}
        g = cls(**known)
        g.extra = {k: v for k, v in raw.items() if k not in known}
        g.validate()
        return g

    def validate(self):
        if self.hidden_size != self.num_attention_heads * HEAD_DIM:
            raise ValueError(
                f"HIP attention kernels are built for head_dim {HEAD_DIM}: hidden_size {self.hidden_size} "
                f"needs {self.hidden_size // HEAD_DIM} heads, config says {self.num_attention_heads}")
        if self.hidden_size % 64 or self.intermediate_size % 64 or self.encoder_width % 64:
            raise ValueError("hidden/intermediate/encoder widths must be multiples of 64")

    def to_dict(self):
        d = asdict(self)
        d.pop("extra")
        return d


@dataclass
class VitGeometry:
    image_size: int = 224
    patch_size: int = 16
    width: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: int = 4
    layer_norm_eps: float = 1e-6  # vit.py:142
    in_chans: int = 3
    drop_path_rate: float = 0.1   # TRAINING mode only (train_vit.py): blip_stage2.py:37 builds the stage-II image encoder with 0.1; block i
                                  # drops each sample's branch with probability linspace(0, rate, depth)[i] (vit.py:153, :98-109)

    @classmethod
    def named(cls, vit: str, image_size: int) -> "VitGeometry":
        # blip.py:194-209
        if vit == "base":
            return cls(image_size=image_size, width=768, depth=12, num_heads=12)
        if vit == "large":
            return cls(image_size=image_size, width=1024, depth=24, num_heads=16)
        raise AssertionError("vit parameter must be base or large")

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_tokens(self) -> int:
        return self.grid * self.grid + 1

    def validate(self):
        if self.width != self.num_heads * HEAD_DIM:
            raise ValueError(f"HIP attention kernels are built for head_dim {HEAD_DIM}")
        if self.image_size % self.patch_size:
            raise ValueError("image_size must be a multiple of patch_size")
