"""Import shim for the upstream reference (container-only; never runs on the GPU box).

TEST INFRASTRUCTURE - not product code.  This module makes the *unmodified* reference sources
under /root/reference/src importable in this image (transformers 5.x, no timm / fairscale /
torchvision), so that `oracle/make_golden.py` can run the real reference forward on CPU and
write golden vectors under tests/golden/.  Nothing from the reference is copied: the shim only
registers stand-in *third-party* modules (timm==0.4.12 PatchEmbed/DropPath/trunc_normal_,
fairscale checkpoint_wrapper) restated from their published definitions, and relocates a few
`transformers` helpers that moved between 4.25 and 5.x.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CIR_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src"))


class _Tok:
    """Stand-in for the HF tokenizer output object (`.input_ids`, `.attention_mask`, `.to`)."""

    def __init__(self, input_ids, attention_mask):
        self.input_ids = input_ids
        self.attention_mask = attention_mask

    def to(self, device):
        self.input_ids = self.input_ids.to(device)
        self.attention_mask = self.attention_mask.to(device)
        return self


class PresetTokenizer:
    """Tokenizer double: returns pre-set ids/mask (no WordPiece vocab is available offline).

    blip.py:186-191 builds `BertTokenizer.from_pretrained('bert-base-uncased')` (network) and
    sets `enc_token_id`; callers only use `tokenizer(text, padding='longest', return_tensors='pt')`
    and `.enc_token_id` (blip_stage2.py:113-114, blip_stage1.py:72-73).
    """

    enc_token_id = 30523

    def __init__(self):
        self.next_ids = None
        self.next_mask = None

    def preset(self, input_ids, attention_mask):
        self.next_ids, self.next_mask = input_ids, attention_mask

    def __call__(self, text, padding="longest", return_tensors="pt"):
        return _Tok(self.next_ids.clone(), self.next_mask.clone())


def install():
    """Register stand-in third-party modules and put the reference on sys.path."""
    import torch
    import torch.nn as nn
    import transformers  # noqa: F401  (must be imported before the fake timm is registered)
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    if "timm" not in sys.modules:
        class PatchEmbed(nn.Module):
            # timm==0.4.12 models/layers/patch_embed.py: Conv2d(k=p, s=p) -> flatten(2).transpose(1, 2)
            def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
                super().__init__()
                self.img_size = (img_size, img_size)
                self.patch_size = (patch_size, patch_size)
                self.grid_size = (img_size // patch_size, img_size // patch_size)
                self.num_patches = self.grid_size[0] * self.grid_size[1]
                self.flatten = flatten
                self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
                self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

            def forward(self, x):
                x = self.proj(x)
                if self.flatten:
                    x = x.flatten(2).transpose(1, 2)
                return self.norm(x)

        class DropPath(nn.Module):
            def __init__(self, drop_prob=None):
                super().__init__()
                self.drop_prob = drop_prob

            def forward(self, x):
                if not self.training or not self.drop_prob:
                    return x
                keep = 1 - self.drop_prob
                shape = (x.shape[0],) + (1,) * (x.ndim - 1)
                mask = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
                mask.floor_()
                return x.div(keep) * mask

        def _mod(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            m.__spec__ = None
            sys.modules[name] = m
            return m

        noop = lambda *a, **k: None
        _mod("timm")
        _mod("timm.models")
        _mod("timm.models.vision_transformer", _cfg=lambda **k: dict(k), PatchEmbed=PatchEmbed)
        _mod("timm.models.registry", register_model=lambda f: f)
        _mod("timm.models.layers", trunc_normal_=nn.init.trunc_normal_, DropPath=DropPath)
        _mod("timm.models.helpers", named_apply=noop, adapt_input_conv=noop)
        _mod("timm.models.hub", download_cached_file=noop)
        _mod("fairscale")
        _mod("fairscale.nn")
        _mod("fairscale.nn.checkpoint")
        _mod("fairscale.nn.checkpoint.checkpoint_activations", checkpoint_wrapper=lambda m, **k: m)

    # helpers that moved / vanished between transformers 4.25 and 5.x
    for name in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = getattr(pu, "find_pruneable_heads_and_indices", None)
    PTM = mu.PreTrainedModel
    PTM.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    PTM.init_weights = lambda self: self.apply(self._init_weights)

    src = os.path.join(REFERENCE_ROOT, "src")
    if src not in sys.path:
        sys.path.insert(0, src)


def load_reference_modules():
    """Return the reference modules (vit, med, nlvr_encoder, blip_stage1, blip_stage2) with the
    tokenizer factory replaced by :class:`PresetTokenizer`."""
    install()
    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)  # the reference opens configs/ by relative path
    try:
        import vit as ref_vit
        import med as ref_med
        import nlvr_encoder as ref_nlvr
        import blip as ref_blip
        import blip_stage1 as ref_s1
        import blip_stage2 as ref_s2
    finally:
        os.chdir(cwd)
    ref_s1.init_tokenizer = PresetTokenizer
    ref_s2.init_tokenizer = PresetTokenizer
    ref_blip.init_tokenizer = PresetTokenizer
    return types.SimpleNamespace(vit=ref_vit, med=ref_med, nlvr=ref_nlvr, blip=ref_blip, s1=ref_s1, s2=ref_s2)
