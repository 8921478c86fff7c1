"""Stage-I model surface used by the stage-II scoring loop: `BLIP_Retrieval.img_txt_fusion(...,
train=False, return_raw=True)` produces z_t (reference: blip_stage1.py:67-92; called at
validate_stage2.py:106, 244, 264).  State-dict layout = the reference's `["BLIP_Retrieval"]` entry
(472 keys).  The stage-II scripts feed the text encoder image tokens from the stage-II ViT
(validate_stage2.py:139, 293); stage-I retrieval (validate.py) additionally
uses this model's own ViT and the 256-d `vision_proj` / `text_proj` heads (blip_stage1.py:48-65, 83).
"""
from __future__ import annotations

import os
from typing import Optional, Union

import torch

from . import ops
from .blip_stage2 import _EngineHost, default_tokenizer, encode_text, load_bert_geometry
from .config import BertGeometry, VitGeometry
from .engine import MedEngine, VitEngine
from .param_tree import populate
from .weights import retrieval_param_spec


class EncoderOutput:
    """What callers read from the HF output object: `.last_hidden_state` (blip_stage2.py:106)."""

    def __init__(self, last_hidden_state: torch.Tensor, last_hidden_state16: Optional[torch.Tensor] = None):
        self.last_hidden_state = last_hidden_state
        self.last_hidden_state16 = last_hidden_state16


class BLIP_Retrieval(_EngineHost):
    def __init__(self, med_config: Union[str, dict, BertGeometry] = "configs/med_config.json", image_size: int = 384,
                 vit: str = "base", vit_grad_ckpt: bool = False, vit_ckpt_layer: int = 0, embed_dim: int = 256, *,
                 vit_geometry: Optional[VitGeometry] = None, tokenizer=None):
        super().__init__()
        self.vit_geometry = vit_geometry or VitGeometry.named(vit, image_size)
        self.bert_geometry = load_bert_geometry(med_config)
        self.bert_geometry.encoder_width = self.vit_geometry.width
        self.tokenizer = tokenizer if tokenizer is not None else default_tokenizer()
        populate(self, retrieval_param_spec(self.bert_geometry, self.vit_geometry, embed_dim))
        self.text_encoder.config = self.bert_geometry

    def engines(self):
        if self._engines is None:
            dev = self.device
            if dev.type != "cuda":
                raise RuntimeError("BLIP_Retrieval runs on an MI355X only: move the model to 'cuda' (no CPU path)")
            sd = self.state_dict()
            f32 = lambda k: sd[k].detach().to(device=dev, dtype=torch.float32).contiguous()
            self._engines = (MedEngine(sd, self.bert_geometry, self.compute_dtype, dev, stream_dtype=self.stream_dtype, cross_dtype=self.token_dtype,
                                       split3=self.text_split3 if self.precision == "text32" else 0),
                             VitEngine(sd, self.vit_geometry, self.token_dtype, dev, stream_dtype=self.vit_stream_dtype),
                             dict(vw=f32("vision_proj.weight"), vb=f32("vision_proj.bias"), tw=f32("text_proj.weight"), tb=f32("text_proj.bias")))
        return self._engines

    @torch.no_grad()
    def img_embed(self, image, atts=False, return_pool_and_normalized=False):
        """blip_stage1.py:48-65: ViT tokens (B, N, D) fp32 [, normalised 256-d pooled features][, ones mask]."""
        _, vit, heads = self.engines()
        y32, _ = vit.forward(image.to(self.device), want32=True)
        out = (y32,)
        if return_pool_and_normalized:
            out += (ops.l2_normalize(ops.linear_f32(y32[:, 0, :], heads["vw"], heads["vb"])),)
        if atts:
            out += (torch.ones(y32.shape[:-1], dtype=torch.long, device=y32.device),)
        return out[0] if len(out) == 1 else out

    @torch.no_grad()
    def z_t(self, ref_tokens: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> EncoderOutput:
        """Batched z_t: reference-image tokens (Q, N, D), ids/mask (Q, L) with [ENC] set."""
        t = ref_tokens.to(self.device)
        fwd = self._dropout_forward() if self.training else None
        if fwd is not None:
            # model_stage1.train() (stage2_train.py:166): the reference forms z_t with every nn.Dropout of the stage-I BERT active, under
            # no_grad (stage2_train.py:200-203) - train_med.py; .eval() (every validation script) takes the inference engine below
            dt = fwd.eng.dtype
            h32, h16 = fwd.forward(input_ids, attention_mask, t if t.dtype == dt else ops.gather_rows(t, None, dt))
            return EncoderOutput(h32, h16)
        if t.dtype != self.token_dtype:
            t = ops.gather_rows(t, None, self.token_dtype)
        h32, h16 = self.engines()[0].forward(input_ids, attention_mask, t)
        return EncoderOutput(h32, h16)

    def _dropout_forward(self):
        """The train-mode forward of the text encoder (train_med.MedDropoutForward), or None when the med_config probabilities are zero.
        Operands as the training plan has them (train.train_dtype): the model's 16-bit compute type, fp16 for the fp32 modes - the inference
        engine's packed weights are shared when they are of that type, otherwise a 16-bit twin is packed once per engine build."""
        geo = self.bert_geometry
        ph, pa = float(geo.hidden_dropout_prob), float(geo.attention_probs_dropout_prob)
        if ph <= 0.0 and pa <= 0.0:
            return None
        from .train_med import MedDropoutForward
        eng = self.engines()[0]
        cur = getattr(self, "_med_dropout", None)
        if cur is None or cur[0] is not self._engines or (cur[1].p_hidden, cur[1].p_attn) != (ph, pa):
            dt = self.compute_dtype if self.compute_dtype in (torch.float16, torch.bfloat16) else torch.float16
            if not (eng.dtype == dt and eng.xdtype == dt):
                eng = MedEngine(self.state_dict(), geo, dt, self.device, stream_dtype=torch.float32, cross_dtype=dt)
            cur = self._med_dropout = (self._engines, MedDropoutForward(eng, ph, pa))
        return cur[1]

    @torch.no_grad()
    def img_txt_fusion(self, r_image_embeds, t_image_embeds, text, train=True, return_raw=False):
        """blip_stage1.py:67-92 in eval mode: `return_raw` -> z_t object (stage-II input); otherwise the normalised
        256-d query feature `F.normalize(text_proj(z_t[:, 0]))` used by stage-I retrieval."""
        if train:
            raise NotImplementedError("forward only: the contrastive training branch (blip_stage1.py:88-92) is out of scope")
        ids, mask = encode_text(self.tokenizer, text, self.device)          # blip_stage1.py:72-73
        z = self.z_t(r_image_embeds, ids, mask)
        if return_raw:
            return z
        heads = self.engines()[2]
        return ops.l2_normalize(ops.linear_f32(z.last_hidden_state[:, 0, :], heads["tw"], heads["tb"]))


def blip_stage1(pretrained: str = "", **kwargs) -> BLIP_Retrieval:
    model = BLIP_Retrieval(**kwargs)
    if pretrained:
        if model.tokenizer is None:
            raise RuntimeError("blip_stage1(pretrained=...): real weights need the real WordPiece tokenizer, and no bert-base-uncased "
                               "vocabulary was found - pass tokenizer=blip.init_tokenizer(vocab_file=...)")
        from .checkpoint import load_stage1_checkpoint
        model, msg = load_stage1_checkpoint(model, pretrained)      # blip.py:215-237 semantics
        print("missing keys:")
        print(msg.missing_keys)
        # Real weights: "text32" - the text side on fp32 rows as split8 operands (fp16 + two scaled-fp8 correction products), fp32 text
        # stream, fp32 self-attention, erf GELU; ViT and cross-attention block fp16 (blip_stage2.blip_stage2 has the numbers).
        model.set_precision("text32")
    return model
