"""Stage-I MED forward in `.train()` mode: the dropout the reference leaves ON while it forms z_t.

The reference's training loop puts BOTH models in training mode (stage2_train.py:165-166 `model.train(); model_stage1.train()`) and then
computes z_t from the frozen stage-I model under `torch.no_grad()` (stage2_train.py:200-203) - so every `nn.Dropout` of the stage-I BERT
is active in that forward: after the embedding LayerNorm (med.py:108), on the attention probabilities of the self- and the
cross-attention (med.py:225), on the output dense of both attention blocks (med.py:252) and of the FFN (med.py:330), all with the
med_config probabilities (0.1).  z_t is therefore a noisy input of the stage-II step, a regulariser the loop relies on.  Rounds 3-5 formed
z_t with the deterministic inference engine (a stated deviation); this module is the train-mode forward (round 6).

Arithmetic = the training plan of train.py: 16-bit operands, fp32 residual stream, the FUSED training kernels (cir_attention_train_fwd:
mask + softmax + dropout + P.V in registers; cir_residual_layernorm_train: dropout(dense) + residual + LayerNorm in one pass) on the
weights of a `MedEngine`.  The draw: one 62-bit base seed per call from torch's global (CPU) generator - `torch.manual_seed` governs it,
as it governs the reference's - and one counter-based site seed per (layer, site); element numbering as in include/cirrank.h, so the
masks can be regenerated on the host (tests/helpers.pair_keep / splitmix_keep; tests/test_train_med_gpu.py hands them to the oracle).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops, train_ops as T
from .engine import MedEngine, additive_encoder_mask, additive_self_mask

# site ids (second argument of `site`): which dropout of a layer
SITE_EMB, SITE_SELF_ATTN, SITE_SELF_OUT, SITE_CROSS_ATTN, SITE_CROSS_OUT, SITE_FFN_OUT = 0, 1, 2, 3, 4, 5


def site_seed(base: int, layer: int, site: int) -> int:
    """Counter-based seed of one dropout site of one call (62 bits)."""
    s = (int(base) * 1000003 + 7919) & (2 ** 62 - 1)
    for v in (layer, site):
        s = (s * 131 + int(v) + 1) & (2 ** 62 - 1)
    return s


class MedDropoutForward:
    """`forward(ids, mask, enc16)` -> (z_t fp32 (Q, L, D), its 16-bit copy) with dropout at the reference's six kinds of sites."""

    def __init__(self, engine: MedEngine, p_hidden: float, p_attn: float):
        assert engine.dtype in (torch.float16, torch.bfloat16) and engine.xdtype == engine.dtype, "16-bit MedEngine (the training plan's operands)"
        self.eng, self.p_hidden, self.p_attn = engine, float(p_hidden), float(p_attn)
        self.last_seed: Optional[int] = None

    def _heads(self, x: torch.Tensor, groups: int, rows: int, part: int, parts: int) -> torch.Tensor:
        """(groups * rows, parts * D) projection(s) -> (groups, H, rows, 64) head view of projection `part` (no copy)."""
        h = self.eng.geo.num_attention_heads
        return x.view(groups, rows, parts, h, 64)[:, :, part].permute(0, 2, 1, 3)

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, enc16: torch.Tensor, enc_mask: Optional[torch.Tensor] = None,
                seed: Optional[int] = None):
        eng = self.eng
        geo, dt = eng.geo, eng.dtype
        ph, pa = self.p_hidden, self.p_attn
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())     # torch's global CPU generator: no device read
        self.last_seed = seed
        q_n, l = input_ids.shape
        d, n = geo.hidden_size, enc16.shape[1]
        r = q_n * l
        eps, scale = geo.layer_norm_eps, 64 ** -0.5
        hs, h16 = ops.embed_layernorm(input_ids, eng.word, eng.posemb, eng.ge, eng.be, eps, dt, stream_dtype=torch.float32)   # med.py:87-107
        hs, h16 = hs.view(r, d), h16.view(r, d)
        if ph > 0:                                                                                                            # med.py:108
            hs = T.eltwise(hs, T.MODE_DROPOUT, p_drop=ph, seed=site_seed(seed, 0, SITE_EMB))
            h16 = T.eltwise(hs, T.MODE_SCALE, out_dtype=dt, p_drop=1.0)
        smask = additive_self_mask(attention_mask).view(q_n, l)
        emask = None
        if enc_mask is not None:                         # (finfo.min -> a finite value: the fused kernel works in the log2 domain)
            emask = additive_encoder_mask(enc_mask).clamp_min(-3.0e4).view(q_n, n).contiguous()
        enc2 = enc16.reshape(q_n * n, enc16.shape[2])
        if enc2.dtype != dt:
            enc2 = ops.gather_rows(enc2, None, dt)
        for i, ly in enumerate(eng.layers):
            qkv = ops.gemm(h16, ly["wqkv"], ly["bqkv"])                                                                       # (r, 3 D)
            ctx = torch.empty((r, d), dtype=dt, device=hs.device)
            T.attention_train_fwd(self._heads(qkv, q_n, l, 0, 3), self._heads(qkv, q_n, l, 1, 3), self._heads(qkv, q_n, l, 2, 3), smask,
                                  self._heads(ctx, q_n, l, 0, 1), scale, pa, site_seed(seed, i, SITE_SELF_ATTN))                # med.py:193-235, dropout :225
            t = ops.gemm(ctx, ly["wo"], ly["bo"], out_dtype=torch.float32)
            _, a32, a16 = T.residual_layernorm_train(t, None, hs, ly["g1"], ly["b1"], eps, dt, p_drop=ph,
                                                     seed=site_seed(seed, i, SITE_SELF_OUT))                                   # med.py:250-253
            qc = ops.gemm(a16, ly["wq"], ly["bq"])
            kv = ops.gemm(enc2, ly["wkv"], ly["bkv"])                                                                          # (Q n, 2 D)
            cx = torch.empty((r, d), dtype=dt, device=hs.device)
            T.attention_train_fwd(self._heads(qc, q_n, l, 0, 1), self._heads(kv, q_n, n, 0, 2), self._heads(kv, q_n, n, 1, 2), emask,
                                  self._heads(cx, q_n, l, 0, 1), scale, pa, site_seed(seed, i, SITE_CROSS_ATTN))               # med.py:361-376
            t = ops.gemm(cx, ly["wco"], ly["bco"], out_dtype=torch.float32)
            _, c32, c16 = T.residual_layernorm_train(t, None, a32, ly["g2"], ly["b2"], eps, dt, p_drop=ph,
                                                     seed=site_seed(seed, i, SITE_CROSS_OUT))
            f = ops.gemm(c16, ly["w1"], ly["c1"], act=ops.ACT_GELU)                                                            # med.py:319-322
            t = ops.gemm(f, ly["w2"], ly["c2"], out_dtype=torch.float32)
            _, hs, h16 = T.residual_layernorm_train(t, None, c32, ly["g3"], ly["b3"], eps, dt, p_drop=ph,
                                                    seed=site_seed(seed, i, SITE_FFN_OUT))                                     # med.py:329-335
        return hs.view(q_n, l, d), h16.view(q_n, l, d)
