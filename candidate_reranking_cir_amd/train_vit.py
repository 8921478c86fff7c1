"""ViT fine-tuning inside the stage-II training step (`--blip-img-tune`, stage2_train.py:87-92, 191-199; SURVEY section 8(f)-4, the part
round 3 refused): the target images' tokens come from `model.img_embed` in .train() mode WITH a graph, the loss's gradient reaches them
through the cross-attention K|V projections of all 12 two-branch layers (`NlvrTrainer.need_dfeats`), and continues through the patch
encoder (vit.py:113-194 + timm PatchEmbed).  The stage-II image encoder is built with drop_path_rate 0.1 (blip_stage2.py:37; vit.py:153:
block i drops each SAMPLE's attention / MLP branch with probability linspace(0, 0.1, depth)[i] and scales the kept ones by 1 / keep,
vit.py:98-109 with timm's DropPath); its dropout probabilities are 0.  DropPath here is counter-based like the text side's dropout: the
per-sample scales of a step come from a seeded host generator (depth x 2 x B numbers), the backward reuses them; with rate 0 the pass is
deterministic and is what tests/golden/train_imgtune.npz pins.

Same construction as `train.NlvrTrainer`: an explicit forward that keeps what the reverse pass needs and a hand-written reverse pass on the
libcirrank kernels - 16-bit MFMA operands, fp32 residual stream and gradients of it, the fused attention pair (197 / 577 keys, no mask, no
dropout: the dropout-free instantiations), the four weight gradients of a block in one grouped launch, parameters / 16-bit copies / gradients
in flat buffers (`train._Slab`) so that `train.AdamW` updates the ViT in one launch.  Pre-LayerNorm blocks: x + f(LayerNorm(x)), so the
LayerNorm adjoint's input is the saved residual stream itself and its result is added to the skip gradient.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch

from . import lib as _lib
from . import ops, train_ops as T
from .train import _Lin, _LN, _Slab, _cast, _install_grads, _unscale_and_check

_P = "visual_encoder."


def vit_train_dtype(model) -> torch.dtype:
    """Operand type of the ViT's training step: the model's token type when it is 16-bit, fp16 under the exact mode (fp32 tokens) - the
    same mapping `train.train_dtype` applies to the text side (the reference trains under fp16 autocast, stage2_train.py:210-218)."""
    return model.token_dtype if model.token_dtype in (torch.float16, torch.bfloat16) else torch.float16


class VitTrainer:
    def __init__(self, model):
        self.model, self.geo = model, model.vit_geometry
        self.dtype = vit_train_dtype(model)
        self._scale = 64 ** -0.5                                                    # vit.py:50 (head dimension 64)
        self.seed, self.step_no = 0, 0

    def _drop_path_scales(self, bsz: int, device):
        """(depth, 2, B) fp32: 0 for a dropped (block, branch, sample), 1 / keep for a kept one; None when the rate is 0 (vit.py:153)."""
        rate = float(getattr(self.geo, "drop_path_rate", 0.0))
        if rate <= 0.0 or self.geo.depth < 2:
            return None, [0.0] * self.geo.depth
        import random
        rng = random.Random(self.seed * 1000003 + self.step_no * 7919 + 17)
        ps = [rate * i / (self.geo.depth - 1) for i in range(self.geo.depth)]      # torch.linspace(0, rate, depth)
        rows = [[[0.0 if rng.random() < p else 1.0 / (1.0 - p) for _ in range(bsz)] for _ in (0, 1)] for p in ps]
        return torch.tensor(rows, dtype=torch.float32, device=device), ps

    def _trained(self, name: str) -> bool:
        return name.startswith(_P)

    def _pack(self):
        slab = getattr(self, "slab", None)
        if slab is None or slab.dtype != self.dtype or not slab.valid():
            P = {n: p for n, p in self.model.named_parameters() if self._trained(n)}
            slab = self.slab = _Slab(P, list(P), self.dtype)
            slab.begin_step()
            lins: List[_Lin] = []

            def lin(name):
                lins.append(_Lin(slab, name))
                return lins[-1]
            ln = lambda name: _LN(slab, name, self.geo.layer_norm_eps)
            self.pe = lin(_P + "patch_embed.proj")                                  # Conv2d(3, D, p, p, stride p) == Linear over flattened patches
            self.blocks: List[Dict] = []
            for i in range(self.geo.depth):
                b = f"{_P}blocks.{i}."
                self.blocks.append(dict(ln1=ln(b + "norm1"), qkv=lin(b + "attn.qkv"), proj=lin(b + "attn.proj"), ln2=ln(b + "norm2"),
                                        fc1=lin(b + "mlp.fc1"), fc2=lin(b + "mlp.fc2")))
            self.lnf = ln(_P + "norm")
            slab.plan = T.TransposePlan([l.transpose_entry for l in lins], slab.flat32.device)
        slab.begin_step()
        d = self.geo.width
        self.cls, self.pos = slab.w32(_P + "cls_token").view(d), slab.w32(_P + "pos_embed").view(-1, d)
        assert self.pos.shape[0] == self.geo.num_tokens, "pos_embed does not match the ViT geometry"
        self.dcls, self.dpos = slab.grad(_P + "cls_token").view(d), slab.grad(_P + "pos_embed").view(-1, d)

    def _heads(self, x: torch.Tensor, bsz: int, n: int, part: int) -> torch.Tensor:
        return x.view(bsz, n, 3, self.geo.num_heads, 64)[:, :, part].permute(0, 2, 1, 3)

    def _ln16(self, ln: _LN, x: torch.Tensor):
        """Pre-LayerNorm: only the 16-bit operand copy is needed (the fp32 input itself is the saved tensor of the adjoint)."""
        return ops.layernorm(x, ln.g, ln.b, ln.eps, want32=False, dtype16=self.dtype, stream_dtype=torch.float32)

    # ------------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        """(B, 3, H, W) -> (B, N, D) fp32 image tokens (VisionTransformer.forward, vit.py:180-194); keeps what backward needs."""
        self._pack()
        self.model._vit_stale = True
        geo, dt = self.geo, self.dtype
        if image.shape[-1] != geo.image_size or image.shape[-2] != geo.image_size:
            raise ValueError(f"image size {tuple(image.shape[-2:])} != model image_size {geo.image_size}")
        if image.dtype not in (torch.float32, dt):
            image = image.float()
        bsz, d, n = image.shape[0], geo.width, geo.num_tokens
        f32 = torch.float32
        patches = ops.patchify(image.contiguous(), geo.patch_size, dt)              # PatchEmbed im2col (B * P, 3 p p)
        x = ops.vit_assemble(self.pe.fwd(patches, f32), self.cls, self.pos, bsz).view(bsz * n, d)      # vit.py:184-187, fp32 stream
        self.step_no += 1
        dp, dp_rates = self._drop_path_scales(bsz, x.device)
        # What the reverse pass needs travels WITH THE CALL (the autograd node keeps it), not on the trainer: the reference embeds the
        # reference and the target images in `blip_bs` mini-batches, every call with a graph (stage2_train.py:191-199) - several calls
        # are live at once, and each backward must find its own activations and its own DropPath draw (round-4 advisor finding).
        sv = {"patches": patches, "bsz": bsz, "blocks": [], "dp": dp, "dp_rates": dp_rates, "epoch": _lib.PARAM_EPOCH[0]}
        heads = lambda t, j: self._heads(t, bsz, n, j)

        def branch(lin, a16, res, i, j):
            """res + DropPath(lin(a16)) (vit.py:108-109): the residual joins in the GEMM epilogue unless block i drops samples."""
            if dp is None or dp_rates[i] <= 0.0:
                return lin.fwd(a16, f32, residual=res)
            return T.rows_scale_add(res, lin.fwd(a16, f32), dp[i, j], n)
        for i, blk in enumerate(self.blocks):
            _, h16 = self._ln16(blk["ln1"], x)                                       # vit.py:108
            qkv = blk["qkv"].fwd(h16, dt)                                            # (B N, 3 D), vit.py:72
            ctx = torch.empty((bsz * n, d), dtype=dt, device=x.device)
            ctx32 = torch.empty((bsz * n, d), dtype=f32, device=x.device)
            c4, c32 = ctx.view(bsz, n, geo.num_heads, 64).permute(0, 2, 1, 3), ctx32.view(bsz, n, geo.num_heads, 64).permute(0, 2, 1, 3)
            lse = T.attention_train_fwd(heads(qkv, 0), heads(qkv, 1), heads(qkv, 2), None, c4, self._scale, 0.0, 0, out32=c32)    # vit.py:73-83
            x1 = branch(blk["proj"], ctx, x, i, 0)                                  # vit.py:84, :108
            _, h2 = self._ln16(blk["ln2"], x1)
            z16 = blk["fc1"].fwd(h2, dt)                                             # vit.py:36
            f16 = T.eltwise(z16, T.MODE_GELU, out_dtype=dt)                         # vit.py:37
            x2 = branch(blk["fc2"], f16, x1, i, 1)                                  # vit.py:39, :109
            sv["blocks"].append(dict(x=x, h16=h16, qkv=qkv, ctx=ctx, ctx32=ctx32, lse=lse, x1=x1, h2=h2, z16=z16, f16=f16))
            x = x2
        sv["xf"] = x
        y32, _ = self.lnf.fwd(x, dt)                                                 # vit.py:192
        self.sv = sv                                                                 # (the latest call's state: tests / tools read it)
        return y32.view(bsz, n, d), sv

    # ------------------------------------------------------------------------------------------------ backward
    @torch.no_grad()
    def backward(self, dfeats: torch.Tensor, sv: Dict = None) -> Dict[str, torch.Tensor]:
        """dfeats (B, N, D) fp32 -> {parameter name: fp32 gradient} for every visual_encoder.* parameter, from the saved state `sv` of
        the forward call being differentiated (default: the latest call's).  Every backward writes a FRESH flat gradient buffer
        (`train._install_grads` folds it into .grad with one flat add), so the mini-batches of one step - and their loss scales - do
        not meet inside a buffer."""
        sv = self.sv if sv is None else sv
        geo, dt = self.geo, self.dtype
        if sv["epoch"] != _lib.PARAM_EPOCH[0]:
            raise RuntimeError("img_embed (train mode): the parameters were updated (optimizer step) between this forward and its backward - "
                               "the reverse pass would run on other weights than the forward did")
        slab = self.slab
        slab.gflat = torch.zeros_like(slab.flat32)
        d = geo.width
        self.dcls, self.dpos = slab.grad(_P + "cls_token").view(d), slab.grad(_P + "pos_embed").view(-1, d)
        bsz, n = sv["bsz"], geo.num_tokens
        dev = dfeats.device
        g = dfeats.contiguous().float().view(bsz * n, d)
        # fp16 operands: run the (linear) pass on S * dfeats, S a power of two putting the largest entry near 512 (train.NlvrTrainer.backward)
        # When the gradient comes out of the two-branch encoder's reverse pass (the normal case), that pass's scale is reused: its own
        # internal gradients - of which dfeats is a 24-term sum - sat in fp16's range under it, and reading max|dfeats| back would make the
        # host wait for the whole fusion backward before the first ViT launch.
        self.grad_scale = 1.0
        if dt == torch.float16:
            hint = getattr(getattr(self.model, "_trainer", None), "dfeats_scale", None)
            if hint is not None:
                self.grad_scale = float(hint)
                self.model._trainer.dfeats_scale = None
            else:
                amax = float(g.abs().max())
                self.grad_scale = 2.0 ** round(math.log2(512.0 / amax)) if amax > 0 and math.isfinite(amax) else 1.0
            if self.grad_scale != 1.0:
                g = T.eltwise(g, T.MODE_SCALE, p_drop=self.grad_scale)
        g = self.lnf.bwd(sv["xf"], g)
        heads = lambda t, j: self._heads(t, bsz, n, j)
        dp, dp_rates = sv["dp"], sv["dp_rates"]

        def branch_grad(gq, i, j):
            """The stream gradient as the branch's 16-bit operand: scaled per sample where block i drops samples."""
            return _cast(gq, dt) if dp is None or dp_rates[i] <= 0.0 else T.rows_scale_add(None, gq, dp[i, j], n, out_dtype=dt)
        for i in reversed(range(len(self.blocks))):
            blk, s = self.blocks[i], sv["blocks"][i]
            wq: list = []
            # x2 = x1 + DropPath(fc2(gelu(fc1(LayerNorm2(x1)))))
            g16 = branch_grad(g, i, 1)
            df16 = blk["fc2"].bwd16(s["f16"], g16, dx_dtype=dt, bias=True, queue=wq)
            dz16 = T.gelu_bwd16(df16, s["z16"], sums=blk["fc1"].db)
            dh2 = blk["fc1"].bwd16(s["h2"], dz16, queue=wq)                         # (B N, D) fp32
            g = T.eltwise(blk["ln2"].bwd_res(s["x1"], dh2, dt, want_dt=False)[0], T.MODE_ADD, g)
            # x1 = x + DropPath(proj(attention(LayerNorm1(x))))
            g16 = branch_grad(g, i, 0)
            dctx16 = blk["proj"].bwd16(s["ctx"], g16, dx_dtype=dt, bias=True, queue=wq)
            dqkv16 = torch.empty((bsz * n, 3 * d), dtype=dt, device=dev)
            c4 = lambda t: t.view(bsz, n, geo.num_heads, 64).permute(0, 2, 1, 3)
            T.attention_train_bwd(heads(s["qkv"], 0), heads(s["qkv"], 1), heads(s["qkv"], 2), None, c4(s["ctx"]), c4(dctx16), s["lse"],
                                  heads(dqkv16, 0), heads(dqkv16, 1), heads(dqkv16, 2), self._scale, 0.0, 0, out32=c4(s["ctx32"]))
            dh = blk["qkv"].bwd16(s["h16"], dqkv16, bias=True, queue=wq)
            g = T.eltwise(blk["ln1"].bwd_res(s["x"], dh, dt, want_dt=False)[0], T.MODE_ADD, g)
            T.wgrad_grouped(wq)
        # x0 = cat(cls, patch_embed(image)) + pos_embed (vit.py:182-187)
        g3 = g.view(bsz, n, d)
        T.colsum(g.view(bsz, n * d), self.dpos.reshape(-1))
        T.colsum(g3[:, 0], self.dcls)
        dproj16 = _cast(g3[:, 1:].contiguous().view(bsz * (n - 1), d), dt)
        self.pe.bwd16(sv["patches"], dproj16, need_dx=False, bias=True)             # pixels are inputs
        self.grads_finite = _unscale_and_check(slab.gflat, self.grad_scale) if dt == torch.float16 else None      # (one pass: train.py)
        slab.checked = None if self.grads_finite is None else (slab.gflat.data_ptr(), self.grads_finite, slab.gflat._version)
        if dt != torch.float16 and self.grad_scale != 1.0:
            slab.gflat = T.eltwise(slab.gflat, T.MODE_SCALE, p_drop=1.0 / self.grad_scale)
        return {name: slab.grad(name) for name in slab.names}


class _VitTrainFn(torch.autograd.Function):
    """One autograd node around VitTrainer.forward / backward.  The node owns the saved state of ITS call, so any number of calls
    may be live between forward and backward - the reference's loop embeds reference and target images in `blip_bs` mini-batches,
    each with a graph (stage2_train.py:191-199); calls whose output never receives a gradient (the reference images: z_t is formed
    under no_grad) simply drop their state with their output.  A second backward through one call raises (its state is released)."""

    @staticmethod
    def forward(ctx, anchor, trainer, image):
        ctx.trainer = trainer
        out, ctx.sv = trainer.forward(image)
        return out

    @staticmethod
    def backward(ctx, dfeats):
        tr, sv = ctx.trainer, ctx.sv
        if sv is None:
            raise RuntimeError("img_embed (train mode): second backward through the same forward; run the forward again")
        ctx.sv = None
        _install_grads(tr, tr.backward(dfeats, sv))
        return None, None, None


def vit_train(model, image: torch.Tensor) -> torch.Tensor:
    """(B, N, D) fp32 image tokens of `img_embed` in training mode, differentiable w.r.t. the model's visual_encoder parameters."""
    tr = getattr(model, "_vit_trainer", None)
    if tr is None or tr.dtype != vit_train_dtype(model):
        tr = model._vit_trainer = VitTrainer(model)
        tr.anchor = torch.zeros((1,), device=model.device, requires_grad=True)
    return _VitTrainFn.apply(tr.anchor, tr, image.to(model.device))
