"""Training-mode `img_txt_fusion` and its backward pass on the libcirrank kernels.

What the reference does per step (stage2_train.py:202-216): z_t from the frozen stage-I model, image tokens from the (by default
frozen) ViT, then under autocast `logits = model.img_txt_fusion(z_t, target_feats, captions, train=True)` - the B x B surface of
blip_stage2.py:65-99: row i's caption / z_t expanded to B rows against all B targets - cross-entropy against arange(B),
`loss.backward()`, AdamW.  Trainable: the two-branch BERT (`text_encoder.*`) and `cls_head.*`.

`NlvrTrainer` runs that forward with every intermediate the backward needs kept on the device, and the backward as an explicit
reverse pass - no autograd graph over the kernels: each step of nlvr_encoder.BertLayer.forward (:414-476), BertSelfAttention
(:140-222), BertSelfOutput (:248-264, incl. the averaging / merge_layer variants), BertIntermediate / BertOutput (:383-409),
BertEmbeddings (:49-91) and cls_head (blip_stage2.py:50-54) has its hand-written adjoint below.  Arithmetic:
dense layers forward / dgrad on the MFMA GEMM (`ops.gemm`; dgrad over a transposed weight copy, the skip connection's fp32 gradient
added in its epilogue); the 13 weight gradients of a layer in ONE grouped launch (`train_ops.wgrad_grouped`: dy and x read as stored,
no row splits, no atomics); attention as one fused kernel with a log-sum-exp output and a recomputing adjoint whose dq | dk | dv come
out as the 16-bit operand of the fused projection's backward; each dense -> dropout -> + residual -> LayerNorm block as one pass per
direction (`residual_layernorm_train` / `layernorm_bwd_fused`, the latter also emitting the dense branch's 16-bit gradient and bias
sums); GELU's adjoint with the bias sums in one 16-bit pass.  Triplets are ordered candidate-major, so the cross-attention keys /
values of a target image are projected once per step and their gradients sum over the B queries inside the dK / dV kernel.
Precision: 16-bit MFMA operands (activations, weights, and every gradient between two dense layers), fp32 accumulation, fp32 residual
stream and its gradient, fp32 LayerNorm inputs, fp32 weight gradients.
Dropout is counter-based (seed per site); with p = 0 the pass has no random state - reproducible up to the order of the fp32 atomic adds in
the column sums / LayerNorm and embedding adjoints - and is what the reference-gradient fixtures pin.

`fusion_train(model, ...)` wraps the pair as ONE `torch.autograd.Function`, so the reference's training step runs unchanged:
`logits = model.img_txt_fusion(z_t, feats, captions)` in `.train()` mode, `loss = F.cross_entropy(logits, gt)`, `loss.backward()`
fills `.grad` of every trainable parameter (accumulating, as autograd does); `AdamW` below is torch.optim.AdamW's update on
`cir_adamw_step`.
"""
from __future__ import annotations

import math
import weakref
from typing import Dict, List, Optional

import torch

from . import lib, ops, train_ops as T


def _cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    return T.eltwise(x.contiguous(), T.MODE_SCALE, out_dtype=dtype, p_drop=1.0)


def _row_split(rows: int, n: int, k: int) -> int:
    """Number of row chunks of a weight-gradient product dW (n, k) over `rows` rows: enough 128 x 128 tiles x chunks for two
    waves of workgroups on 256 CUs (measured, tools/bmm_bench.py: 16 chunks for a 768 x 768 weight, 8 for the 3072-wide ones;
    32 is slower again), as a divisor of `rows` that leaves >= 128 rows per chunk."""
    tiles = ((n + 127) // 128) * ((k + 127) // 128)
    want = min(16, max(8, 576 // tiles))
    best = 1
    for nb in range(2, want + 1):
        if rows % nb == 0 and rows // nb >= 128:
            best = nb
    return best


def _gemm_ok(m: int, n: int, k: int) -> bool:
    return k % 64 == 0 and n % 16 == 0


_SLABS: Dict[int, "weakref.ref"] = {}     # flat32 base pointer -> the slab that owns it (AdamW.step finds the 16-bit copy to write along)


class _Slab:
    """The trained parameters as ONE flat fp32 buffer (each nn.Parameter's `.data` re-pointed to its slice: optimizers update the
    buffer in place), their 16-bit operand copies as one flat buffer refreshed by one cast per step, and their gradients as one
    flat fp32 buffer zeroed once per step - three launches where per-tensor copies took ~900.  Slices start at multiples of 8
    elements (16-byte rows for the 16-bit views)."""

    def __init__(self, params: Dict[str, torch.nn.Parameter], names: List[str], dtype: torch.dtype):
        self.params, self.names, self.dtype = params, names, dtype
        self.off, o = {}, 0
        for n in names:
            self.off[n] = o
            o += (params[n].numel() + 7) // 8 * 8
        self.total = o
        dev = params[names[0]].device
        self.flat32 = torch.zeros((o,), dtype=torch.float32, device=dev)
        for n in names:
            p = params[n]
            v = self._view(self.flat32, n)
            v.copy_(p.data)
            p.data = v
        _SLABS[self.flat32.data_ptr()] = weakref.ref(self)
        self.flat16 = self.flat16t = self.gflat = self.plan = None
        self.checked = None                                                         # (gradient buffer pointer, its "all finite" device flag)
        self._fresh16 = None                                                        # (flat32._version, PARAM_EPOCH) flat16 was written for

    def _view(self, flat: torch.Tensor, n: str) -> torch.Tensor:
        p = self.params[n]
        return flat[self.off[n]:self.off[n] + p.numel()].view(p.shape)

    def valid(self) -> bool:
        base = self.flat32.data_ptr()
        return all(self.params[n].data_ptr() == base + 4 * self.off[n] for n in self.names)

    def begin_step(self):
        """One cast launch refreshes the persistent 16-bit copy, one multi-transpose launch the dgrad operands (`plan`, built by the
        trainer from its dense layers), one fill the fresh gradient buffer."""
        if self.flat16 is None:
            self.flat16 = torch.empty(self.flat32.shape, dtype=self.dtype, device=self.flat32.device)
            self.flat16t = torch.zeros_like(self.flat16)
        if self._fresh16 != (self.flat32._version, lib.PARAM_EPOCH[0]):             # (AdamW.step below writes flat16 in its own pass)
            T.eltwise(self.flat32, T.MODE_SCALE, p_drop=1.0, out=self.flat16)
        if self.plan is not None:
            self.plan.run(self.flat16, self.flat16t)
        self.gflat = torch.zeros_like(self.flat32)

    def mark_fresh16(self):
        """flat16 holds the 16-bit copy of flat32 AS IT IS NOW (the optimizer wrote both): the next begin_step skips its cast unless a torch
        op (another optimizer, a manual edit: the version counter moves) or another of our launches (PARAM_EPOCH) touches the parameters first."""
        self._fresh16 = (self.flat32._version, lib.PARAM_EPOCH[0])

    def w32(self, n): return self._view(self.flat32, n)
    def w16(self, n): return self._view(self.flat16, n)
    def grad(self, n): return self._view(self.gflat, n)

    def span_range(self, names: List[str]):
        """(offset, rows, trailing shape) of the slices of `names` stacked along dim 0 (they must be adjacent in the buffer: `NlvrTrainer._order`
        lays the q / k / v weights - and biases - of one attention out that way, so the three projections are one 2304-wide Linear)."""
        o = self.off[names[0]]
        rows = 0
        for n in names:
            assert self.off[n] == o + rows * (self.params[n].numel() // self.params[n].shape[0]), "group not adjacent in the slab"
            rows += self.params[n].shape[0]
        return o, rows, tuple(self.params[names[0]].shape[1:])

    def span(self, flat: torch.Tensor, names: List[str]) -> torch.Tensor:
        o, rows, tail = self.span_range(names)
        numel = rows
        for t in tail:
            numel *= t
        return flat[o:o + numel].view((rows,) + tail)


class _Lin:
    """One nn.Linear of the reference (weight (N, K), bias (N)) - or several of one input stacked -: views of the slab's persistent 16-bit
    weights, their transposed copy (the dgrad GEMM's operand; refreshed by the slab's one multi-transpose launch per step) and fp32 bias,
    built ONCE; the gradient views follow the slab's per-step gradient buffer lazily."""

    def __init__(self, slab: _Slab, name, group: bool = False):
        names = list(name) if group else [name]                                     # a group: several Linears of one input, stacked
        has_bias = (names[0] + ".bias") in slab.off
        self.slab = slab
        self.ws, self.bs = [n + ".weight" for n in names], ([n + ".bias" for n in names] if has_bias else None)
        off, n, tail = slab.span_range(self.ws)
        k = 1
        for t in tail:                                                              # a conv kernel (N, C, p, p) is the (N, C p p) Linear over patches
            k *= t
        self.n, self.k, self.off_w = n, k, off
        self.w16 = slab.flat16[off:off + n * k].view(n, k)                          # (N, K): forward operand
        self.w16t = slab.flat16t[off:off + n * k].view(k, n)                        # (K, N): dgrad operand
        self.transpose_entry = (off, n, k)
        self.bias = slab.span(slab.flat32, self.bs) if has_bias else None
        self._g = None

    def _grads(self):
        g = self.slab.gflat
        if self._g is not g:
            self._g, self._dw = g, g[self.off_w:self.off_w + self.n * self.k].view(self.n, self.k)
            self._db = self.slab.span(g, self.bs) if self.bs is not None else None

    @property
    def dw(self):
        self._grads()
        return self._dw

    @property
    def db(self):
        self._grads()
        return self._db

    def fwd(self, x16: torch.Tensor, out_dtype: torch.dtype, out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        m, k = x16.shape
        n = self.w16.shape[0]
        if _gemm_ok(m, n, k):
            return ops.gemm(x16, self.w16, self.bias, residual=residual, out_dtype=out_dtype, out=out)
        assert out is None and residual is None
        y = T.bmm(x16.unsqueeze(0), self.w16.unsqueeze(0), False, True, out_dtype=torch.float32)[0]
        if self.bias is not None:
            y = T.eltwise(y, T.MODE_ADD, self.bias.unsqueeze(0).expand(m, n).contiguous())
        return y if out_dtype == torch.float32 else _cast(y, out_dtype)

    def bwd(self, x16: torch.Tensor, dy: torch.Tensor, need_dx: bool = True) -> Optional[torch.Tensor]:
        """dy fp32 (M, N): accumulates dW, db; returns dx fp32 (M, K)."""
        m, k = x16.shape
        n = self.w16.shape[0]
        dy16 = _cast(dy, x16.dtype)
        if self.db is not None:
            T.colsum(dy, self.db)
        # dW (N, K) = dy^T x on cir_bmm (operands read as stored: trans_a), split over row chunks into partial sums so that the
        # 36-tile products of a 768 x 768 weight fill the chip; the partials are summed into dW by the column-sum kernel
        nb = _row_split(m, n, k)
        if nb == 1:
            T.bmm(dy16.unsqueeze(0), x16.unsqueeze(0), True, False, out=self.dw.unsqueeze(0), accumulate=True)
        else:
            part = T.bmm(dy16.view(nb, m // nb, n), x16.view(nb, m // nb, k), True, False, out_dtype=torch.float32)
            T.colsum(part.view(nb, n * k), self.dw.view(-1))
        if not need_dx:
            return None
        if _gemm_ok(m, k, n):                                                       # dx (M, K) = dy (M, N) . (W^T (K, N))^T
            return ops.gemm(dy16, self.w16t, None, out_dtype=torch.float32)
        return T.bmm(dy16.unsqueeze(0), self.w16.unsqueeze(0), False, False, out_dtype=torch.float32)[0]


    def bwd16(self, x16: torch.Tensor, dy16: torch.Tensor, need_dx: bool = True, dx_dtype: torch.dtype = torch.float32,
              residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, bias: bool = False,
              queue: Optional[list] = None) -> Optional[torch.Tensor]:
        """dy16 (M, N) ALREADY the 16-bit operand (any row stride: written by the fused row kernels, the attention adjoint or the dgrad
        product before): accumulates dW - and db when `bias` (otherwise the producer of dy16 summed it) -; returns
        dx = dy . W (+ residual: the fp32 gradient arriving over the skip connection, added in the GEMM epilogue) in `dx_dtype`."""
        m, k = x16.shape
        n = self.w16.shape[0]
        if bias and self.db is not None:
            T.colsum16(dy16, self.db)
        # dW (N, K) += dy^T x with both operands read as stored, the rows split over workgroups so that the 36-tile product of a
        # 768 x 768 weight fills the chip; every workgroup adds its partial tile straight into dW (atomics: no partial tensor)
        if n % 128 == 0 and k % 128 == 0:
            # LDS-DMA / transposing-read kernel (train_wgrad.hip).  `queue` (a list): the product is deferred and launched together with
            # the layer's other weight gradients - ~940 output tiles fill the chip without splitting any tile's rows over workgroups
            if queue is not None:
                queue.append((dy16, x16, self.dw))
            else:
                T.wgrad(dy16, x16, self.dw)
        else:
            nb = _row_split(m, n, k)
            T.bmm(dy16.unflatten(0, (nb, m // nb)), x16.unflatten(0, (nb, m // nb)), True, False, out=self.dw.unsqueeze(0).expand(nb, n, k),
                  accumulate="atomic")
        if not need_dx:
            return None
        return ops.gemm(dy16, self.w16t, None, residual=residual, out_dtype=dx_dtype, out=out)


class _Lin2:
    """The two branches' Linears of one kind (adjacent in the slab, `NlvrTrainer._order`): forward and dgrad of BOTH as one batched GEMM
    (batch 2; measured on the 8192-row shapes of the step: 22 against 35 us for the 768 x 768 products, 60 against 88 us for the stacked
    q|k|v dgrad - a 9.7-GFLOP product is mostly launch, prologue and epilogue).  Weight / bias gradients stay per branch (`.l[b]`)."""

    def __init__(self, l0: _Lin, l1: _Lin):
        slab, n, k = l0.slab, l0.n, l0.k
        assert l1.n == n and l1.k == k and l1.off_w == l0.off_w + n * k, "branch twins not adjacent in the slab"
        self.l = (l0, l1)
        self.w16 = slab.flat16[l0.off_w:l0.off_w + 2 * n * k].view(2, n, k)
        self.w16t = slab.flat16t[l0.off_w:l0.off_w + 2 * n * k].view(2, k, n)
        ob = slab.off[l0.bs[0]]
        assert slab.off[l1.bs[0]] == ob + n
        self.bias = slab.flat32[ob:ob + 2 * n].view(2, n)

    BATCHED = True        # False: the same products as two launches into the same tensors (A/B: tools/train_dbg.py, CIR_TRAIN_PAIRS=0)

    def _gemm(self, a3, w3, bias, residual, out_dtype, out):
        if self.BATCHED:
            return ops.gemm(a3, w3, bias, residual=residual, out_dtype=out_dtype, out=out)
        if out is None:
            out = torch.empty((2, a3.shape[1], w3.shape[1]), dtype=out_dtype, device=a3.device)
        for b in (0, 1):
            ops.gemm(a3[b], w3[b], None if bias is None else bias[b], residual=None if residual is None else residual[b], out_dtype=out_dtype, out=out[b])
        return out

    def fwd(self, x3: torch.Tensor, out_dtype: torch.dtype, out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x3 (2, M, K) (a stride-0 batch dimension shares one input) -> (2, M, N)."""
        return self._gemm(x3, self.w16, self.bias, residual, out_dtype, out)

    def dgrad(self, dy3: torch.Tensor, dx_dtype: torch.dtype, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dy3 (2, M, N) -> dx (2, M, K) = dy . W (+ residual) per branch."""
        return self._gemm(dy3, self.w16t, None, residual, dx_dtype, out)

    def wgrad(self, x3: torch.Tensor, dy3: torch.Tensor, queue: list, bias: bool = False):
        for b in (0, 1):
            self.l[b].bwd16(x3[b], dy3[b], need_dx=False, bias=bias, queue=queue)


class _LN:
    def __init__(self, slab: _Slab, name: str, eps: float):
        self.eps, self.slab, self.name = eps, slab, name
        self.g, self.b = slab.w32(name + ".weight"), slab.w32(name + ".bias")
        self._g = None

    def _grads(self):
        g = self.slab.gflat
        if self._g is not g:
            self._g, self._dg, self._db = g, self.slab.grad(self.name + ".weight"), self.slab.grad(self.name + ".bias")

    @property
    def dg(self):
        self._grads()
        return self._dg

    @property
    def db(self):
        self._grads()
        return self._db

    def fwd(self, pre: torch.Tensor, dtype: torch.dtype):
        return ops.layernorm(pre, self.g, self.b, self.eps, want32=True, dtype16=dtype, stream_dtype=torch.float32)

    def bwd(self, pre: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
        return T.layernorm_bwd(pre, self.g, dy, self.dg, self.db, self.eps)

    def fwd_res(self, t0, t1, res, dtype, alpha=1.0, p_drop=0.0, seed=0, **out):
        """(pre, y32, y16) of LayerNorm(dropout(alpha * (t0 + t1)) + res): one launch (cir_residual_layernorm_train)."""
        return T.residual_layernorm_train(t0, t1, res, self.g, self.b, self.eps, dtype, alpha, p_drop, seed, **out)

    def bwd_res(self, pre, dy, dtype, **kw):
        """(d pre fp32, 16-bit gradient of the dense branch behind the dropout) - cir_layernorm_bwd_fused."""
        return T.layernorm_bwd_fused(pre, self.g, dy, self.dg, self.db, self.eps, dtype, **kw)


def train_dtype(model) -> torch.dtype:
    """Operand type of the training step: the model's own when it is 16-bit; fp16 under the inference-only modes with an fp32 text side
    ("text32" - what the factories set for real weights - and "exact"): the reference trains under fp16 autocast (stage2_train.py:210-218)."""
    return model.compute_dtype if model.compute_dtype in (torch.float16, torch.bfloat16) else torch.float16


class NlvrTrainer:
    """Forward (with saved activations) and backward of the two-branch encoder + cls_head for a B x B training batch."""

    def __init__(self, model, p_hidden: float = 0.1, p_attn: float = 0.1, seed: int = 0):
        self.model, self.p_hidden, self.p_attn, self.seed = model, float(p_hidden), float(p_attn), int(seed)
        self.geo = model.bert_geometry
        self.dtype = train_dtype(model)
        self.step_no = 0
        self._hd = self.geo.hidden_size // self.geo.num_attention_heads
        self.need_dfeats = False          # blip_img_tune (stage2_train.py:183-199): also return the gradient of the target image tokens
        self.dfeats = self.dfeats_scale = None
        self._scale = self._hd ** -0.5

    # ------------------------------------------------------------------------------------------------ parameters
    _EMB = "text_encoder.embeddings."

    def _trained(self, name: str) -> bool:
        """The parameters the reference's step gives a gradient (tests/golden/train768.npz: 572 of them): every encoder-layer and
        cls_head tensor, word / position embeddings and the embedding LayerNorm (token-type embeddings and the pooler are unused)."""
        e = self._EMB
        return name.startswith(("text_encoder.encoder.layer.", "cls_head.")) or name in (
            e + "word_embeddings.weight", e + "position_embeddings.weight", e + "LayerNorm.weight", e + "LayerNorm.bias")

    @staticmethod
    def _order(names: List[str]) -> List[str]:
        """Slab order: per layer the twin (branch 0 | branch 1) dense layers as adjacent groups - all weights of a group, then its biases - so
        that (a) the q, k, v projections of one self-attention (k, v of one cross-attention) are ONE stacked Linear (`_Slab.span`) and (b) the
        two branches' Linears of one kind sit a constant stride apart: one BATCHED GEMM serves both (`_Lin2`).  Everything else keeps the
        model's own order."""
        groups = {}
        for n in names:
            if not n.endswith(".weight"):
                continue
            stem = None
            if ".attention.self0.query." in n:
                stem, parts = n[:n.index("attention.self0.query.")], [f"attention.self{b}.{x}" for b in (0, 1) for x in ("query", "key", "value")]
            elif ".crossattention.self0.key." in n:
                stem, parts = n[:n.index("crossattention.self0.key.")], [f"crossattention.self{b}.{x}" for b in (0, 1) for x in ("key", "value")]
            elif ".crossattention.self0.query." in n:
                stem, parts = n[:n.index("crossattention.self0.query.")], [f"crossattention.self{b}.query" for b in (0, 1)]
            elif ".attention.output.dense0." in n and ".crossattention." not in n:
                stem, parts = n[:n.index("attention.output.dense0.")], [f"attention.output.dense{b}" for b in (0, 1)]
            elif ".crossattention.output.dense0." in n:
                stem, parts = n[:n.index("crossattention.output.dense0.")], [f"crossattention.output.dense{b}" for b in (0, 1)]
            if stem is not None:
                groups[n] = [stem + q + "." + y for y in ("weight", "bias") for q in parts]
        grouped = {m for g in groups.values() for m in g}
        out, seen = [], set()
        for n in names:
            if n in seen:
                continue
            if n in groups:
                for m in groups[n]:
                    out.append(m); seen.add(m)
            elif n not in grouped:
                out.append(n); seen.add(n)
        for n in names:                                                             # (a grouped name whose group head is missing: keep it)
            if n not in seen:
                out.append(n); seen.add(n)
        assert sorted(out) == sorted(names)
        return out

    def _pack(self):
        """Per step: refresh the 16-bit parameter copies and a zeroed gradient buffer (three launches).  The layer objects - views of the
        persistent buffers - are built once and rebuilt only when the model was moved / re-cast."""
        slab = getattr(self, "slab", None)
        if slab is None or slab.dtype != self.dtype or not slab.valid():           # first step, or the model was moved / re-cast
            P = dict(self.model.named_parameters())
            slab = self.slab = _Slab(P, self._order([n for n in P if self._trained(n)]), self.dtype)
            slab.begin_step()                                                       # allocates the 16-bit buffers the views below slice
            self._build_layers(slab)
        slab.begin_step()
        e = self._EMB
        self.dword, self.dpos = slab.grad(e + "word_embeddings.weight"), slab.grad(e + "position_embeddings.weight")

    def _build_layers(self, slab: _Slab):
        g = self.geo
        lins: List[_Lin] = []

        def lin(name, group=False):
            lins.append(_Lin(slab, name, group))
            return lins[-1]
        grp = lambda names: lin(names, True)
        ln = lambda name: _LN(slab, name, g.layer_norm_eps)
        e = self._EMB
        self.word, self.pos = slab.w32(e + "word_embeddings.weight"), slab.w32(e + "position_embeddings.weight")
        self.ln_e = ln(e + "LayerNorm")
        self.layers: List[Dict] = []
        for i in range(g.num_hidden_layers):
            p = f"text_encoder.encoder.layer.{i}."
            ly = {}
            for b in (0, 1):
                ly[f"qkv{b}"] = grp([p + f"attention.self{b}.{n}" for n in ("query", "key", "value")])     # one 2304-wide Linear
                ly[f"o{b}"] = lin(p + f"attention.output.dense{b}")
                ly[f"cq{b}"] = lin(p + f"crossattention.self{b}.query")
                ly[f"ckv{b}"] = grp([p + f"crossattention.self{b}.{n}" for n in ("key", "value")])           # one 1536-wide Linear
                ly[f"d{b}"] = lin(p + f"crossattention.output.dense{b}")
            for kind in ("qkv", "o", "cq", "ckv", "d"):                                # the twins as one batched GEMM each
                ly[kind] = _Lin2(ly[kind + "0"], ly[kind + "1"])
            for c, b in (("A", 0), ("B", 1)):
                ly[f"ln1{b}"] = ln(p + f"attention.output.LayerNorm{c}")
                ly[f"ln2{b}"] = ln(p + f"crossattention.output.LayerNorm{c}")
            mk = p + "crossattention.output.merge_layer"
            ly["merge"] = lin(mk) if (mk + ".weight") in slab.off else None
            ly["w1"], ly["w2"], ly["ln3"] = lin(p + "intermediate.dense"), lin(p + "output.dense"), ln(p + "output.LayerNorm")
            self.layers.append(ly)
        self.c0, self.c2 = lin("cls_head.0"), lin("cls_head.2")
        slab.plan = T.TransposePlan([l.transpose_entry for l in lins], slab.flat32.device)

    def _site(self, *ids) -> int:
        s = self.seed * 1000003 + self.step_no * 7919
        for v in ids:
            s = s * 131 + int(v) + 1
        return s & (2 ** 62 - 1)

    def _drop(self, x: torch.Tensor, site: int) -> torch.Tensor:
        return x if self.p_hidden <= 0 else T.eltwise(x, T.MODE_DROPOUT, p_drop=self.p_hidden, seed=site)

    # ------------------------------------------------------------------------------------------------ attention
    def _heads(self, x: torch.Tensor, nb1: int, rows: int, part: int = 0, parts: int = 1) -> torch.Tensor:
        """(nb1 * rows, parts * D) projection(s) -> (nb1, H, rows, head_dim) view of the head slices of projection `part` (no copy)."""
        return x.view(nb1, rows, parts, self.geo.num_attention_heads, self._hd)[:, :, part].permute(0, 2, 1, 3)

    def _attn_fwd(self, q4, k4, v4, mask, site, ctx, ctx32):
        """q4 (G, H, mq, hd), k4 / v4 (G, H, mk, hd) head views of 16-bit projections: G groups of mq query rows and mk key rows;
        mask (groups, mk) additive fp32, one row per mq * H score rows, or None.  Self-attention: a group is a triplet;
        cross-attention: a group is a CANDIDATE with the B queries scored against it stacked in mq = B * L rows - its keys /
        values exist once.  ONE kernel - scores, mask, softmax, dropout, P.V tile by tile in registers - and a log-sum-exp per row
        for the recomputing backward; no score / probability tensor is materialised (cir_attention_train_fwd; head dimension 64,
        which config.BertGeometry enforces).  Writes the context into `ctx` (G*mq, D) 16-bit and its fp32 twin `ctx32` (the backward's
        D = rowsum(dO * O), cirrank.h); returns what the adjoint needs."""
        nb1, h_n, mq, _ = q4.shape
        lse = T.attention_train_fwd(q4, k4, v4, mask, self._heads(ctx, nb1, mq), self._scale, self.p_attn, site, out32=self._heads(ctx32, nb1, mq))
        return (lse, mask, site, ctx, ctx32)

    def _attn_bwd(self, dctx16, q4, k4, v4, saved, dq4, dk4, dv4):
        """dctx16 (G*mq, D) in the operand type -> dq4 / dk4 / dv4: head views (same type) of the buffer the fused projection's
        backward reads as its dy operand."""
        nb1, h_n, mq, _ = q4.shape
        lse, mask, site, ctx, ctx32 = saved
        T.attention_train_bwd(q4, k4, v4, mask, self._heads(ctx, nb1, mq), self._heads(dctx16, nb1, mq), lse, dq4, dk4, dv4,
                              self._scale, self.p_attn, site, out32=self._heads(ctx32, nb1, mq))

    # ------------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, z_t: torch.Tensor, feats: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """z_t (B, L, D) fp32, feats (B, N, Dv), ids / mask (B, L) with [ENC] set -> logits (B, B) fp32; keeps what backward needs."""
        self._pack()
        self.model._text_stale = True                                               # a training step is about to change text_encoder / cls_head:
        self.step_no += 1                                                           # the inference engine of that part repacks on its next use
        g, dt, dev = self.geo, self.dtype, z_t.device
        b_n, l = input_ids.shape
        n, d = feats.shape[1], g.hidden_size
        t_n = b_n * b_n
        r = t_n * l
        ph = self.p_hidden
        f32 = torch.float32
        # triplet t = j * B + i scores query i (caption, z_t) against target j - candidate-major, so that the B queries of one
        # target are consecutive rows and its cross-attention keys / values are projected ONCE (the reference recomputes them
        # for every query, blip_stage2.py:80-92; same values); the (B_j, B_i) result is transposed on the way out
        qi = torch.arange(b_n, device=dev).repeat(b_n)
        ids_t = input_ids.to(dev)[qi].contiguous()                                  # (T, L)
        self.sv = sv = {"ids": ids_t, "t_n": t_n, "l": l, "n": n, "b_n": b_n}
        # embeddings (BertEmbeddings: LayerNorm(word + pos), dropout) -> branch 1; z_t -> branch 0 (nlvr_encoder.py:880-892)
        pos_idx = torch.arange(l, device=dev).repeat(t_n)
        pre_e = T.eltwise(ops.gather_rows(self.word, ids_t.view(-1), f32), T.MODE_ADD, ops.gather_rows(self.pos, pos_idx, f32))
        sv["pre_e"] = pre_e
        e32, _ = self.ln_e.fwd(pre_e, dt)
        e32 = self._drop(e32, self._site(9000))
        # Both branches live in ONE (2, R, .) tensor per activation: the twin dense layers are then one batched GEMM each (`_Lin2`), and the
        # shared FFN reads the same memory as its 2R stacked rows.
        h32 = torch.empty((2, r, d), dtype=f32, device=dev)
        h32[0].copy_(ops.gather_rows(z_t.to(dev).float().contiguous().view(b_n, l * d), qi, f32).view(r, d))
        h32[1].copy_(e32)
        h16 = _cast(h32, dt)
        cand16 = _cast(feats.to(dev).float().contiguous(), dt).view(b_n * n, -1)                          # (B*N, Dv): each target once
        sv["cand16"] = cand16
        cand2 = cand16.unsqueeze(0).expand(2, b_n * n, cand16.shape[1])                                   # one input, two branches (batch stride 0)
        smask = ((1.0 - attention_mask.to(dev).float()) * -10000.0)[qi].contiguous()                      # (T, L), nlvr_encoder.py:773-774
        smask2 = smask.repeat(2, 1)                                                                       # the same key masks for both branches' groups
        sv["layers"] = []
        for i, ly in enumerate(self.layers):
            s = {"h16": h16}
            qkv = ly["qkv"].fwd(h16, dt)                                              # (2, R, 3D): one batched GEMM for both branches
            ctx = torch.empty((2, r, d), dtype=dt, device=dev)
            ctx32 = torch.empty((2, r, d), dtype=f32, device=dev)                     # fp32 twin of the context: the backward's D = rowsum(dO * O)
            # both branches' self-attentions in ONE launch: group = (branch, triplet) - 2T groups of 32 x 32 one-tile problems fill the
            # chip better than T (3072 waves are 3 per SIMD), and the dropout rows of the two branches are distinct rows of one site
            s["sa"] = self._attn_fwd(*(self._heads(qkv.view(2 * r, 3 * d), 2 * t_n, l, j, 3) for j in range(3)), smask2, self._site(i, 0, 1),
                                     ctx.view(2 * r, d), ctx32.view(2 * r, d))
            # BertSelfOutput (nlvr_encoder.py:399-409): LayerNorm(dropout(dense(ctx)) + h) - dropout, sum and LayerNorm in one pass per branch
            t = ly["o"].fwd(ctx, f32)
            pre1, a32, a16 = torch.empty((2, r, d), dtype=f32, device=dev), torch.empty((2, r, d), dtype=f32, device=dev), torch.empty((2, r, d), dtype=dt, device=dev)
            for b in (0, 1):
                ly[f"ln1{b}"].fwd_res(t[b], None, h32[b], dt, 1.0, ph, self._site(i, b, 2), pre=pre1[b], y32=a32[b], y16=a16[b])
            cq = ly["cq"].fwd(a16, dt)
            ckv = ly["ckv"].fwd(cand2, dt)                                            # (2, B*N, 2D): each target's keys | values, once per branch
            c = torch.empty((2, r, d), dtype=dt, device=dev)
            c32 = torch.empty((2, r, d), dtype=f32, device=dev)
            # ... and both cross-attentions: group = (branch, target image), its B * L stacked query rows against the target's 577 keys
            kv2 = ckv.view(2 * b_n * n, 2 * d)
            s["ca"] = self._attn_fwd(self._heads(cq.view(2 * r, d), 2 * b_n, b_n * l), self._heads(kv2, 2 * b_n, n, 0, 2), self._heads(kv2, 2 * b_n, n, 1, 2),
                                     None, self._site(i, 0, 3), c.view(2 * r, d), c32.view(2 * r, d))
            if ly["merge"] is None:                                                 # layers < 6: average (nlvr_encoder.py:257-260)
                dd = ly["d"].fwd(c, f32)
                t0, t1, alpha = dd[0], dd[1], 0.5
            else:                                                                   # layers >= 6: merge_layer(cat) (:252-256): the operand is written
                cat16 = torch.empty((r, 2 * d), dtype=dt, device=dev)               # in place, 16-bit, branch b into columns [bD, (b+1)D)
                ly["d"].fwd(c, dt, out=cat16.view(r, 2, d).permute(1, 0, 2))
                s["cat16"] = cat16
                t0, t1, alpha = ly["merge"].fwd(cat16, f32), None, 1.0
            # BertSelfOutput of the cross-attention: m = dropout(average | merge) (ONE mask for both branches), LayerNormA / B (m + a_b).
            # FFN: the SAME weights serve both branches (nlvr_encoder.py:469-476) - one pass over the 2R stacked rows (one GEMM pair,
            # one GELU / LayerNorm launch, and in the backward one dgrad / wgrad product each instead of two half-sized ones)
            pre2 = torch.empty((2 * r, d), dtype=f32, device=dev)
            x32, x16 = torch.empty_like(pre2), torch.empty((2 * r, d), dtype=dt, device=dev)
            for b in (0, 1):
                rows = slice(b * r, (b + 1) * r)
                ly[f"ln2{b}"].fwd_res(t0, t1, a32[b], dt, alpha, ph, self._site(i, 2, 4), pre=pre2[rows], y32=x32[rows], y16=x16[rows])
            z16 = ly["w1"].fwd(x16, dt)                                             # the dense output in the operand type, as autocast leaves it
            f16 = T.eltwise(z16, T.MODE_GELU, out_dtype=dt)
            pre3, hn, hn16 = ly["ln3"].fwd_res(ly["w2"].fwd(f16, f32), None, x32, dt, 1.0, ph, self._site(i, 0, 5))
            s.update(qkv=qkv, ctx=ctx, pre1=pre1, a16=a16, cq=cq, ckv=ckv, c=c, pre2=pre2, x16=x16, z16=z16, f16=f16, pre3=pre3)
            sv["layers"].append(s)
            h32, h16 = hn.view(2, r, d), hn16.view(2, r, d)
        # cat(CLS_0, CLS_1) -> cls_head (nlvr_encoder.py:906-908, blip_stage2.py:50-54, 94-99)
        cls_rows = torch.arange(t_n, device=dev) * l
        hid16 = torch.cat([ops.gather_rows(h16[0], cls_rows, dt), ops.gather_rows(h16[1], cls_rows, dt)], dim=1).contiguous()
        z1 = self.c0.fwd(hid16, f32)
        y16 = T.eltwise(z1, T.MODE_RELU, out_dtype=dt)
        logits2 = self.c2.fwd(y16, f32)                                             # (T, 2)
        sv.update(hid16=hid16, z1=z1, y16=y16, cls_rows=cls_rows)
        return logits2[:, 0].contiguous().view(b_n, b_n).t().contiguous()           # (B_j, B_i) -> (B_i, B_j)

    def head_mask(self) -> torch.Tensor:
        """(B*B, hidden) bool: which cls_head.0 units were active in the last forward, rows in the reference's order (i * B + j)."""
        b = self.sv["b_n"]
        return (self.sv["z1"] > 0).view(b, b, -1).transpose(0, 1).reshape(b * b, -1)

    # ------------------------------------------------------------------------------------------------ backward
    @torch.no_grad()
    def backward(self, dlogits: torch.Tensor) -> Dict[str, torch.Tensor]:
        """dlogits (B, B) fp32 -> {parameter name: fp32 gradient} for every text_encoder.* / cls_head.* parameter."""
        sv, g, dt = self.sv, self.geo, self.dtype
        dev = dlogits.device
        t_n, l, n, d = sv["t_n"], sv["l"], sv["n"], g.hidden_size
        r = t_n * l
        # Gradient scaling (what the reference's GradScaler does for its fp16 autocast, stage2_train.py:215-218, done inside):
        # every adjoint below is linear in the incoming gradient, so the pass runs on S * dlogits with S a power of two that
        # puts the largest entry near 512 - the 16-bit copies fed to the dgrad / wgrad GEMMs then sit in fp16's normal range
        # (hidden-state gradients are ~1e-5 per element unscaled, fp16's smallest normal is 6e-5) - and `_collect` divides by S.
        # bf16 has fp32's exponent range and needs none of this: scale 1, no unscaling pass.
        self.grad_scale = 1.0
        if dt == torch.float16:
            amax = float(dlogits.abs().max())
            self.grad_scale = 2.0 ** round(math.log2(512.0 / amax)) if amax > 0 and math.isfinite(amax) else 1.0
        dl2 = torch.zeros((t_n, 2), dtype=torch.float32, device=dev)
        dl2[:, 0] = T.eltwise(dlogits.float().t().contiguous().view(-1), T.MODE_SCALE, p_drop=self.grad_scale)
        dy1 = self.c2.bwd(sv["y16"], dl2)
        dz1 = T.eltwise(sv["z1"], T.MODE_RELU_BWD, dy1)
        dhid = self.c0.bwd(sv["hid16"], dz1)                                        # (T, 2D)
        # Between the dense layers every gradient is the 16-bit operand of the next product (written by the kernel that forms it -
        # a fused LayerNorm adjoint, the attention adjoint, a dgrad epilogue), bias gradients are summed where it is formed, and
        # the fp32 gradient of the residual stream joins in the dgrad GEMM's epilogue: no stand-alone cast / add / dropout pass.
        dh = torch.zeros((2 * r, d), dtype=torch.float32, device=dev)               # both branches stacked, as the FFN saw them
        dh[sv["cls_rows"]] = dhid[:, :d]
        dh[sv["cls_rows"] + r] = dhid[:, d:]
        ph, b_n = self.p_hidden, sv["b_n"]
        dfeats = torch.empty((b_n * n, sv["cand16"].shape[1]), dtype=torch.float32, device=dev) if self.need_dfeats else None
        dfeats_live = False
        for i in reversed(range(len(self.layers))):
            ly, s = self.layers[i], sv["layers"][i]
            w1, w2 = ly["w1"], ly["w2"]
            wq: list = []                                                           # this layer's weight-gradient products: ONE launch at its end
            dpre3, do16 = ly["ln3"].bwd_res(s["pre3"], dh, dt, dbias=w2.db, p_drop=ph, seed=self._site(i, 0, 5))
            df16 = w2.bwd16(s["f16"], do16, dx_dtype=dt, queue=wq)
            dz16 = T.gelu_bwd16(df16, s["z16"], sums=w1.db)
            dx = w1.bwd16(s["x16"], dz16, residual=dpre3, queue=wq)                 # (2R, D) fp32: FFN branch + skip
            # the two LayerNorms over m + a_b: d m = dropout'(d pre2_0 + d pre2_1) comes out of the second one's kernel
            merge = ly["merge"]
            dpre2 = torch.empty((2, r, d), dtype=torch.float32, device=dev)
            ly["ln20"].bwd_res(s["pre2"][:r], dx[:r], dt, want_dt=False, dx=dpre2[0])
            if merge is None:                                                       # average: both output denses see 0.5 * d m
                kw = dict(alpha=0.5, dbias=ly["d0"].db, dbias2=ly["d1"].db)
            else:
                kw = dict(alpha=1.0, dbias=merge.db)
            _, dm16 = ly["ln21"].bwd_res(s["pre2"][r:], dx[r:], dt, t_add=dpre2[0], p_drop=ph, seed=self._site(i, 2, 4), dx=dpre2[1], **kw)
            if merge is None:
                dd16 = dm16.unsqueeze(0).expand(2, r, d)                            # one gradient, two branches (batch stride 0)
            else:
                dcat16 = merge.bwd16(s["cat16"], dm16, dx_dtype=dt, queue=wq)       # (R, 2D)
                dd16 = dcat16.view(r, 2, d).permute(1, 0, 2)                        # branch b = columns [bD, (b+1)D)
            ly["d"].wgrad(s["c"], dd16, wq, bias=merge is not None)
            dc16 = ly["d"].dgrad(dd16, dt)                                          # (2, R, D): both branches' output-dense dgrads, one GEMM
            cq, ckv = s["cq"], s["ckv"]
            dcq16 = torch.empty((2, r, d), dtype=dt, device=dev)
            dckv16 = torch.empty((2, b_n * n, 2 * d), dtype=dt, device=dev)
            kv2, dkv2 = ckv.view(2 * b_n * n, 2 * d), dckv16.view(2 * b_n * n, 2 * d)
            self._attn_bwd(dc16.view(2 * r, d), self._heads(cq.view(2 * r, d), 2 * b_n, b_n * l), self._heads(kv2, 2 * b_n, n, 0, 2),
                           self._heads(kv2, 2 * b_n, n, 1, 2), s["ca"], self._heads(dcq16.view(2 * r, d), 2 * b_n, b_n * l),
                           self._heads(dkv2, 2 * b_n, n, 0, 2), self._heads(dkv2, 2 * b_n, n, 1, 2))
            cand2 = sv["cand16"].unsqueeze(0).expand(2, b_n * n, sv["cand16"].shape[1])
            ly["ckv"].wgrad(cand2, dckv16, wq, bias=True)
            if dfeats is not None:                                                  # ViT fine-tuning: every layer and branch adds its share
                for b in (0, 1):                                                    # (image tokens are inputs otherwise: no gradient beyond the weights)
                    ops.gemm(dckv16[b], ly[f"ckv{b}"].w16t, None, residual=dfeats if dfeats_live else None, out_dtype=torch.float32, out=dfeats)
                    dfeats_live = True
            ly["cq"].wgrad(s["a16"], dcq16, wq, bias=True)
            da = ly["cq"].dgrad(dcq16, torch.float32, residual=dpre2)               # (2, R, D) fp32: cross-attention query branch + skip
            dpre1 = torch.empty((2, r, d), dtype=torch.float32, device=dev)
            dt16 = torch.empty((2, r, d), dtype=dt, device=dev)
            for b in (0, 1):
                ly[f"ln1{b}"].bwd_res(s["pre1"][b], da[b], dt, dbias=ly[f"o{b}"].db, p_drop=ph, seed=self._site(i, b, 2), dx=dpre1[b], dt16=dt16[b])
            ly["o"].wgrad(s["ctx"], dt16, wq)
            dctx16 = ly["o"].dgrad(dt16, dt)
            qkv = s["qkv"]
            dqkv16 = torch.empty((2, r, 3 * d), dtype=dt, device=dev)
            self._attn_bwd(dctx16.view(2 * r, d), *(self._heads(qkv.view(2 * r, 3 * d), 2 * t_n, l, j, 3) for j in range(3)), s["sa"],
                           *(self._heads(dqkv16.view(2 * r, 3 * d), 2 * t_n, l, j, 3) for j in range(3)))
            ly["qkv"].wgrad(s["h16"], dqkv16, wq, bias=True)
            dh = ly["qkv"].dgrad(dqkv16, torch.float32, residual=dpre1).view(2 * r, d)      # both branches stacked, as the layer below's FFN saw them
            T.wgrad_grouped(wq)
        # branch 1 entered through BertEmbeddings; branch 0 is z_t (frozen stage I)
        de = dh[r:] if ph <= 0 else T.eltwise(dh[r:], T.MODE_DROPOUT, p_drop=ph, seed=self._site(9000))
        dpre_e = self.ln_e.bwd(sv["pre_e"], de)
        T.embed_bwd(sv["ids"].view(-1), dpre_e, self.dword, self.dpos, l)
        # the loss-scaled gradient of the target tokens, unscaled for the ViT's own (separately scaled) reverse pass
        self.dfeats = None if dfeats is None else (dfeats if self.grad_scale == 1.0 else T.eltwise(dfeats, T.MODE_SCALE, p_drop=1.0 / self.grad_scale))
        self.dfeats_scale = None if dfeats is None else self.grad_scale            # train_vit.VitTrainer.backward runs under the same scale
        return self._collect()

    def _collect(self) -> Dict[str, torch.Tensor]:
        """{name: gradient}: views of the flat gradient buffer (unscaled in one launch when the pass ran on S * dlogits)."""
        slab = self.slab
        # fp16 operands: an intermediate gradient above 65504 turns into inf -> NaN in the weight gradients.  What GradScaler's unscale_ /
        # found_inf do for the reference (stage2_train.py:215-218) in ONE pass over the buffer: divide by S and note any non-finite element
        # in a device flag that AdamW.step consumes on the device (round 6; before: one scaling pass + the five passes of torch.isfinite)
        self.grads_finite = _unscale_and_check(slab.gflat, self.grad_scale) if self.dtype == torch.float16 else None
        slab.checked = None if self.grads_finite is None else (slab.gflat.data_ptr(), self.grads_finite, slab.gflat._version)     # (AdamW.step: this buffer is tested)
        if self.dtype != torch.float16 and self.grad_scale != 1.0:
            slab.gflat = T.eltwise(slab.gflat, T.MODE_SCALE, p_drop=1.0 / self.grad_scale)
        return {n: slab.grad(n) for n in slab.names}


def _unscale_and_check(gflat: torch.Tensor, grad_scale: float) -> torch.Tensor:
    """gflat /= grad_scale in place; returns a 0-dim bool tensor "all finite" (no host read)."""
    st = torch.zeros((8,), dtype=torch.int32, device=gflat.device)
    T.grads_check(gflat, st, 1.0 / grad_scale)
    return st[0] == 0


def _install_grads(tr, grads: Dict[str, torch.Tensor]):
    """Accumulate a trainer's gradients into `.grad` as autograd's AccumulateGrad would (a first gradient is the trainer's own slice of its
    flat gradient buffer - no copy).  Gradient accumulation over micro-batches (stage2_train.py's grad_accumulation_step): when every .grad
    is still a slice of the flat buffer a previous backward installed, ONE flat add folds it into the new buffer and .grad is re-pointed to
    the new slices - so the optimizer keeps its one-launch flat path (per-tensor adds: ~570 launches, and AdamW falls back to 570 more)."""
    slab = tr.slab
    live = [(n, slab.params[n]) for n in slab.names if n in grads and slab.params[n].requires_grad]
    prev = getattr(tr, "acc_gflat", None)
    if (prev is not None and prev is not slab.gflat and prev.numel() == slab.gflat.numel()
            and all(p.grad is not None and p.grad.data_ptr() == prev.data_ptr() + 4 * slab.off[n] and p.grad.is_contiguous() for n, p in live)):
        slab.gflat = T.eltwise(slab.gflat, T.MODE_ADD, prev)
        slab.checked = None                                   # (the sum is a buffer nobody has tested: AdamW.step tests it)
        for n, p in live:
            p.grad = slab.grad(n)
    else:
        for n, p in live:
            gq = grads[n]
            p.grad = gq if p.grad is None else T.eltwise(p.grad.contiguous(), T.MODE_ADD, gq.contiguous())
    tr.acc_gflat = slab.gflat


class _FusionTrainFn(torch.autograd.Function):
    """One autograd node around NlvrTrainer.forward / backward: `loss.backward()` of the reference's training step reaches
    the hand-written reverse pass through it.  `anchor` is a one-element leaf that only makes the node differentiable; the
    parameters' gradients are accumulated into `.grad` directly, as autograd's AccumulateGrad would (a first gradient is the
    trainer's own slice of its flat gradient buffer - no copy; later ones are added)."""

    @staticmethod
    def forward(ctx, anchor, trainer, z_t, feats, ids, mask):
        ctx.trainer = trainer
        ctx.feats_shape = tuple(feats.shape)
        out = trainer.forward(z_t, feats, ids, mask)
        # The saved activations, the dropout site counter and the flat gradient buffer are single slots on the trainer: this
        # node may only be differentiated while they still belong to ITS forward, and only once.
        trainer.generation = ctx.generation = getattr(trainer, "generation", 0) + 1
        trainer.consumed = False
        return out

    @staticmethod
    def backward(ctx, dlogits):
        tr = ctx.trainer
        if tr.generation != ctx.generation:
            raise RuntimeError("img_txt_fusion (train mode): another training-mode forward ran before this one's backward - the saved "
                               "activations belong to the later forward.  Call backward() after each forward (gradients accumulate "
                               "in .grad across steps), or run the other forward under torch.no_grad() / in .eval() mode")
        if tr.consumed:
            raise RuntimeError("img_txt_fusion (train mode): second backward through the same forward (retain_graph): the hand-written "
                               "reverse pass keeps one gradient buffer per forward; run the forward again")
        tr.consumed = True
        grads = tr.backward(dlogits.contiguous().float())
        _install_grads(tr, grads)
        dfeats = None if tr.dfeats is None else tr.dfeats.view(ctx.feats_shape)
        tr.dfeats = None
        return None, None, None, dfeats, None, None


def cosine_lr_schedule(optimizer, epoch: int, max_epoch: int, init_lr: float, min_lr: float) -> float:
    """utils.cosine_lr_schedule (utils.py:216-221): the per-epoch decay stage2_train.py:159 applies; works on `AdamW` below and on torch.optim."""
    lr = (init_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * epoch / max_epoch)) + min_lr
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr


def fusion_train(model, z_t, feats, ids, mask, p_hidden: float = 0.1, p_attn: float = 0.1, seed: int = 0) -> torch.Tensor:
    """(B, B) logits of `img_txt_fusion` in training mode, differentiable w.r.t. the model's text_encoder / cls_head parameters - and
    w.r.t. the target image tokens when they require a gradient (blip_img_tune, stage2_train.py:191-199: the tokens then come from
    `train_vit.vit_train`, whose reverse pass continues into the ViT)."""
    if torch.is_tensor(z_t) and z_t.requires_grad:
        raise NotImplementedError("z_t requires a gradient: the reference computes it from the frozen stage-I model under torch.no_grad() "
                                  "(stage2_train.py:201-203); the backward pass stops at the two-branch encoder's z_t input")
    tr = getattr(model, "_trainer", None)
    if tr is None or (tr.p_hidden, tr.p_attn) != (float(p_hidden), float(p_attn)) or tr.dtype != train_dtype(model):
        tr = model._trainer = NlvrTrainer(model, p_hidden, p_attn, seed)
        tr.anchor = torch.zeros((1,), device=z_t.device, requires_grad=True)
    tr.need_dfeats = bool(torch.is_tensor(feats) and feats.requires_grad)
    return _FusionTrainFn.apply(tr.anchor, tr, z_t, feats, ids, mask)


class AdamW:
    """torch.optim.AdamW's update rule on cir_adamw_step (stage2_train.py:138 builds that optimizer), fp32 master parameters.
    When the parameters and their gradients are the trainer's flat buffers (the normal case after `fusion_train`), one launch
    updates all of them; otherwise one launch per tensor."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, model=None, check_finite=None):
        """`model`: the BLIP_NLVR whose parameters these are - its packed inference engine is marked stale by every step()
        (without it the training forward marks it, which misses an eval call made between backward() and step()).
        `check_finite`: test the gradients step() is about to apply for inf / NaN and skip the update then (GradScaler.step's found_inf,
        stage2_train.py:215-218; one reduction over the flat gradient buffer + one host read).  None = automatic: always, unless `model`
        is given and its trainers run bf16 operands (whose pass cannot overflow: no loss scale)."""
        self.model = model
        self.check_finite = check_finite
        self.params = [p for p in params if p.requires_grad]
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self._state = None                                    # device: [found_inf, t, skipped, bc1, bc2, ...] (cir_adamw_begin)
        self._calls = 0
        self._plans: Dict[tuple, tuple] = {}                  # (param storage, first param) -> cached flat layout of a parameter group
        # torch.optim's surface as far as the reference's loop uses it: utils.cosine_lr_schedule (utils.py:216-221, called once per epoch at
        # stage2_train.py:159) writes `param_group['lr']`; one group, its 'lr' is what step() applies
        self.param_groups = [{"params": self.params, "lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]
        self.m: Dict[int, torch.Tensor] = {}
        self.v: Dict[int, torch.Tensor] = {}
        self._flats: Dict[int, tuple] = {}                    # param storage ptr -> (m flat, v flat)

    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value: float):
        self.param_groups[0]["lr"] = value

    @staticmethod
    def _flat_range(tensors):
        """(base pointer, elements) when `tensors` tile ONE storage completely in slices padded to 8 elements, else None."""
        st = tensors[0].untyped_storage()
        if any(t.untyped_storage().data_ptr() != st.data_ptr() or not t.is_contiguous() for t in tensors):
            return None
        if sum((t.numel() + 7) // 8 * 8 for t in tensors) * 4 != st.nbytes():
            return None
        return st.data_ptr(), st.nbytes() // 4

    # applied / skipped step counts live on the device (the skip decision is taken there): reading them is a host read
    @property
    def t(self) -> int:
        return 0 if self._state is None else int(self._state[1])

    @property
    def skipped_steps(self) -> int:
        return 0 if self._state is None else int(self._state[2])

    def _plan(self, grp):
        """Flat layout of a group of parameters that tile ONE fp32 storage (the trainer's slab): (base pointer, elements, per-parameter
        element offsets) - computed once per group; None when they do not tile one."""
        key = (grp[0].data.untyped_storage().data_ptr(), len(grp), id(grp[0]), id(grp[-1]))
        if key not in self._plans:
            fp = self._flat_range([p.data for p in grp])
            self._plans[key] = None if fp is None else (fp[0], fp[1], [(p.data_ptr() - fp[0]) // 4 for p in grp])
        return self._plans[key]

    @staticmethod
    def _grads_match(grp, plan):
        """The gradients of `grp` are slices of ONE flat buffer laid out like the parameters (what the trainers install): its base pointer."""
        g0 = grp[0].grad
        base = g0.data_ptr() - 4 * plan[2][0]
        if g0.untyped_storage().data_ptr() != base or g0.untyped_storage().nbytes() != 4 * plan[1]:
            return None
        for p, o in zip(grp, plan[2]):
            g = p.grad
            if g.data_ptr() != base + 4 * o or g.dtype != torch.float32 or not g.is_contiguous():
                return None
        return base

    @torch.no_grad()
    def step(self):
        """One AdamW step; with fp16 operands the update is skipped when a gradient is inf / NaN (GradScaler.step, stage2_train.py:215-218).
        Nothing here reads the device (round 6): the finite test ORs into a device flag, cir_adamw_begin turns it into the step count /
        bias corrections or the skip count, and the update kernels return at once under a set flag.  The test runs on the buffers this call
        APPLIES - .grad as it is now, after any accumulation over micro-batches - and does not depend on `model=`."""
        ps = [p for p in self.params if p.grad is not None]
        if not ps:
            return
        dev = ps[0].device
        if self._state is None:
            self._state = torch.zeros((8,), dtype=torch.int32, device=dev)
        st = self._state
        self._calls += 1
        st[0:1].zero_()
        need = self.check_finite
        trainers = [] if self.model is None else [tr for tr in (getattr(self.model, "_trainer", None), getattr(self.model, "_vit_trainer", None)) if tr is not None]
        if need is None:
            need = self.model is None or not trainers or any(getattr(tr, "dtype", None) == torch.float16 for tr in trainers)
        for tr in trainers:                                   # a flag a trainer's last backward (or a test / caller) set
            gf = getattr(tr, "grads_finite", None)
            if gf is not None:
                st[0:1] |= (~torch.as_tensor(gf, device=dev).reshape(1)).to(torch.int32)
        # one launch per FLAT STORAGE (the two-branch encoder's slab; the ViT's when it is fine-tuned), per tensor for what is left
        groups: Dict[int, list] = {}
        for p in ps:
            groups.setdefault(p.data.untyped_storage().data_ptr(), []).append(p)
        work = []                                             # (p flat, g flat, m, v, p16 or None, slab or None)
        for grp in groups.values():
            plan = self._plan(grp) if len(grp) > 1 else None
            gbase = self._grads_match(grp, plan) if plan is not None else None
            if gbase is None:
                for p in grp:
                    if id(p) not in self.m:
                        self.m[id(p)], self.v[id(p)] = torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32)
                    m, v = self.m[id(p)], self.v[id(p)]
                    if not (m.is_contiguous() and v.is_contiguous()):
                        m, v = self.m[id(p)], self.v[id(p)] = m.contiguous(), v.contiguous()
                    pd = p.data if p.data.is_contiguous() and p.data_ptr() % 16 == 0 else None       # (else: stepped through a copy)
                    g = p.grad.contiguous().float()
                    work.append((pd if pd is not None else p.data.contiguous().clone(), g if g.data_ptr() % 16 == 0 else g.clone(), m, v, None, None,
                                 None if pd is not None else p))
                continue
            n = plan[1]
            flat = self._flats.get(plan[0])
            if flat is None:
                mf, vf = (torch.zeros((n,), dtype=torch.float32, device=dev) for _ in range(2))
                for p, o in zip(grp, plan[2]):                # carry over moments from per-tensor steps, then keep views
                    for store, fl in ((self.m, mf), (self.v, vf)):
                        view = fl[o:o + p.numel()].view(p.shape)
                        if id(p) in store:
                            view.copy_(store[id(p)])
                        store[id(p)] = view
                flat = self._flats[plan[0]] = (mf, vf)
            pflat = torch.empty(0, dtype=torch.float32, device=dev).set_(grp[0].data.untyped_storage(), 0, (n,))
            gflat = torch.empty(0, dtype=torch.float32, device=dev).set_(grp[0].grad.untyped_storage(), 0, (n,))
            slab = _SLABS.get(plan[0])                        # the trainer's slab these parameters live in: its 16-bit copy is written along
            slab = slab() if slab is not None else None
            if slab is not None and (slab.flat32.data_ptr() != plan[0] or slab.flat16 is None or slab.flat16.numel() != n):
                slab = None
            work.append((pflat, gflat, flat[0], flat[1], None if slab is None else slab.flat16, slab, None))
        if need:
            for w in work:
                # the trainer's backward tested exactly this buffer and no torch op has written to it since (version counter of the buffer
                # and its views): its flag stands; anything else - accumulated sums, edited gradients, foreign buffers - is tested here
                ck = None if w[5] is None else w[5].checked
                if ck is not None and ck[0] == w[1].data_ptr() and w[5].gflat is not None and w[5].gflat.data_ptr() == ck[0] and w[5].gflat._version == ck[2]:
                    st[0:1] |= (~ck[1].reshape(1)).to(torch.int32)
                else:
                    T.grads_check(w[1], st)
        T.adamw_begin(st, self.betas)
        for pf, gf_, m, v, p16, slab, back in work:
            T.adamw_step_dev(pf, gf_, m, v, st, self.lr, self.betas, self.eps, self.wd, p16=p16)
            if back is not None:                              # (a non-contiguous parameter stepped through a contiguous copy)
                back.data.copy_(pf)
            if slab is not None:
                slab.mark_fresh16()                           # (a skipped step leaves both copies as they were: still consistent)
        if self.model is not None:
            self.model._text_stale = True                     # the weights change HERE: the next eval / score call repacks
            if getattr(self.model, "_vit_trainer", None) is not None:
                self.model._vit_stale = True

    def zero_grad(self):
        for p in self.params:
            p.grad = None
