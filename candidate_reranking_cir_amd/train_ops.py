"""Tensor-level wrappers of the training-mode entry points of libcirrank (include/cirrank.h, "Training-mode operators";
SURVEY section 8(f)-4).  Same rules as `ops`: CUDA tensors only, no fallback, work on torch's current stream."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import lib as _lib
from .ops import _DT, _need_cuda, _ptr, _stream

MODE_GELU, MODE_GELU_BWD, MODE_RELU, MODE_RELU_BWD, MODE_DROPOUT, MODE_ADD, MODE_SCALE = 0, 1, 2, 3, 4, 5, 6


def transpose16(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (R, C) or (B, R, C) 16-bit with unit last stride -> (…, C, R) contiguous."""
    _need_cuda(x, out)
    x3 = x if x.dim() == 3 else x.unsqueeze(0)
    b, r, c = x3.shape
    assert x3.stride(2) == 1 and x3.dtype in (torch.bfloat16, torch.float16)
    if out is None:
        out = torch.empty((b, c, r) if x.dim() == 3 else (c, r), dtype=x.dtype, device=x.device)
    o3 = out if out.dim() == 3 else out.unsqueeze(0)
    assert o3.shape == (b, c, r) and o3.stride(2) == 1
    _lib.check(_lib.load().cir_transpose16(x3.data_ptr(), o3.data_ptr(), r, c, x3.stride(1), o3.stride(1), b, x3.stride(0) if b > 1 else 0,
                                           o3.stride(0) if b > 1 else 0, _DT[x.dtype], _stream()), "cir_transpose16")
    return out


class TransposePlan:
    """The table of cir_transpose16_multi for matrices (offset, rows, cols) of one flat 16-bit buffer, uploaded once."""

    def __init__(self, entries, device):
        rows, tile = [], 0
        for off, r, c in entries:
            rows.append([int(off), int(r), int(c), tile])
            tile += ((r + 31) // 32) * ((c + 31) // 32)
        self.count, self.tiles = len(rows), tile
        self.table = torch.tensor(rows, dtype=torch.int64, device=device).contiguous()

    def run(self, src: torch.Tensor, dst: torch.Tensor):
        _need_cuda(src, dst)
        assert src.dtype == dst.dtype and src.dtype in (torch.bfloat16, torch.float16) and src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
        _lib.check(_lib.load().cir_transpose16_multi(src.data_ptr(), dst.data_ptr(), self.table.data_ptr(), self.count, self.tiles, _DT[src.dtype], _stream()),
                   "cir_transpose16_multi")


def bmm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False, out: Optional[torch.Tensor] = None,
        out_dtype: Optional[torch.dtype] = None, alpha: float = 1.0, accumulate=False) -> torch.Tensor:
    """out[z] = alpha * op(a[z]) @ op(b[z]) (+ out[z]); a, b, out are (B, ., .) or (B1, B2, ., .) views with unit last stride and
    any row / batch strides (0 = broadcast over that batch level).  16-bit operands run on the MFMA kernel.  accumulate: False / True
    (out += per batch item) / "atomic" (fp32 out, batch items may share an out through stride-0 views: split-K sums)."""
    _need_cuda(a, b, out)
    assert a.dim() == b.dim() and a.dim() in (3, 4) and a.shape[:-2] == b.shape[:-2] and a.stride(-1) == 1 and b.stride(-1) == 1 and a.dtype == b.dtype
    was3 = a.dim() == 3
    if was3:
        a, b = a.unsqueeze(1), b.unsqueeze(1)
        out4 = None if out is None else out.unsqueeze(1)
    else:
        out4 = out
    nb1, nb2 = a.shape[0], a.shape[1]
    m, k = (a.shape[3], a.shape[2]) if trans_a else (a.shape[2], a.shape[3])
    k2, n = (b.shape[3], b.shape[2]) if trans_b else (b.shape[2], b.shape[3])
    assert k == k2, (a.shape, b.shape, trans_a, trans_b)
    out_dtype = out_dtype or (out.dtype if out is not None else a.dtype)
    if out4 is None:
        out4 = torch.empty((nb1, nb2, m, n), dtype=out_dtype, device=a.device)
    assert out4.shape == (nb1, nb2, m, n) and out4.stride(3) == 1 and out4.dtype == out_dtype
    lib = _lib.load()
    step = max(1, 65535 // nb2)                                       # grid.z limit
    for z0 in range(0, nb1, step):
        z1 = min(nb1, z0 + step)
        _lib.check(lib.cir_bmm(a[z0:z1].data_ptr(), b[z0:z1].data_ptr(), out4[z0:z1].data_ptr(), m, n, k, a.stride(2), b.stride(2), out4.stride(2),
                               int(trans_a), int(trans_b), z1 - z0, nb2, a.stride(0), a.stride(1), b.stride(0), b.stride(1), out4.stride(0),
                               out4.stride(1), float(alpha), 2 if accumulate == "atomic" else int(bool(accumulate)), _DT[a.dtype], _DT[out_dtype], _stream()), "cir_bmm")
    return out4.squeeze(1) if was3 else out4


def wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, splits: int = 0) -> torch.Tensor:
    """dw (N, K) fp32 += dy^T x (cir_wgrad): dy (rows, N), x (rows, K) 16-bit with unit last stride, read as stored; N, K multiples of 128."""
    _need_cuda(dy, x, dw)
    rows, n = dy.shape
    k = x.shape[1]
    assert x.shape[0] == rows and dy.dtype == x.dtype and dy.stride(1) == 1 and x.stride(1) == 1
    assert dw.dtype == torch.float32 and dw.shape == (n, k) and dw.stride(1) == 1
    _lib.check(_lib.load().cir_wgrad(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0), rows, n, k, int(splits),
                                     _DT[dy.dtype], _stream()), "cir_wgrad")
    return dw


WGRAD_GROUP_MAX = 16


def wgrad_grouped(problems) -> None:
    """[(dy (rows, N), x (rows, K), dw (N, K) fp32), ...]: dw_i += dy_i^T x_i, up to 16 problems per launch (cir_wgrad_grouped) - the
    weight gradients of one encoder layer fill the chip together without row splits."""
    lib = _lib.load()
    for g0 in range(0, len(problems), WGRAD_GROUP_MAX):
        grp = problems[g0:g0 + WGRAD_GROUP_MAX]
        arr = (_lib.WgradDesc * len(grp))()
        for d, (dy, x, dw) in zip(arr, grp):
            _need_cuda(dy, x, dw)
            rows, n = dy.shape
            k = x.shape[1]
            assert x.shape[0] == rows and dy.dtype == x.dtype == grp[0][0].dtype and dy.stride(1) == 1 and x.stride(1) == 1
            assert dw.dtype == torch.float32 and dw.shape == (n, k) and dw.stride(1) == 1
            d.dy, d.ldy, d.x, d.ldx, d.dw, d.ldw, d.rows, d.N, d.K, d.splits = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), \
                dw.stride(0), rows, n, k, 0
        _lib.check(lib.cir_wgrad_grouped(ctypes.addressof(arr), len(grp), _DT[grp[0][0].dtype], _stream()), "cir_wgrad_grouped")


def softmax_dropout(s: torch.Tensor, mask: Optional[torch.Tensor], rows_per_mask: int, scale: float, p_drop: float, seed: int, dtype: torch.dtype,
                    cols: Optional[int] = None):
    """s fp32 (rows, ld) contiguous, the first `cols` (default ld) entries of a row are scores; mask fp32 (groups, cols) additive, one
    row per `rows_per_mask` consecutive rows of s, or None -> (P, dropout(P)) in `dtype`, same (rows, ld) layout."""
    _need_cuda(s, mask)
    rows, ld = s.shape
    cols = ld if cols is None else int(cols)
    assert s.dtype == torch.float32 and s.is_contiguous() and 0 < cols <= ld
    p = torch.empty((rows, ld), dtype=dtype, device=s.device)
    pd = torch.empty_like(p)
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.stride(-1) == 1 and mask.shape[-1] == cols
    _lib.check(_lib.load().cir_softmax_dropout(s.data_ptr(), ld, _ptr(mask), int(rows_per_mask), mask.stride(0) if mask is not None else 0,
                                               p.data_ptr(), pd.data_ptr(), ld, rows, cols, float(scale), float(p_drop), int(seed) & (2 ** 63 - 1),
                                               _DT[dtype], _stream()), "cir_softmax_dropout")
    return p, pd


def softmax_dropout_bwd(p: torch.Tensor, dpd: torch.Tensor, scale: float, p_drop: float, seed: int, cols: Optional[int] = None) -> torch.Tensor:
    """P 16-bit (rows, ld), dPd fp32 (rows, ld), scores in the first `cols` entries -> dS 16-bit (rows, ld)."""
    _need_cuda(p, dpd)
    rows, ld = p.shape
    cols = ld if cols is None else int(cols)
    assert p.is_contiguous() and dpd.is_contiguous() and dpd.dtype == torch.float32 and dpd.shape == p.shape and 0 < cols <= ld
    ds = torch.empty_like(p)
    _lib.check(_lib.load().cir_softmax_dropout_bwd(p.data_ptr(), ld, dpd.data_ptr(), ld, ds.data_ptr(), ld, rows, cols, float(scale), float(p_drop),
                                                   int(seed) & (2 ** 63 - 1), _DT[p.dtype], _stream()), "cir_softmax_dropout_bwd")
    return ds


def _hv(t: torch.Tensor):
    """(G, H, rows, 64) head view -> (pointer, group / head / row strides); unit last stride."""
    assert t.dim() == 4 and t.shape[3] == 64 and t.stride(3) == 1
    return t.data_ptr(), t.stride(0), t.stride(1), t.stride(2)


def attention_train_fwd(q4: torch.Tensor, k4: torch.Tensor, v4: torch.Tensor, mask: Optional[torch.Tensor], out4: torch.Tensor, scale: float,
                        p_drop: float, seed: int, out32: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused training attention (cir_attention_train_fwd): q4 / out4 (G, H, Lq, 64), k4 / v4 (G, H, Lk, 64) 16-bit head views, mask fp32
    (G, Lk) additive or None; writes out4, returns the log2-domain log-sum-exp (G, H, Lq) fp32 the backward needs."""
    _need_cuda(q4, k4, v4, mask, out4)
    g, h, lq, _ = q4.shape
    lk = k4.shape[2]
    assert k4.shape == v4.shape == (g, h, lk, 64) and out4.shape == q4.shape and q4.dtype == k4.dtype == v4.dtype == out4.dtype
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.shape == (g, lk) and mask.is_contiguous()
    lse = torch.empty((g, h, lq), dtype=torch.float32, device=q4.device)
    if out32 is not None:                       # fp32 twin of the context in out4's layout (element strides equal)
        assert out32.dtype == torch.float32 and out32.shape == out4.shape and out32.stride() == out4.stride()
    _lib.check(_lib.load().cir_attention_train_fwd(*_hv(q4), *_hv(k4), *_hv(v4), _ptr(mask), *_hv(out4), _ptr(out32), lse.data_ptr(), g, h, lq, lk, float(scale),
                                                   float(p_drop), int(seed) & (2 ** 63 - 1), _DT[q4.dtype], _stream()), "cir_attention_train_fwd")
    return lse


def attention_train_bwd(q4, k4, v4, mask, out4, dout4, lse, dq4, dk4, dv4, scale: float, p_drop: float, seed: int, out32=None):
    """Recomputing backward of `attention_train_fwd`: dout4 16-bit in out4's layout; dq4 / dk4 / dv4 head views (written), all fp32 or all in
    the operand type (then the projection's backward products read them as they are)."""
    _need_cuda(q4, k4, v4, mask, out4, dout4, lse, dq4, dk4, dv4)
    g, h, lq, _ = q4.shape
    lk = k4.shape[2]
    assert dout4.shape == out4.shape and dout4.stride() == out4.stride() and dout4.dtype == out4.dtype == q4.dtype
    assert lse.shape == (g, h, lq) and lse.is_contiguous() and lse.dtype == torch.float32
    assert dq4.shape == q4.shape and dk4.shape == k4.shape and dv4.shape == v4.shape and dq4.dtype == dk4.dtype == dv4.dtype and dq4.dtype in (torch.float32, q4.dtype)
    dsum = torch.empty((g, h, lq), dtype=torch.float32, device=q4.device)
    qo = _hv(out4)
    assert out32 is None or (out32.dtype == torch.float32 and out32.shape == out4.shape and out32.stride() == out4.stride())
    _lib.check(_lib.load().cir_attention_train_bwd(*_hv(q4), *_hv(k4), *_hv(v4), _ptr(mask), qo[0], dout4.data_ptr(), qo[1], qo[2], qo[3],
                                                   _ptr(out32), lse.data_ptr(),
                                                   dsum.data_ptr(), *_hv(dq4), *_hv(dk4), *_hv(dv4), _DT[dq4.dtype], g, h, lq, lk, float(scale), float(p_drop),
                                                   int(seed) & (2 ** 63 - 1), _DT[q4.dtype], _stream()), "cir_attention_train_bwd")


def layernorm_bwd(x: torch.Tensor, gamma: torch.Tensor, dy: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor, eps: float) -> torch.Tensor:
    """x, dy fp32 (rows, cols) contiguous; dgamma / dbeta fp32 (cols) are ACCUMULATED into -> dx fp32."""
    _need_cuda(x, gamma, dy, dgamma, dbeta)
    rows, cols = x.shape
    assert x.dtype == dy.dtype == gamma.dtype == dgamma.dtype == dbeta.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    _lib.check(_lib.load().cir_layernorm_bwd(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), rows, cols,
                                             float(eps), _stream()), "cir_layernorm_bwd")
    return dx


def residual_layernorm_train(t0: torch.Tensor, t1: Optional[torch.Tensor], residual: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                             dtype16: torch.dtype, alpha: float = 1.0, p_drop: float = 0.0, seed: int = 0, pre: Optional[torch.Tensor] = None,
                             y32: Optional[torch.Tensor] = None, y16: Optional[torch.Tensor] = None, want32: bool = True):
    """pre = dropout(alpha * (t0 + t1)) + residual; y = LayerNorm(pre) in one pass (cir_residual_layernorm_train).  fp32 (rows, cols)
    contiguous inputs -> (pre fp32, y fp32 or None, y 16-bit); outputs may be given (contiguous row ranges of larger buffers)."""
    _need_cuda(t0, t1, residual, gamma, beta, pre, y32, y16)
    rows, cols = t0.shape
    for t in (t0, t1, residual, pre, y32):
        assert t is None or (t.dtype == torch.float32 and t.shape == (rows, cols) and t.is_contiguous())
    pre = torch.empty_like(t0) if pre is None else pre
    if y32 is None and want32:
        y32 = torch.empty_like(t0)
    y16 = torch.empty((rows, cols), dtype=dtype16, device=t0.device) if y16 is None else y16
    assert y16.dtype == dtype16 and y16.shape == (rows, cols) and y16.is_contiguous()
    _lib.check(_lib.load().cir_residual_layernorm_train(t0.data_ptr(), _ptr(t1), residual.data_ptr(), gamma.data_ptr(), beta.data_ptr(), pre.data_ptr(),
                                                        _ptr(y32), y16.data_ptr(), rows, cols, float(eps), float(alpha), float(p_drop),
                                                        int(seed) & (2 ** 63 - 1), _DT[dtype16], _stream()), "cir_residual_layernorm_train")
    return pre, y32, y16


def layernorm_bwd_fused(x: torch.Tensor, gamma: torch.Tensor, dy: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor, eps: float,
                        dtype16: torch.dtype, t_add: Optional[torch.Tensor] = None, dbias: Optional[torch.Tensor] = None,
                        dbias2: Optional[torch.Tensor] = None, alpha: float = 1.0, p_drop: float = 0.0, seed: int = 0, want_dx: bool = True,
                        want_dt: bool = True, dx: Optional[torch.Tensor] = None, dt16: Optional[torch.Tensor] = None):
    """Adjoint of `residual_layernorm_train` (cir_layernorm_bwd_fused): x (the saved pre) / dy fp32 (rows, cols) -> (dx fp32: gradient of pre =
    of the residual; dt16: alpha * dropout'(dx + t_add) in `dtype16`, the dense branch's gradient, its column sums accumulated into dbias /
    dbias2).  dgamma / dbeta are accumulated."""
    _need_cuda(x, gamma, dy, dgamma, dbeta, t_add, dbias, dbias2, dx, dt16)
    rows, cols = x.shape
    for t in (x, dy, t_add, dx):
        assert t is None or (t.dtype == torch.float32 and t.shape == (rows, cols) and t.is_contiguous())
    if dx is None and want_dx:
        dx = torch.empty_like(x)
    if dt16 is None and want_dt:
        dt16 = torch.empty((rows, cols), dtype=dtype16, device=x.device)
    assert dt16 is None or (dt16.dtype == dtype16 and dt16.shape == (rows, cols) and dt16.is_contiguous())
    _lib.check(_lib.load().cir_layernorm_bwd_fused(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), _ptr(dx), dgamma.data_ptr(), dbeta.data_ptr(),
                                                   _ptr(t_add), _ptr(dt16), _ptr(dbias), _ptr(dbias2), rows, cols, float(eps), float(alpha),
                                                   float(p_drop), int(seed) & (2 ** 63 - 1), _DT[dtype16], _stream()), "cir_layernorm_bwd_fused")
    return dx, dt16


def colsum16(a: torch.Tensor, sums: torch.Tensor) -> torch.Tensor:
    """sums (cols) fp32 += column sums of the 16-bit rows a (rows, cols) (unit last stride, row stride a multiple of 8)."""
    _need_cuda(a, sums)
    assert a.dim() == 2 and a.stride(1) == 1 and sums.dtype == torch.float32 and sums.shape == (a.shape[1],) and sums.is_contiguous()
    _lib.check(_lib.load().cir_rows16_colsum(a.data_ptr(), a.stride(0), None, 0, None, 0, sums.data_ptr(), a.shape[0], a.shape[1], 0, _DT[a.dtype],
                                             _stream()), "cir_rows16_colsum")
    return sums


def gelu_bwd16(df: torch.Tensor, z: torch.Tensor, sums: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dz = df * gelu'(z) on 16-bit (rows, cols) tensors; sums (cols) fp32 += column sums of dz (the dense layer's bias gradient)."""
    _need_cuda(df, z, sums)
    assert df.dim() == 2 and df.shape == z.shape and df.dtype == z.dtype and df.stride(1) == 1 and z.stride(1) == 1
    assert sums is None or (sums.dtype == torch.float32 and sums.shape == (df.shape[1],) and sums.is_contiguous())
    out = torch.empty(df.shape, dtype=df.dtype, device=df.device)
    _lib.check(_lib.load().cir_rows16_colsum(df.data_ptr(), df.stride(0), z.data_ptr(), z.stride(0), out.data_ptr(), out.stride(0), _ptr(sums), df.shape[0],
                                             df.shape[1], 1, _DT[df.dtype], _stream()), "cir_rows16_colsum")
    return out


def rows_scale_add(a: Optional[torch.Tensor], b: torch.Tensor, scale: torch.Tensor, rows_per_group: int, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """out = a + scale[row // rows_per_group] * b (a None: the scaled b alone) - fp32 (rows, cols) inputs, scale fp32 (groups); DropPath."""
    _need_cuda(a, b, scale)
    rows, cols = b.shape
    assert b.dtype == scale.dtype == torch.float32 and b.is_contiguous() and scale.is_contiguous() and scale.numel() * rows_per_group == rows
    assert a is None or (a.dtype == torch.float32 and a.shape == b.shape and a.is_contiguous())
    out = torch.empty((rows, cols), dtype=out_dtype, device=b.device)
    _lib.check(_lib.load().cir_rows_scale_add(_ptr(a), b.data_ptr(), scale.data_ptr(), out.data_ptr(), rows, cols, int(rows_per_group), _DT[out_dtype],
                                              _stream()), "cir_rows_scale_add")
    return out


def eltwise(z: torch.Tensor, mode: int, dy: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, p_drop: float = 0.0,
            seed: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Elementwise modes of cir_eltwise on contiguous tensors (see the MODE_* constants)."""
    _need_cuda(z, dy, out)
    assert z.is_contiguous() and (dy is None or (dy.is_contiguous() and dy.dtype == torch.float32 and dy.numel() == z.numel()))
    if out is None:
        out = torch.empty(z.shape, dtype=out_dtype or z.dtype, device=z.device)
    assert out.is_contiguous() and out.numel() == z.numel()
    _lib.check(_lib.load().cir_eltwise(z.data_ptr(), _DT[z.dtype], _ptr(dy), out.data_ptr(), _DT[out.dtype], z.numel(), mode, float(p_drop),
                                       int(seed) & (2 ** 63 - 1), _stream()), "cir_eltwise")
    return out


def colsum(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out (cols) fp32 += column sums of x fp32 (rows, cols) (unit last stride)."""
    _need_cuda(x, out)
    assert x.dtype == out.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and out.shape == (x.shape[1],)
    _lib.check(_lib.load().cir_colsum(x.data_ptr(), x.stride(0), out.data_ptr(), x.shape[0], x.shape[1], _stream()), "cir_colsum")
    return out


def embed_bwd(ids: torch.Tensor, dy: torch.Tensor, dword: torch.Tensor, dpos: torch.Tensor, length: int):
    """ids (rows,) int64, dy fp32 (rows, cols): dword[ids[r]] += dy[r], dpos[r % length] += dy[r]."""
    _need_cuda(ids, dy, dword, dpos)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and dy.dtype == torch.float32 and dy.is_contiguous()
    _lib.check(_lib.load().cir_embed_bwd(ids.data_ptr(), dy.data_ptr(), dword.data_ptr(), dpos.data_ptr(), dy.shape[0], int(length), dy.shape[1], _stream()),
               "cir_embed_bwd")


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
               weight_decay: float = 0.01, step: int = 1):
    _need_cuda(p, g, m, v)
    assert all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel() for t in (p, g, m, v))
    _lib.check(_lib.load().cir_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                                          float(eps), float(weight_decay), int(step), _stream()), "cir_adamw_step")
    _lib.PARAM_EPOCH[0] += 1


def grads_check(g: torch.Tensor, state: torch.Tensor, scale: float = 1.0):
    """g *= scale (in place, when scale != 1) and state[0] |= any non-finite element - one pass, no host read (cir_grads_check)."""
    _need_cuda(g, state)
    assert g.dtype == torch.float32 and g.is_contiguous() and state.dtype == torch.int32 and state.numel() >= 8
    _lib.check(_lib.load().cir_grads_check(g.data_ptr(), g.numel(), float(scale), state.data_ptr(), _stream()), "cir_grads_check")


def adamw_begin(state: torch.Tensor, betas=(0.9, 0.999)):
    """After the step's checks: found_inf -> skipped += 1, else t += 1 and the bias corrections of the new t (cir_adamw_begin)."""
    _need_cuda(state)
    _lib.check(_lib.load().cir_adamw_begin(state.data_ptr(), float(betas[0]), float(betas[1]), _stream()), "cir_adamw_begin")


def adamw_step_dev(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, state: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                   weight_decay: float = 0.01, p16: torch.Tensor = None):
    """adamw_step with the step count / skip decision read from `state` on the device (cir_adamw_step_dev); `p16`: the 16-bit copy of the
    updated parameters, written in the same pass."""
    _need_cuda(p, g, m, v, state)
    assert all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel() for t in (p, g, m, v))
    assert p16 is None or (p16.is_contiguous() and p16.numel() == p.numel() and p16.dtype in (torch.float16, torch.bfloat16))
    _lib.check(_lib.load().cir_adamw_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                                              float(eps), float(weight_decay), state.data_ptr(), None if p16 is None else p16.data_ptr(),
                                              0 if p16 is None else _DT[p16.dtype], _stream()), "cir_adamw_step_dev")
    _lib.PARAM_EPOCH[0] += 1

