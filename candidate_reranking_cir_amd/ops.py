"""Torch-tensor front end of the C ABI (include/cirrank.h): argument marshalling only.

Every function enqueues HIP kernels from libcirrank.so on torch's current stream and returns
torch tensors that own the memory.  Nothing here computes with torch ops; there is no fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import lib as _lib
from .lib import ACT_GELU, ACT_NONE, ACT_RELU, CIR_BF16, CIR_F16, CIR_F32  # noqa: F401

_DT = {torch.bfloat16: CIR_BF16, torch.float16: CIR_F16, torch.float32: CIR_F32}


# bench.py sets these to lists to time every GEMM / attention launch with HIP events on the launch stream
PROFILE_GEMM = None
PROFILE_ATTN = None


def gemm_kernel_name(m: int, n: int, k: int, nb: int, has_residual: bool, act: int, out_dtype, in_dtype: torch.dtype, tile: int = 0,
                     res_dtype=None) -> str:
    """Which kernel instantiation cir_gemm_bias_act launches for a shape (mirror of the dispatch in csrc/gemm.hip; for
    reporting only - the library decides).  `out_dtype`: torch dtype (or True / False = fp32 / operand type)."""
    if in_dtype == torch.float32:
        return "cir::gemm_kernel<float,1>"
    if isinstance(out_dtype, bool):
        out_dtype = torch.float32 if out_dtype else in_dtype
    t = "__bf16" if in_dtype == torch.bfloat16 else "_Float16"
    stream16 = out_dtype == torch.float16 and (in_dtype != torch.float16 or (has_residual and res_dtype == torch.float16))
    kind = 1 if out_dtype == torch.float32 else (2 if stream16 else 0)
    nblk256 = -(-m // 256) * -(-n // 256) * nb
    use256 = n >= 256 and nblk256 >= 192
    can256 = not (has_residual and (act != ACT_NONE or kind == 0)) and k % 128 == 0
    if tile == 128:
        use256 = False
    elif tile == 256:
        use256 = True
    if use256 and can256:
        name = f"cir::gemm256_kernel<{t},{'true' if kind == 1 else 'false'},{'true' if has_residual else 'false'}"
        if kind == 2:       # fp16 residual-stream C: the activation is a template constant when there is none
            return name + (",_Float16,0>" if act == ACT_NONE else ",_Float16>")
        if kind == 0 and act in (ACT_NONE, ACT_GELU):
            return name + f",float,{act}>"
        return name + ">"
    return f"cir::gemm_kernel<{t},{kind}>"


def _stream() -> int:
    """Raw handle of torch's current stream on the current device (the C accessor: torch.cuda.current_stream().cuda_stream builds two
    Python objects per call - measurable at ~800 launches per training step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.CirrankError("cirrank ops need device tensors (no CPU fallback exists)")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class SplitOperand:
    """An fp32 matrix held as fp16 terms, rows [hi | lo | hi] (`split16`): the A operand of a 3-product GEMM.  Returned by `gemm` for a
    GELU output on the split path, so that the FFN's intermediate never exists in fp32; accepted by `gemm` as `a`."""
    dtype = torch.float32

    def __init__(self, cat: torch.Tensor, k: int):
        self.cat, self.k, self.shape, self.device = cat, k, cat.shape[:-1] + (k,), cat.device

    @property
    def hi(self):
        return self.cat[..., :self.k]

    @property
    def lo(self):
        return self.cat[..., self.k:2 * self.k]

    def dim(self):
        return self.cat.dim()


def split16(x: torch.Tensor, act: int = ACT_NONE) -> SplitOperand:
    """fp32 (..., K) with contiguous rows and ONE row stride -> SplitOperand with rows [fp16(y) | fp16(y - hi) | fp16(y)], y = act(x)."""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    k = x.shape[-1]
    x2 = x if x.dim() == 2 else x.reshape(-1, k) if x.is_contiguous() else None
    if x2 is None:                                   # (B, M, K) views with a uniform row stride (e.g. the CLS rows of a (.., L, D) tensor)
        if x.dim() == 3 and x.stride(0) == x.shape[1] * x.stride(1):
            x2 = x.as_strided((x.shape[0] * x.shape[1], k), (x.stride(1), 1))
        else:
            x2 = x.contiguous().view(-1, k)
    cat = torch.empty(x.shape[:-1] + (3 * k,), dtype=torch.float16, device=x.device)
    p = cat.data_ptr()
    _lib.check(_lib.load().cir_split16(x2.data_ptr(), x2.stride(0), p, p + 2 * k, p + 4 * k, 3 * k, x2.shape[0], k, act, _stream()), "cir_split16")
    return SplitOperand(cat, k)


def split_weight(w32: torch.Tensor) -> torch.Tensor:
    """Mark an fp32 weight (.., N, K) for the 3-product path: `gemm(a32, w32, ..)` then runs ONE fp16 GEMM of depth 3K,
    [a_hi | a_lo | a_hi] [w_hi | w_hi | w_lo]^T = a_hi w_hi^T + a_lo w_hi^T + a_hi w_lo^T, in one fp32 accumulator."""
    hi = w32.to(torch.float16)
    lo = (w32 - hi.float()).to(torch.float16)
    w32._split3 = torch.cat([hi, hi, lo], dim=-1).contiguous()
    return w32


def _gemm_split3(a, w_cat, bias, residual, act, out):
    """out = act(a w^T + bias) (+ residual) in fp32 from fp16 term pairs (see split_weight).  GELU output: returned as a SplitOperand
    (the exact-erf activation rides in the split pass that forms the next GEMM's operand)."""
    sa = a if isinstance(a, SplitOperand) else split16(a)
    assert act == ACT_NONE or residual is None, "activation and residual do not meet on this path"
    n_rec = len(PROFILE_GEMM) if PROFILE_GEMM is not None else 0
    s_ = gemm(sa.cat, w_cat, bias, residual=residual, out_dtype=torch.float32, out=out)
    if PROFILE_GEMM is not None and len(PROFILE_GEMM) > n_rec:      # record the USEFUL flops (2 m n k, not the 3 k deep launch) under a name of its own
        fl, e0, e1, nbytes, name = PROFILE_GEMM[-1]
        PROFILE_GEMM[-1] = (fl / 3.0, e0, e1, nbytes, name.replace("cir::", "cir::split3:"))
    if act == ACT_GELU:
        return split16(s_, ACT_GELU)
    if act == ACT_RELU:
        s_.relu_()
    return s_


class Split8Operand:
    """An fp32 matrix held as "split8" rows (include/cirrank.h: cir_split8): `.rows` is a uint8 tensor (..., 4K) =
    [K x fp16(y) | K x e4m3((y - hi) 2^12) | K x e4m3(hi)] - the A operand of `cir_gemm_split8`.  Produced by `split8`, by the LayerNorm /
    fp32-attention kernels' split outputs and by a split8 GEMM with a GELU (fc1 -> fc2: the FFN's intermediate never exists in fp32)."""
    dtype = torch.float32

    def __init__(self, rows: torch.Tensor, k: int):
        assert rows.dtype == torch.uint8 and rows.shape[-1] == 4 * k and rows.stride(-1) == 1
        self.rows, self.k, self.shape, self.device = rows, k, rows.shape[:-1] + (k,), rows.device

    def dim(self):
        return self.rows.dim()

    def view(self, *lead) -> "Split8Operand":
        """the same rows under other leading dimensions, e.g. (2, R, K) <-> (2 R, K)"""
        return Split8Operand(self.rows.view(*lead, 4 * self.k), self.k)

    def float(self) -> torch.Tensor:
        """hi + lo as fp32 (diagnostics / tests: what the GEMM's first two terms see of the matrix)."""
        k = self.k
        hi = self.rows[..., :2 * k].contiguous().view(torch.float16).float()
        lo = self.rows[..., 2 * k:3 * k].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -12
        return hi + lo


def split8(x: torch.Tensor, act: int = ACT_NONE, out: Optional[torch.Tensor] = None) -> Split8Operand:
    """fp32 (..., K) with contiguous rows and ONE row stride -> Split8Operand of act(x) (cir_split8)."""
    _need_cuda(x, out)
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    k = x.shape[-1]
    x2 = x if x.dim() == 2 else x.reshape(-1, k) if x.is_contiguous() else None
    if x2 is None:
        if x.dim() == 3 and x.stride(0) == x.shape[1] * x.stride(1):
            x2 = x.as_strided((x.shape[0] * x.shape[1], k), (x.stride(1), 1))
        else:
            x2 = x.contiguous().view(-1, k)
    if out is None:
        out = torch.empty(x.shape[:-1] + (4 * k,), dtype=torch.uint8, device=x.device)
    assert out.is_contiguous() and out.dtype == torch.uint8 and out.numel() == x2.shape[0] * 4 * k
    _lib.check(_lib.load().cir_split8(x2.data_ptr(), x2.stride(0), out.data_ptr(), 4 * k, x2.shape[0], k, act, _stream()), "cir_split8")
    return Split8Operand(out, k)


def _e4m3(x: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even e4m3 bytes of a clamped fp32 tensor (on the CPU when the device build lacks the cast)"""
    x = x.clamp(-448.0, 448.0)
    try:
        return x.to(torch.float8_e4m3fn).view(torch.uint8)
    except (RuntimeError, TypeError):
        return x.cpu().to(torch.float8_e4m3fn).view(torch.uint8).to(x.device)


def split_weight8(w32: torch.Tensor) -> torch.Tensor:
    """Mark an fp32 weight (.., N, K) for the split8 path: rows [W_hi fp16 | e4m3(W_hi 2^e1) | e4m3(W_lo 2^e2)] with per-tensor exponents
    that bring the largest |W_hi| / |W_lo| to [112, 224) (e4m3's top binades, clear of its 448 ceiling)."""
    import math
    hi = w32.to(torch.float16)
    lo = w32 - hi.float()
    e1 = int(math.floor(math.log2(224.0 / max(float(hi.float().abs().max()), 1e-30))))
    e2 = int(math.floor(math.log2(224.0 / max(float(lo.abs().max()), 1e-30))))
    e1, e2 = max(min(e1, 60), -60), max(min(e2, 60), -60)
    rows = torch.cat([hi.contiguous().view(torch.uint8), _e4m3(hi.float() * 2.0 ** e1), _e4m3(lo * 2.0 ** e2)], dim=-1).contiguous()
    w32._split8 = (rows, e1, e2)
    return w32


def _gemm_split8(a, wpack, bias, residual, act, out):
    """act(a w^T + bias) (+ residual) on split8 operands (cir_gemm_split8): fp32 out, or - with a GELU - split8 rows out."""
    rows_w, e1, e2 = wpack
    sa = a if isinstance(a, Split8Operand) else split8(a)
    _need_cuda(sa.rows, rows_w, bias, residual, out)
    k = sa.k
    batched = sa.rows.dim() == 3
    a3 = sa.rows if batched else sa.rows.unsqueeze(0)
    w3 = rows_w if rows_w.dim() == 3 else rows_w.unsqueeze(0)
    nb, m, _ = a3.shape
    n = w3.shape[1]
    assert w3.shape[0] == nb and w3.shape[2] == 4 * k and a3.stride(2) == 1 and w3.stride(2) == 1
    assert act == ACT_NONE or residual is None, "activation and residual do not meet on this path"
    out_split = act == ACT_GELU
    if out_split:
        assert out is None
        res = torch.empty((nb, m, 4 * n) if batched else (m, 4 * n), dtype=torch.uint8, device=a3.device)
        o3 = res if batched else res.unsqueeze(0)
        ldc, sc = o3.stride(1), o3.stride(0)
    else:
        if out is None:
            out = torch.empty((nb, m, n) if batched else (m, n), dtype=torch.float32, device=a3.device)
        res = out
        o3 = out if out.dim() == 3 else out.unsqueeze(0)
        assert o3.shape == (nb, m, n) and o3.stride(2) == 1 and o3.dtype == torch.float32
        ldc, sc = o3.stride(1), o3.stride(0)
    sb = 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.stride(-1) == 1
        sb = bias.stride(0) if bias.dim() == 2 else 0
    ldr = sr = 0
    if residual is not None:
        r3 = residual if residual.dim() == 3 else residual.unsqueeze(0)
        assert r3.shape == (nb, m, n) and r3.stride(2) == 1 and r3.dtype == torch.float32
        ldr, sr = r3.stride(1), r3.stride(0)
    if PROFILE_GEMM is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    code = _lib.load().cir_gemm_split8(a3.data_ptr(), a3.stride(1), a3.stride(0), w3.data_ptr(), w3.stride(1), w3.stride(0), _ptr(bias), sb,
                                       _ptr(residual), ldr, sr, o3.data_ptr(), ldc, sc, m, n, k, nb, act, int(out_split), e1, e2, _stream())
    if PROFILE_GEMM is not None:
        ev1.record()     # USEFUL flops (2 m n k): the two correction products are this path's overhead, not work done
        alg_bytes = nb * ((m * k + n * k) * 4 + m * n * 4 + (m * n * 4 if residual is not None else 0) + (n * 4 if bias is not None else 0))
        nblk256 = -(-m // 256) * -(-n // 256) * nb
        name = "cir::gemm256_kernel<split8%s>" % (",split8-out" if out_split else (",residual" if residual is not None else "")) \
            if (n >= 256 and nblk256 >= 192) else "cir::gemm_split8_kernel"
        PROFILE_GEMM.append((2.0 * nb * m * n * k, ev0, ev1, float(alg_bytes), name))
    _lib.check(code, "cir_gemm_split8")
    return Split8Operand(res, n) if out_split else res


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         act: int = ACT_NONE, out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(a @ w.T + bias) (+ residual).  a (M,K) or (B,M,K) 16-bit - or fp32 with fp32 w / residual / out: the "exact"
    mode on the f32-input MFMA - with contiguous rows (any row stride); w (N,K) / (B,N,K); bias fp32 (N) / (B,N); out in a.dtype (operand copy), fp32 or fp16 (residual
    stream, also from bf16 operands); residual shaped like out: fp32 (out in a.dtype or fp32) or fp16 (with an fp16 out; the only
    residual an fp16 out from bf16 operands takes)."""
    if isinstance(a, Split8Operand) or (not isinstance(a, SplitOperand) and a.dtype == torch.float32 and getattr(w, "_split8", None) is not None):
        assert out_dtype in (None, torch.float32) and getattr(w, "_split8", None) is not None
        return _gemm_split8(a, w._split8, bias, residual, act, out)
    if isinstance(a, SplitOperand) or (a.dtype == torch.float32 and getattr(w, "_split3", None) is not None):
        assert out_dtype in (None, torch.float32) and getattr(w, "_split3", None) is not None
        return _gemm_split3(a, w._split3, bias, residual, act, out)
    _need_cuda(a, w, bias, residual, out)
    batched = a.dim() == 3
    if not batched:
        a3, w3 = a.unsqueeze(0), w.unsqueeze(0)
    else:
        a3, w3 = a, w
    nb, m, k = a3.shape
    n = w3.shape[1]
    assert w3.shape[0] == nb and w3.shape[2] == k and a3.stride(2) == 1 and w3.stride(2) == 1
    assert w3.dtype == a3.dtype, "operands of one type (16-bit, or fp32 for the exact mode)"
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty((nb, m, n) if batched else (m, n), dtype=out_dtype, device=a.device)
    o3 = out if out.dim() == 3 else out.unsqueeze(0)
    assert o3.shape == (nb, m, n) and o3.stride(2) == 1 and o3.dtype == out_dtype
    sb = 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.stride(-1) == 1
        sb = bias.stride(0) if bias.dim() == 2 else 0
    ldr = sr = 0
    if residual is not None:
        r3 = residual if residual.dim() == 3 else residual.unsqueeze(0)
        assert r3.shape == (nb, m, n) and r3.stride(2) == 1
        assert r3.dtype == torch.float32 or (r3.dtype == torch.float16 and out_dtype == torch.float16), "residual: fp32, or fp16 with an fp16 out"
        assert not (out_dtype == torch.float16 and a.dtype != torch.float16 and r3.dtype != torch.float16), \
            "an fp16 out from bf16 operands (residual stream) takes an fp16 residual"
        ldr, sr = r3.stride(1), r3.stride(0)
    if PROFILE_GEMM is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    code = _lib.load().cir_gemm_bias_act(
        a3.data_ptr(), a3.stride(1), a3.stride(0), w3.data_ptr(), w3.stride(1), w3.stride(0),
        _ptr(bias), sb, _ptr(residual), _DT[residual.dtype] if residual is not None else CIR_F32, ldr, sr,
        o3.data_ptr(), o3.stride(1), o3.stride(0), m, n, k, nb, act, _DT[a.dtype], _DT[out_dtype], _stream())
    if PROFILE_GEMM is not None:
        ev1.record()
        alg_bytes = nb * ((m * k + n * k) * a.element_size() + m * n * o3.element_size()
                          + (m * n * residual.element_size() if residual is not None else 0) + (n * 4 if bias is not None else 0))
        PROFILE_GEMM.append((2.0 * nb * m * n * k, ev0, ev1, float(alg_bytes),
                             gemm_kernel_name(m, n, k, nb, residual is not None, act, out_dtype, a.dtype,
                                              res_dtype=residual.dtype if residual is not None else None)))
    _lib.check(code, "cir_gemm_bias_act")
    return out


def ln_fold_pack(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, dtype=torch.float16):
    """Fold a LayerNorm's affine into the Linear behind it (cir_gemm_ln_bias_act): fp32 W (N,K), b (N), gamma / beta (K) ->
    (Wg = W diag(gamma) in `dtype`, colsum of the ROUNDED Wg in fp32, b' = b + W beta in fp32)."""
    w32, g32 = w.detach().float(), gamma.detach().float()
    wg = (w32 * g32[None, :]).to(dtype).contiguous()
    colsum = wg.double().sum(dim=1).float().contiguous()
    bias = (b.detach().double() + w32.double() @ beta.detach().double()).float().contiguous()
    return wg, colsum, bias


def gemm_ln(x: torch.Tensor, wg: torch.Tensor, colsum: torch.Tensor, bias: torch.Tensor, eps: float, act: int = ACT_NONE,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(LayerNorm(x) @ W.T + b) from the RAW fp16 stream rows x (M,K) and the packed (Wg, colsum, b') of `ln_fold_pack`:
    statistics inside the GEMM, no LayerNorm pass (vit.py:107-109 + :72 / :36-37)."""
    _need_cuda(x, wg, colsum, bias, out)
    m, k = x.shape
    n = wg.shape[0]
    assert x.dtype == torch.float16 and wg.dtype == torch.float16 and x.stride(1) == 1 and wg.stride(1) == 1 and wg.shape[1] == k
    assert colsum.dtype == torch.float32 and bias.dtype == torch.float32 and colsum.shape == (n,) and bias.shape == (n,)
    if out is None:
        out = torch.empty((m, n), dtype=torch.float16, device=x.device)
    assert out.shape == (m, n) and out.dtype == torch.float16 and out.stride(1) == 1
    if PROFILE_GEMM is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    code = _lib.load().cir_gemm_ln_bias_act(x.data_ptr(), x.stride(0), wg.data_ptr(), wg.stride(0), colsum.data_ptr(), bias.data_ptr(),
                                            out.data_ptr(), out.stride(0), m, n, k, float(eps), act, CIR_F16, _stream())
    if PROFILE_GEMM is not None:
        ev1.record()
        PROFILE_GEMM.append((2.0 * m * n * k, ev0, ev1, float((m * k + n * k + m * n) * 2 + n * 8),
                             f"cir::gemm256_kernel<_Float16,false,false,float,{act},true>"))
    _lib.check(code, "cir_gemm_ln_bias_act")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None,
              out32: Optional[torch.Tensor] = None, out16: Optional[torch.Tensor] = None, want32: bool = True,
              dtype16: Optional[torch.dtype] = torch.bfloat16, stream_dtype: Optional[torch.dtype] = None):
    """LayerNorm over the last dim of x (rows, cols) or (B, rows, cols) [+ residual]; x / residual are residual-stream
    tensors (fp32 or fp16, same dtype).  Returns (y_stream or None, y16 or None): `y_stream` (kept name `out32`) is the
    copy that feeds the next residual, in `stream_dtype` (default: x.dtype); `y16` the 16-bit operand copy in `dtype16`.
    gamma/beta (cols) or (B, cols); x/residual may be batch-broadcast (stride 0) views.  `dtype16=None` skips the operand
    copy, `want32=False` the stream copy."""
    _need_cuda(x, gamma, beta, residual)
    x3 = x if x.dim() == 3 else x.unsqueeze(0)
    nb = max(x3.shape[0], gamma.shape[0] if gamma.dim() == 2 else 1, (residual.shape[0] if residual is not None and residual.dim() == 3 else 1))
    rows, cols = x3.shape[1], x3.shape[2]
    assert x3.dtype in (torch.float32, torch.float16) and x3.stride(2) == 1 and x3.stride(1) == cols
    stream_dtype = stream_dtype or (out32.dtype if out32 is not None else x3.dtype)
    assert stream_dtype in (torch.float32, torch.float16)
    shape = (nb, rows, cols) if (x.dim() == 3 or nb > 1) else (rows, cols)
    if out32 is None and want32:
        out32 = torch.empty(shape, dtype=stream_dtype, device=x.device)
    if out16 is None and dtype16 is not None:
        out16 = torch.empty(shape, dtype=dtype16, device=x.device)
    sx = x3.stride(0) if x3.shape[0] > 1 else 0
    sr = 0
    if residual is not None:
        r3 = residual if residual.dim() == 3 else residual.unsqueeze(0)
        assert r3.dtype == x3.dtype and r3.stride(2) == 1 and r3.stride(1) == cols
        sr = r3.stride(0) if r3.shape[0] > 1 else 0
    sg = gamma.stride(0) if gamma.dim() == 2 else 0
    d16 = _DT[out16.dtype] if out16 is not None else CIR_BF16
    code = _lib.load().cir_layernorm(x3.data_ptr(), _DT[x3.dtype], sx, _ptr(residual), sr, gamma.data_ptr(), beta.data_ptr(), sg,
                                     _ptr(out32), _DT[out32.dtype] if out32 is not None else CIR_F32, _ptr(out16), rows * cols, rows, cols, nb,
                                     float(eps), d16, _stream())
    _lib.check(code, "cir_layernorm")
    return out32, out16


def layernorm_split8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None,
                     want_stream: bool = True):
    """LayerNorm of fp32 stream rows -> (fp32 stream copy or None, Split8Operand of the same values): cir_layernorm_split8.  Shapes and
    broadcasting as `layernorm` (x / residual (rows, cols) or (B, rows, cols), gamma / beta (cols) or (B, cols))."""
    _need_cuda(x, gamma, beta, residual)
    x3 = x if x.dim() == 3 else x.unsqueeze(0)
    nb = max(x3.shape[0], gamma.shape[0] if gamma.dim() == 2 else 1, (residual.shape[0] if residual is not None and residual.dim() == 3 else 1))
    rows, cols = x3.shape[1], x3.shape[2]
    assert x3.dtype == torch.float32 and x3.stride(2) == 1 and x3.stride(1) == cols
    shape = (nb, rows) if (x.dim() == 3 or nb > 1) else (rows,)
    ys = torch.empty(shape + (cols,), dtype=torch.float32, device=x.device) if want_stream else None
    sp = torch.empty(shape + (4 * cols,), dtype=torch.uint8, device=x.device)
    sx = x3.stride(0) if x3.shape[0] > 1 else 0
    sr = 0
    if residual is not None:
        r3 = residual if residual.dim() == 3 else residual.unsqueeze(0)
        assert r3.dtype == torch.float32 and r3.stride(2) == 1 and r3.stride(1) == cols
        sr = r3.stride(0) if r3.shape[0] > 1 else 0
    sg = gamma.stride(0) if gamma.dim() == 2 else 0
    code = _lib.load().cir_layernorm_split8(x3.data_ptr(), sx, _ptr(residual), sr, gamma.data_ptr(), beta.data_ptr(), sg, _ptr(ys), rows * cols,
                                            sp.data_ptr(), 4 * cols, rows * 4 * cols, rows, cols, nb, float(eps), _stream())
    _lib.check(code, "cir_layernorm_split8")
    return ys, Split8Operand(sp, cols)


def attention_split8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, mask: Optional[torch.Tensor] = None) -> Split8Operand:
    """fp32 attention (the `attention` contract with fp32 tensors) whose context comes out as split8 rows: (B1, B0, Lq, H*64) Split8Operand."""
    _need_cuda(q, k, v, mask)
    b1, b0, lq, d = q.shape
    lk = k.shape[2]
    assert k.shape == (b1, b0, lk, d) and v.shape == k.shape and d % 64 == 0
    assert q.dtype == k.dtype == v.dtype == torch.float32 and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    out = torch.empty((b1, b0, lq, 4 * d), dtype=torch.uint8, device=q.device)
    ms1 = ms0 = 0
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.shape == (b1, b0, lk) and mask.stride(2) == 1
        ms1, ms0 = mask.stride(0), mask.stride(1)
    if PROFILE_ATTN is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    code = _lib.load().cir_attention_split8(
        q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
        v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), _ptr(mask), ms1, ms0,
        out.data_ptr(), out.stride(0), out.stride(1), out.stride(2), b1, b0, d // 64, lq, lk, float(scale), _stream())
    if PROFILE_ATTN is not None:
        ev1.record()
        PROFILE_ATTN.append((4.0 * b1 * b0 * lq * lk * d, ev0, ev1, (lq, lk)))
    _lib.check(code, "cir_attention_split8")
    return Split8Operand(out, d)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, scale: float,
              mask: Optional[torch.Tensor] = None, kv_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q/out: (B1, B0, Lq, H*64) views, k/v: (B1, B0, Lk, H*64) views (any strides, unit last stride);
    mask: additive fp32 (B1, B0, Lk) view or None.  With `kv_index` (B1,) int64, k/v are banks
    (rows, B0, Lk, H*64) and item b1 attends to bank row kv_index[b1].  Writes `out` and returns it."""
    _need_cuda(q, k, v, out, mask, kv_index)
    b1, b0, lq, d = q.shape
    lk = k.shape[2]
    if kv_index is None:
        assert k.shape == (b1, b0, lk, d)
    else:
        assert kv_index.dtype == torch.int64 and kv_index.shape == (b1,) and kv_index.is_contiguous() and k.shape[1:] == (b0, lk, d)
    assert d % 64 == 0 and v.shape == k.shape and out.shape == q.shape
    assert q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1 and out.stride(3) == 1
    assert q.dtype == k.dtype == v.dtype == out.dtype
    ms1 = ms0 = 0
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.shape == (b1, b0, lk) and mask.stride(2) == 1
        ms1, ms0 = mask.stride(0), mask.stride(1)
    if PROFILE_ATTN is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    code = _lib.load().cir_attention(
        q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), k.data_ptr(), k.stride(0), k.stride(1), k.stride(2),
        v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), _ptr(mask), ms1, ms0, _ptr(kv_index),
        out.data_ptr(), out.stride(0), out.stride(1), out.stride(2), b1, b0, d // 64, lq, lk, float(scale), _DT[q.dtype], _stream())
    if PROFILE_ATTN is not None:
        ev1.record()
        PROFILE_ATTN.append((4.0 * b1 * b0 * lq * lk * d, ev0, ev1, (lq, lk)))
    _lib.check(code, "cir_attention")
    return out


def cls_cross_attention(x: torch.Tensor, qp: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None,
                        x_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[t, r] = sum_j softmax_j(qp[t, r] . x[t, j] * scale) x[t, j]: x (T, Lk, D) 16-bit tokens (rows contiguous), qp (T, 32, D)
    contiguous (every row finite) -> out (T, 32, D).  With `x_index` (T,) int64, x is a bank (rows, Lk, D) and item t attends
    bank row x_index[t].  The K / V projections live in the caller's two small GEMMs (include/cirrank.h: cir_cls_cross_attention)."""
    _need_cuda(x, qp, out, x_index)
    lk, d = x.shape[1], x.shape[2]
    t_n = qp.shape[0]
    assert x_index is not None or x.shape[0] == t_n
    assert x_index is None or (x_index.dtype == torch.int64 and x_index.shape == (t_n,) and x_index.is_contiguous())
    assert x.stride(2) == 1 and x.stride(1) == d and qp.shape == (t_n, 32, d) and qp.is_contiguous() and qp.dtype == x.dtype
    if out is None:
        out = torch.empty((t_n, 32, d), dtype=x.dtype, device=x.device)
    assert out.shape == (t_n, 32, d) and out.is_contiguous() and out.dtype == x.dtype
    _lib.check(_lib.load().cir_cls_cross_attention(x.data_ptr(), x.stride(0), _ptr(x_index), qp.data_ptr(), out.data_ptr(), t_n, lk, d,
                                                   float(scale), _DT[x.dtype], _stream()), "cir_cls_cross_attention")
    return out


def fold_pack_key(wk: torch.Tensor) -> torch.Tensor:
    """key.weight (B, D, D) [(h, d) rows, feature columns] -> the MFMA-fragment order cir_cross_attention_folded reads (include/cirrank.h):
    (B, unit = D/32, fbh 2, head 12, ks 2, lane 64, 8): lane (g = lane >> 4, i = lane & 15) of block (unit, fbh, head, ks) holds
    W_k[head 64 + 32 ks + 8 g + j][32 unit + 8 (i >> 2) + 4 fbh + (i & 3)], j < 8 - one contiguous KiB per wave-load."""
    b, d, _ = wk.shape
    h = d // 64
    w = wk.view(b, h, 2, 4, 8, d // 32, 4, 2, 4)          # [b][head][ks][g][j][unit][i>>2][fbh][i&3]
    return w.permute(0, 5, 7, 1, 2, 3, 6, 8, 4).contiguous().view(b, d, d)      # [b][unit][fbh][head][ks][g][i>>2][i&3][j]


def fold_pack_value(wv: torch.Tensor) -> torch.Tensor:
    """value.weight (B, D, D) -> (B, unit = D/32, db 4, head 12, lane 64, 8): lane (g, i) of block (unit, db, head) holds
    W_v[head 64 + 16 db + i][32 unit + (j < 4 ? 4 g + j : 16 + 4 g + j - 4)] - the k-slot order in which two 16x16 accumulator tiles of
    P X become one MFMA B operand."""
    b, d, _ = wv.shape
    h = d // 64
    w = wv.view(b, h, 4, 16, d // 32, 2, 4, 4)            # [b][head][db][i][unit][half: features 0-15 / 16-31][g][r]
    return w.permute(0, 4, 2, 1, 6, 3, 5, 7).contiguous().view(b, d, d)          # [b][unit][db][head][g][i][half][r]: j = 4 half + r


def cross_attention_folded(q: torch.Tensor, x: torch.Tensor, wkt: torch.Tensor, wvp: torch.Tensor, bv: torch.Tensor, out: torch.Tensor, l: int,
                           scale: float, heads: int = 12, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Folded two-branch cross-attention (cir_cross_attention_folded): q (2, T*L, D) cross-query projection, x (T, N, D) tokens, wkt (2, D, D) =
    fold_pack_key(key.weight), wvp (2, D, D) = fold_pack_value(value.weight), bv (2, D) fp32 -> out (T, L, 2, D) view (written, returned)."""
    _need_cuda(q, x, wkt, wvp, bv, out)
    t_n, n, d = x.shape
    assert q.shape == (2, t_n * l, d) and q.stride(2) == 1 and x.stride(2) == 1 and x.stride(1) == d
    assert wkt.shape == (2, d, d) and wvp.shape == (2, d, d) and wkt.is_contiguous() and wvp.is_contiguous() and bv.shape == (2, d) and bv.is_contiguous()
    assert out.shape == (t_n, l, 2, d) and out.stride(3) == 1 and q.dtype == x.dtype == wkt.dtype == wvp.dtype == out.dtype and bv.dtype == torch.float32
    if mask is not None:       # additive fp32 key mask (T, N), shared by the two branches
        _need_cuda(mask)
        assert mask.dtype == torch.float32 and mask.shape == (t_n, n) and mask.stride(1) == 1
    if PROFILE_ATTN is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    code = _lib.load().cir_cross_attention_folded(q.data_ptr(), q.stride(0), q.stride(1), x.data_ptr(), x.stride(0), wkt.data_ptr(), wvp.data_ptr(), d * d,
                                                  bv.data_ptr(), _ptr(mask), mask.stride(0) if mask is not None else 0,
                                                  out.data_ptr(), out.stride(0), out.stride(1), out.stride(2), t_n, l, n, d, heads,
                                                  float(scale), _DT[x.dtype], _stream())
    if PROFILE_ATTN is not None:
        ev1.record()      # executed flops: per (candidate, branch) 2 x (H L x 64 x D) projections + 2 x (H L x D x N) products
        PROFILE_ATTN.append((2.0 * t_n * 2 * (2 * heads * l * 64 * d + 2 * heads * l * d * n), ev0, ev1, ("folded", l, n)))
    _lib.check(code, "cir_cross_attention_folded")
    return out


def embed_layernorm(ids: torch.Tensor, word: torch.Tensor, pos: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                    eps: float, dtype16: torch.dtype = torch.bfloat16, stream_dtype: torch.dtype = torch.float32):
    """BERT embeddings: LayerNorm(word[ids] + pos[:L]); ids (R, L) int64 -> (y_stream, y16) of shape (R, L, cols)."""
    _need_cuda(ids, word, pos, gamma, beta)
    r, l = ids.shape
    cols = word.shape[1]
    if l > pos.shape[0]:      # the reference raises too (position_ids[:, :L] indexes a (max_position_embeddings,) table)
        raise IndexError(f"caption of {l} tokens exceeds the {pos.shape[0]}-row position-embedding table")
    ids = ids.contiguous()
    y32 = torch.empty((r, l, cols), dtype=stream_dtype, device=ids.device)
    if dtype16 == torch.float32:       # "exact" mode: the fp32 stream copy is the operand
        assert stream_dtype == torch.float32
        y16 = None
    else:
        y16 = torch.empty((r, l, cols), dtype=dtype16, device=ids.device)
    code = _lib.load().cir_embed_layernorm(ids.data_ptr(), word.data_ptr(), pos.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                           y32.data_ptr(), _DT[stream_dtype], _ptr(y16), r * l, l, cols, word.shape[0], float(eps),
                                           CIR_F16 if y16 is None else _DT[dtype16], _stream())
    _lib.check(code, "cir_embed_layernorm")
    return y32, (y32 if y16 is None else y16)


def patchify(image: torch.Tensor, patch: int, dtype16: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """(B,C,H,W) fp32 or 16-bit -> (B*gh*gw, C*patch*patch) 16-bit (fp32 with dtype16 = torch.float32), column order (c, ky, kx)."""
    _need_cuda(image)
    image = image.contiguous()
    b, c, h, w = image.shape
    out = torch.empty((b * (h // patch) * (w // patch), c * patch * patch), dtype=dtype16, device=image.device)
    code = _lib.load().cir_patchify(image.data_ptr(), _DT[image.dtype], out.data_ptr(), _DT[dtype16], b, c, h, w, patch, _stream())
    _lib.check(code, "cir_patchify")
    return out


def vit_assemble(proj: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, batch: int) -> torch.Tensor:
    """proj (B*P, D) in the stream dtype (fp32 / fp16), cls (D), pos (P+1, D) fp32 -> x (B, P+1, D) in proj.dtype."""
    _need_cuda(proj, cls, pos)
    assert proj.dtype in (torch.float32, torch.float16) and proj.is_contiguous()
    p = proj.shape[0] // batch
    d = proj.shape[1]
    x = torch.empty((batch, p + 1, d), dtype=proj.dtype, device=proj.device)
    code = _lib.load().cir_vit_assemble(proj.data_ptr(), cls.data_ptr(), pos.data_ptr(), x.data_ptr(), _DT[proj.dtype], batch, p, d, _stream())
    _lib.check(code, "cir_vit_assemble")
    return x


def small_linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """(M,K) 16-bit @ (N<=8, K)^T + bias -> (M,N) fp32."""
    _need_cuda(x, w, bias)
    if x.dtype == torch.float32:       # "exact" mode: the fp32 dot-product kernel
        return linear_f32(x, w, bias)
    m, k = x.shape
    n = w.shape[0]
    assert x.stride(1) == 1 and w.is_contiguous()
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    code = _lib.load().cir_small_linear(x.data_ptr(), x.stride(0), w.data_ptr(), _ptr(bias), y.data_ptr(), m, n, k, _DT[x.dtype], _stream())
    _lib.check(code, "cir_small_linear")
    return y


def argsort_desc(logits: torch.Tensor) -> torch.Tensor:
    """Row-wise descending argsort of fp32 (Q,K), ties -> lower index first (int64)."""
    _need_cuda(logits)
    logits = logits.contiguous()
    q, k = logits.shape
    idx = torch.empty((q, k), dtype=torch.int64, device=logits.device)
    code = _lib.load().cir_topk_desc(logits.data_ptr(), idx.data_ptr(), q, k, _stream())
    _lib.check(code, "cir_topk_desc")
    return idx


def gather_rows(src: torch.Tensor, index: Optional[torch.Tensor], dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """dst[i] = src[index[i]] converted to `dtype`; rows are src.shape[1:] flattened (index None = plain convert)."""
    _need_cuda(src, index)
    src = src.contiguous()
    dtype = dtype or src.dtype
    row_elems = src[0].numel()
    n = src.shape[0] if index is None else index.numel()
    dst = torch.empty((n,) + tuple(src.shape[1:]), dtype=dtype, device=src.device)
    if index is not None:
        index = index.to(torch.int64).contiguous()
    code = _lib.load().cir_gather_rows(src.data_ptr(), _DT[src.dtype], _ptr(index), dst.data_ptr(), _DT[dtype], n, row_elems,
                                       src.shape[0], _stream())
    _lib.check(code, "cir_gather_rows")
    return dst


def linear_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, mode: int = 0) -> torch.Tensor:
    """fp32 y = x @ w.T + bias (mode 0), 1 - x @ w.T (mode 1: cosine distance) or x @ w.T - 1 (mode 2: its exact
    negative); x rows may be strided."""
    _need_cuda(x, w, bias)
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and x.stride(1) == 1 and w.is_contiguous()
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    code = _lib.load().cir_linear_f32(x.data_ptr(), x.stride(0), w.data_ptr(), _ptr(bias), y.data_ptr(), m, n, k, mode, _stream())
    _lib.check(code, "cir_linear_f32")
    return y


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    code = _lib.load().cir_l2_normalize(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], _stream())
    _lib.check(code, "cir_l2_normalize")
    return y
