"""ctypes binding of libcirrank.so (C ABI in include/cirrank.h).

The product path has no fallback: if the HIP library is missing or an entry point fails, the
caller gets an exception - never a silent PyTorch/CPU substitute.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_uint64, c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CIR_LIB", os.path.join(_HERE, "libcirrank.so"))   # CIR_LIB: A/B a second build

CIR_BF16, CIR_F16, CIR_F32 = 0, 1, 2
TUNE_GEMM_TILE, TUNE_GEMM_GROUP_W, TUNE_ATTN_SHARED_MAX = 0, 1, 2   # cir_set_tuning knobs (tests / A-B only)
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2

class WgradDesc(ctypes.Structure):
    """cir_wgrad_desc of include/cirrank.h (one problem of cir_wgrad_grouped)."""
    _fields_ = [("dy", c_void_p), ("ldy", c_int64), ("x", c_void_p), ("ldx", c_int64), ("dw", c_void_p), ("ldw", c_int64), ("rows", c_int64),
                ("N", c_int), ("K", c_int), ("splits", c_int)]


# name -> argtypes; mirrors include/cirrank.h declaration by declaration
SIGNATURES = {
    "cir_version": (c_int, []),
    "cir_strerror": (c_char_p, [c_int]),
    "cir_set_tuning": (c_int, [c_int, c_int]),
    "cir_gemm_bias_act": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                  c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                  c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cir_gemm_ln_bias_act": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_int64, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "cir_split16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "cir_split8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "cir_gemm_split8": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cir_layernorm_split8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                     c_int64, c_int, c_int, c_float, c_void_p]),
    "cir_layernorm": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                              c_int64, c_int64, c_int, c_int, c_float, c_int, c_void_p]),
    "cir_attention": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                              c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                              c_void_p, c_int64, c_int64, c_int64,
                              c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "cir_attention_split8": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                     c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "cir_cls_cross_attention": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "cir_cross_attention_folded": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                           c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "cir_embed_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                    c_int64, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "cir_patchify": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cir_vit_assemble": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cir_small_linear": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "cir_gather_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "cir_topk_desc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "cir_linear_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "cir_l2_normalize": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    # training-mode operators
    "cir_transpose16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int, c_int64, c_int64, c_int, c_void_p]),
    "cir_bmm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int,
                        c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_int, c_int, c_void_p]),
    "cir_softmax_dropout": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_float, c_float,
                                    c_uint64, c_int, c_void_p]),
    "cir_softmax_dropout_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_float, c_float, c_uint64, c_int,
                                        c_void_p]),
    "cir_attention_train_fwd": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                        c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_uint64,
                                        c_int, c_void_p]),
    "cir_attention_train_bwd": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                        c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                        c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_uint64, c_int, c_void_p]),
    "cir_residual_layernorm_train": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float,
                                             c_float, c_float, c_uint64, c_int, c_void_p]),
    "cir_layernorm_bwd_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                        c_int, c_float, c_float, c_float, c_uint64, c_int, c_void_p]),
    "cir_rows16_colsum": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "cir_transpose16_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "cir_wgrad_grouped": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "cir_wgrad": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "cir_rows_scale_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p]),
    "cir_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "cir_eltwise": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_uint64, c_void_p]),
    "cir_colsum": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "cir_embed_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "cir_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "cir_grads_check": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    "cir_adamw_begin": (c_int, [c_void_p, c_float, c_float, c_void_p]),
    "cir_adamw_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_int, c_void_p]),
}

_lib = None
PARAM_EPOCH = [0]     # moved by every cir_adamw_step launch (train_ops.adamw_step): packed inference engines of a trained model compare it


class CirrankError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libcirrank.so once; raise loudly (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the gfx950 kernels first "
            f"(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C candidate_reranking_cir_amd/csrc`). "
            f"There is no CPU/PyTorch fallback for this path.")
    # torch FIRST: its wheel bundles its own libamdhip64; a process that loads libcirrank.so before torch would bind the kernels to the
    # system HIP runtime and torch's tensors to the bundled one - two runtimes, and every launch fails with "no ROCm-capable device"
    # (seen with `g.build(); g.smoke()` in one process).  With torch loaded the library's HIP symbols resolve to the runtime torch uses.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    from . import ABI_VERSION
    if lib.cir_version() != ABI_VERSION:
        raise ImportError(f"libcirrank.so ABI {lib.cir_version()} != expected {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def set_tuning(knob: int, value: int):
    """Kernel-selection override (include/cirrank.h: cir_set_tuning); 0 restores the automatic choice."""
    check(load().cir_set_tuning(knob, value), "cir_set_tuning")


def check(code: int, what: str):
    if code != 0:
        msg = load().cir_strerror(code).decode()
        raise CirrankError(f"{what} failed: {msg} (code {code})")
