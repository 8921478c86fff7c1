"""CPU-side checks of the C-ABI boundary: the library builds in-tree, loads, and exports exactly
the entry points include/cirrank.h declares (no compute calls here - there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cirrank.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cir_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from candidate_reranking_cir_amd import lib
    assert _declared() == sorted(lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    from candidate_reranking_cir_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    cdll = lib.load()
    for name in _declared():
        assert hasattr(cdll, name), name
    assert cdll.cir_version() == 15
    assert b"aligned" in cdll.cir_strerror(-3)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from candidate_reranking_cir_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU/PyTorch fallback"):
        lib.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from candidate_reranking_cir_amd import ops
    from candidate_reranking_cir_amd.lib import CirrankError
    with pytest.raises(CirrankError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(16, 64, dtype=torch.bfloat16))


def test_argument_validation_happens_before_any_launch():
    """Every entry point checks pointers, extents, alignment and dtype codes on the host and returns a negative CIR_E*
    code without touching the device - so the error contract of include/cirrank.h can be checked without a GPU
    (fake, never dereferenced addresses stand in for device pointers)."""
    from candidate_reranking_cir_amd import lib
    c = lib.load()
    EINVAL, ESHAPE, EALIGN, EDTYPE = -1, -2, -3, -4
    P = 0x10000          # 16-byte aligned fake device address
    BF16, F16, F32 = 0, 1, 2
    gemm = c.cir_gemm_bias_act
    ok_args = [P, 64, 0, P, 64, 0, None, 0, None, F32, 0, 0, P, 16, 0, 4, 16, 64, 1, 0, BF16, BF16, None]

    def call(**over):
        names = ["A", "lda", "sA", "W", "ldw", "sW", "bias", "sB", "res", "res_dtype", "ldr", "sR", "C", "ldc", "sC", "M", "N", "K", "batch",
                 "act", "in_dtype", "out_dtype", "stream"]
        a = dict(zip(names, ok_args))
        a.update(over)
        return gemm(*[a[n] for n in names])

    assert call(A=None) == EINVAL and call(C=None) == EINVAL and call(M=0) == EINVAL and call(batch=0) == EINVAL
    assert call(act=7) == EINVAL
    assert call(K=60) == ESHAPE and call(N=17) == ESHAPE
    assert call(A=P + 2) == EALIGN and call(lda=60) == EALIGN and call(bias=P + 4) == EALIGN and call(res=P + 8, ldr=16) == EALIGN
    assert call(in_dtype=F32) == EDTYPE and call(out_dtype=7) == EDTYPE
    # residual stream: an fp16 residual only as part of an fp16 C; a bf16 residual never
    assert call(res=P, res_dtype=F16, ldr=16, out_dtype=F32) == EDTYPE and call(res=P, res_dtype=BF16, ldr=16) == EDTYPE
    assert call(res=P, res_dtype=F16, ldr=12, out_dtype=F16) == EALIGN
    # bf16 operands + fp16 C is the residual stream: its epilogue reads an fp16 residual, an fp32 one is refused (not reinterpreted)
    assert call(res=P, res_dtype=F32, ldr=16, out_dtype=F16, in_dtype=BF16) == EDTYPE
    # LayerNorm folded into the GEMM (ABI v12): fp16 rows only, K-tile pairs, every vector present
    lnf = lambda **o: c.cir_gemm_ln_bias_act(*[{**dict(X=P, ldx=768, W=P, ldw=768, cs=P, b=P, C=P, ldc=768, M=4, N=768, K=768, eps=1e-6, act=0, dt=F16, st=None), **o}[k]
                                               for k in ("X", "ldx", "W", "ldw", "cs", "b", "C", "ldc", "M", "N", "K", "eps", "act", "dt", "st")])
    assert lnf(cs=None) == EINVAL and lnf(b=None) == EINVAL and lnf(eps=0.0) == EINVAL and lnf(act=2) == EINVAL      # no ReLU form
    assert lnf(dt=BF16) == EDTYPE and lnf(dt=F32) == EDTYPE
    assert lnf(K=64, ldx=64, ldw=64) == ESHAPE and lnf(N=48, ldc=48) == ESHAPE and lnf(ldx=772) == EALIGN and lnf(X=P + 2) == EALIGN
    # LayerNorm / attention / top-k: the same contract
    assert c.cir_layernorm(None, F32, 0, None, 0, P, P, 0, P, F32, None, 0, 4, 64, 1, 1e-6, BF16, None) == EINVAL
    assert c.cir_layernorm(P, F32, 0, None, 0, P, P, 0, None, F32, None, 0, 4, 64, 1, 1e-6, BF16, None) == EINVAL      # no output at all
    assert c.cir_layernorm(P, F32, 0, None, 0, P, P, 0, P, F32, None, 0, 4, 2048, 1, 1e-6, BF16, None) == ESHAPE       # cols > 1024
    assert c.cir_layernorm(P, F32, 0, None, 0, P, P, 0, P, F32, None, 0, 4, 64, 1, 1e-6, F32, None) == EDTYPE
    assert c.cir_layernorm(P, BF16, 0, None, 0, P, P, 0, P, F32, None, 0, 4, 64, 1, 1e-6, BF16, None) == EDTYPE        # stream is fp32 or fp16
    assert c.cir_layernorm(P, F16, 0, None, 0, P, P, 0, P, BF16, None, 0, 4, 64, 1, 1e-6, BF16, None) == EDTYPE
    assert c.cir_vit_assemble(P, P, P, P, BF16, 1, 4, 64, None) == EDTYPE and c.cir_vit_assemble(P, P, P, P, F16, 1, 4, 60, None) == ESHAPE
    att = [P, 64, 64, 64, P, 64, 64, 64, P, 64, 64, 64, None, 0, 0, None, P, 64, 64, 64, 1, 1, 1, 4, 4, 0.125, BF16, None]
    # (the valid argument list itself is never passed: it would launch)
    bad = list(att); bad[0] = None
    assert c.cir_attention(*bad) == EINVAL
    bad = list(att); bad[1] = 60
    assert c.cir_attention(*bad) == EALIGN
    bad = list(att); bad[26] = 7                  # (CIR_F32 is a valid operand type since ABI v11: the exact mode)
    assert c.cir_attention(*bad) == EDTYPE
    assert c.cir_topk_desc(P, P, 1, 9000, None) == ESHAPE and c.cir_topk_desc(None, P, 1, 8, None) == EINVAL
    clsx = c.cir_cls_cross_attention
    assert clsx(None, 0, None, P, P, 1, 4, 128, 0.125, BF16, None) == EINVAL and clsx(P, 0, None, P, P, 0, 4, 128, 0.125, BF16, None) == EINVAL
    assert clsx(P, 0, None, P, P, 1, 4, 1024, 0.125, BF16, None) == ESHAPE and clsx(P, 0, None, P, P, 1, 4, 192, 0.125, BF16, None) == ESHAPE
    assert clsx(P, 4, None, P, P, 1, 4, 128, 0.125, BF16, None) == EALIGN and clsx(P, 0, None, P + 2, P, 1, 4, 128, 0.125, BF16, None) == EALIGN
    assert clsx(P, 0, None, P, P, 1, 4, 128, 0.125, F32, None) == EDTYPE
    # training-mode operators: the same contract
    bmm = lambda **o: c.cir_bmm(*[{**dict(A=P, B=P, C=P, M=4, N=4, K=4, lda=4, ldb=4, ldc=4, ta=0, tb=0, nb1=1, nb2=1, sA1=0, sA2=0, sB1=0, sB2=0,
                                          sC1=0, sC2=0, alpha=1.0, acc=0, it=BF16, ot=F32, st=None), **o}[k]
                                  for k in ("A", "B", "C", "M", "N", "K", "lda", "ldb", "ldc", "ta", "tb", "nb1", "nb2", "sA1", "sA2", "sB1", "sB2", "sC1",
                                            "sC2", "alpha", "acc", "it", "ot", "st")])
    assert bmm(A=None) == EINVAL and bmm(K=0) == EINVAL and bmm(nb2=0) == EINVAL and bmm(nb1=300, nb2=300) == ESHAPE
    assert bmm(it=BF16, ot=F16) == EDTYPE and bmm(it=F32, ot=BF16) == EDTYPE and bmm(it=7) == EDTYPE
    assert c.cir_transpose16(None, P, 4, 4, 4, 4, 1, 0, 0, BF16, None) == EINVAL and c.cir_transpose16(P, P, 4, 4, 4, 4, 1, 0, 0, F32, None) == EDTYPE
    assert c.cir_softmax_dropout(P, 8, None, 0, 0, P, P, 8, 4, 8, 0.125, 1.0, 1, BF16, None) == EINVAL              # p_drop = 1
    assert c.cir_softmax_dropout(P, 8, None, 0, 0, P, P, 8, 4, 8, 0.125, 0.1, 1, F32, None) == EDTYPE
    assert c.cir_softmax_dropout_bwd(P, 8, None, 8, P, 8, 4, 8, 0.125, 0.1, 1, BF16, None) == EINVAL
    assert c.cir_layernorm_bwd(P, P, P, P, P, None, 4, 64, 1e-12, None) == EINVAL and c.cir_layernorm_bwd(P, P, P, P, P, P, 4, 2048, 1e-12, None) == ESHAPE
    assert c.cir_eltwise(P, F32, None, P, F32, 8, 1, 0.0, 0, None) == EINVAL                                          # GELU' without dy
    assert c.cir_eltwise(P, F32, None, P, F32, 8, 9, 0.0, 0, None) == EINVAL and c.cir_eltwise(P, 7, None, P, F32, 8, 0, 0.0, 0, None) == EDTYPE
    assert c.cir_colsum(P, 8, None, 4, 8, None) == EINVAL and c.cir_embed_bwd(P, P, P, None, 4, 2, 8, None) == EINVAL
    assert c.cir_adamw_step(P, P, P, None, 8, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, None) == EINVAL
    assert c.cir_adamw_step(P, P, P, P, 8, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0, None) == EINVAL                           # step counts from 1
    assert c.cir_grads_check(None, 8, 1.0, P, None) == EINVAL and c.cir_grads_check(P, 0, 1.0, P, None) == EINVAL
    assert c.cir_adamw_begin(None, 0.9, 0.999, None) == EINVAL
    assert c.cir_adamw_step_dev(P, P, P, P, 8, 1e-3, 0.9, 0.999, 1e-8, 0.01, None, None, 0, None) == EINVAL
    assert c.cir_adamw_step_dev(P, P, P, P, 0, 1e-3, 0.9, 0.999, 1e-8, 0.01, P, None, 0, None) == EINVAL
    # kernel-selection overrides: range-checked, default automatic, and the library reads no environment variables
    assert c.cir_set_tuning(7, 0) == EINVAL and c.cir_set_tuning(0, 64) == EINVAL and c.cir_set_tuning(2, 9000) == EINVAL
    assert c.cir_set_tuning(0, 128) == 0 and c.cir_set_tuning(0, 0) == 0 and c.cir_set_tuning(2, -1) == 0 and c.cir_set_tuning(2, 0) == 0
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms
    for code, word in ((EINVAL, b"null"), (ESHAPE, b"extent"), (EALIGN, b"aligned"), (EDTYPE, b"dtype")):
        assert word in c.cir_strerror(code).lower()
