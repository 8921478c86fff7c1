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
    assert cdll.cir_version() == 3
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
