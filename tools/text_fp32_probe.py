"""Attribution probe (not a shipped mode): text-side operands in fp32 (the exact mode's kernels) with the ViT and the cross-attention block
left in fp16 - how much of the reference's rank order would a higher-precision TEXT side alone recover?  python tools/text_fp32_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import precision_modes as P  # noqa: E402

F32, HF = torch.float32, torch.float16


def apply(m, mode):
    if mode == "probe":
        if hasattr(m, "set_text_stream32_from"):
            m.set_text_stream32_from(None)
        m.compute_dtype, m.image_dtype, m._engines = F32, HF, None         # past the all-or-nothing guard: text fp32, ViT + cross block fp16
        m._stream_dtype, m._vit_stream_dtype = None, None
        return m
    return _orig(m, mode)


_orig, P.apply = P.apply, apply
P.MODES["probe"] = (F32, HF, F32, "same")
dev = torch.device("cuda")
for fx in ("rank224_wide_c100", "rank224_wide_c200", "rank224_wide_f50", "outlier224_wide"):
    for mode in ("f16 | streams f16", "f16 | ViT stream f16, text f32", "probe", "exact (fp32 everywhere, f32-input MFMA)"):
        w = P.fixture_stats(fx, mode, dev)
        extra = f"  well-conditioned tau {w['well_tau']:.4f} top10 {w['well_top10']:.2f}" if "well_tau" in w else ""
        print(f"{fx:20s} {mode:44s} max|d| {w['max_abs']:.2e} exact {w['exact']:.3f} tau {w['tau']:.4f} top10 {w['top10']:.3f}{extra}", flush=True)
print("probe timing (64 x 105 from pixels, 3 steps):", round(P.timing("probe", dev), 1), "triplets/s", flush=True)
