"""Attribution probe: the ViT alone on bf16 operands (the matrix pipe sustains 7 % more bf16 than fp16 under the power cap), its tokens handed
on in fp16, everything else as in the default mode - does the reference's rank order survive?  python tools/vit_bf16_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import precision_modes as P  # noqa: E402
from candidate_reranking_cir_amd.engine import VitEngine  # noqa: E402

BF, HF = torch.bfloat16, torch.float16


class Bf16Vit:
    """VitEngine on bf16 operands (fp16 residual stream) whose tokens leave in fp16."""

    def __init__(self, model):
        self.inner = VitEngine(model.state_dict(), model.vit_geometry, BF, model.device, stream_dtype=HF)
        self.geo, self.blocks, self.ln_fold = self.inner.geo, self.inner.blocks, 0

    def forward(self, image, want32=False, **kw):
        y32, y16 = self.inner.forward(image.to(BF) if image.dtype == HF else image, want32=want32)
        return y32, y16.to(HF)


_orig = P.apply


def apply(m, mode):
    probe = mode == "probe"
    m = _orig(m, "f16 | streams f16" if probe else mode)
    if probe:
        e = list(m.engines())
        i = 0 if isinstance(e[0], VitEngine) else 1
        e[i] = Bf16Vit(m)
        m._engines = tuple(e)
    return m


P.apply = apply
P.MODES["probe"] = P.MODES["f16 | streams f16"]
dev = torch.device("cuda")
for fx in ("rank224_wide_c100", "rank224_wide_c200", "rank224_wide_f50", "outlier224_wide"):
    for mode in ("f16 | streams f16", "probe", "mixed (ViT+cross bf16, text f16) | streams f16"):
        w = P.fixture_stats(fx, mode, dev)
        extra = f"  well-conditioned tau {w['well_tau']:.4f} top10 {w['well_top10']:.2f}" if "well_tau" in w else ""
        print(f"{fx:20s} {mode[:44]:44s} max|d| {w['max_abs']:.2e} exact {w['exact']:.3f} tau {w['tau']:.4f} top10 {w['top10']:.3f}{extra}", flush=True)
