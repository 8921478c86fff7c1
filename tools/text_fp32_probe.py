"""Attribution probe behind the text32 mode: how much of the reference's rank order does a higher-precision TEXT side alone recover?
Rows: the all-fp16 default, the split-stream mode, text32 with its text-side GEMMs on the f32-input MFMA (`text_split3 = False`, the probe's
first form), text32 as shipped (3-product fp16 GEMMs), the exact mode.  python tools/text_fp32_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import precision_modes as P  # noqa: E402

TEXT32 = next(m for m in P.MODES if m.startswith("text32"))
_orig = P.apply


def apply(m, mode):
    if mode == "text32 on the f32-input MFMA":
        m.text_split3 = False
        return _orig(m, TEXT32)
    m.text_split3 = True
    return _orig(m, mode)


P.apply = apply
P.MODES["text32 on the f32-input MFMA"] = P.MODES[TEXT32]
dev = torch.device("cuda")
ROWS = ("f16 | streams f16", "f16 | ViT stream f16, text f32", "text32 on the f32-input MFMA", TEXT32, "exact (fp32 everywhere, f32-input MFMA)")
for fx in ("rank224_wide_c100", "rank224_wide_c200", "rank224_wide_f50", "outlier224_wide"):
    for mode in ROWS:
        w = P.fixture_stats(fx, mode, dev)
        extra = f"  well-conditioned tau {w['well_tau']:.4f} top10 {w['well_top10']:.2f}" if "well_tau" in w else ""
        print(f"{fx:20s} {mode[:44]:44s} max|d| {w['max_abs']:.2e} exact {w['exact']:.3f} tau {w['tau']:.4f} top10 {w['top10']:.3f}{extra}", flush=True)
for mode in ROWS[2:4]:
    print(f"{mode[:44]:44s} {P.timing(mode, dev):8.1f} triplets/s (64 x 105 from pixels, 3 steps)", flush=True)
