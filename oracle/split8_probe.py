"""Which arithmetic does the text side of the "text32" mode need?  CPU emulation on the WIDE rank fixtures (TEST INFRASTRUCTURE).

`attribute_rounding.py`'s emulated path (a rounding at every site the HIP path stores or consumes a value) with the text-side Linears
replaced by explicit multi-product forms, measured against the reference's own fp32 logits of tests/golden/rank224_wide.npz (c100)
and tests/golden/outlier224_wide.npz:

    x3     : A_hi W_hi + A_lo W_hi + A_hi W_lo, every term fp16 (the 3-product path of round 5)
    x1f8   : A_hi W_hi in fp16, the two correction products with BOTH factors rounded to fp8 e4m3 (power-of-two scaled, saturating) -
             the scaled-MFMA path of round 6 (v_mfma_scale_f32_16x16x128_f8f6f4 at twice the fp16 rate)
    x1bf8  : the same with e5m2 factors
    x2a/x2w: two fp16 products (A_lo W_hi or A_hi W_lo dropped)
    <v>b   : variant <v> with the LayerNorm offset beta of the producing LayerNorm kept out of the operand (folded into the bias)
    x1     : fp16 operands, fp32 stream (round 4's "split" setting)

    python oracle/split8_probe.py [rank|outlier] [variant ...] [q=N]   -> one line per variant (+ profiles/r6_split8_probe_<fixture>.json)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from candidate_reranking_cir_amd import config, synthetic, weights
from oracle import attribute_rounding as AR

HF = torch.float16
ROOT = AR.ROOT
F8 = {"e4m3": (torch.float8_e4m3fn, 448.0), "e5m2": (torch.float8_e5m2, 57344.0)}
MODE = ["x3"]            # arithmetic of a text-side Linear
A_EXP = {"hi": 0, "lo": 12}   # activations: fixed power-of-two scales (hi8 = q8(a_hi), lo8 = q8(a_lo 2^12)); weights: per tensor
_W8 = {}
ONLY_OPS = tuple(o for o in os.environ.get("SPLIT_ONLY", "").split(",") if o)   # if set: only these text-side ops take the variant's arithmetic
EXTRA_OPS = tuple(o for o in os.environ.get("SPLIT_OPS", "").split(",") if o)   # cross-block Linears (cq, cproj) that also take the split arithmetic


def q8(x, fmt):
    dt, mx = F8[fmt]
    return x.clamp(-mx, mx).to(dt).float()


def w_terms(w, key, fmt):
    ck = (id(w), key, fmt)
    if ck not in _W8:
        W = w[key]
        hi = W.to(HF).float()
        lo = W - hi
        if fmt is None:
            _W8[ck] = (hi, lo.to(HF).float())
        else:
            mx = F8[fmt][1]
            e1 = int(np.floor(np.log2(mx / 2 / max(float(hi.abs().max()), 1e-30))))
            e2 = int(np.floor(np.log2(mx / 2 / max(float(lo.abs().max()), 1e-30))))
            _W8[ck] = (hi, lo.to(HF).float(), q8(hi * 2.0 ** e1, fmt) * 2.0 ** -e1, q8(lo * 2.0 ** e2, fmt) * 2.0 ** -e2)
    return _W8[ck]


def ln_tagged(w, key, x, eps):
    """LayerNorm whose result remembers its offset beta: a consuming Linear can then keep beta out of its rounded operand
    (variants ending in "b": (LN(x) - beta) W^T + (W beta + b) - the offset's share of the product is exact)."""
    y = AR._ln0(w, key, x, eps)
    y._beta = w[key + ".bias"]
    return y


def text_lin(w, key, x, site_op):
    eng, layer, op = site_op
    if eng == "vit" or (op in AR.CROSS and op not in EXTRA_OPS) or MODE[0] == "exact":
        return AR._lin0(w, key, x, site_op)
    m = MODE[0]
    if ONLY_OPS and op not in ONLY_OPS:
        m = "x1"                                  # every other text-side Linear: one fp16 product
    b = w[key + ".bias"]
    if m.endswith("b"):
        m = m[:-1]
        beta = getattr(x, "_beta", None)
        if beta is not None:
            x = x - beta
            b = b + (w[key + ".weight"].double() @ beta.double()).float()
    a_hi = x.to(HF).float()
    a_lo = x - a_hi
    if m == "x1":
        return F.linear(a_hi, w_terms(w, key + ".weight", None)[0], b)
    if m in ("x3", "x2a", "x2w"):
        w_hi, w_lo = w_terms(w, key + ".weight", None)
        y = F.linear(a_hi, w_hi, b)
        if m != "x2w":
            y = y + F.linear(a_lo.to(HF).float(), w_hi)
        if m != "x2a":
            y = y + F.linear(a_hi, w_lo)
        return y
    fmt = "e4m3" if m == "x1f8" else "e5m2"
    w_hi, w_lo, w_hi8, w_lo8 = w_terms(w, key + ".weight", fmt)
    a_lo8 = q8(a_lo * 2.0 ** A_EXP["lo"], fmt) * 2.0 ** -A_EXP["lo"]
    a_hi8 = q8(a_hi * 2.0 ** A_EXP["hi"], fmt) * 2.0 ** -A_EXP["hi"]
    return F.linear(a_hi, w_hi, b) + F.linear(a_lo8, w_hi8) + F.linear(a_hi8, w_lo8)


def load_wide(name):
    g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
    z = np.load(os.path.join(ROOT, f"tests/golden/{'outlier224_wide' if name == 'outlier' else 'rank224_wide'}.npz"))
    pre = "" if name == "outlier" else "c100_"
    refs, cand, groups, caps, labels = z[pre + "refs"], z[pre + "cand"], z[pre + "groups"], [str(c) for c in z[pre + "caps"]], z[pre + "labels"]
    ref, gref = z[pre + "logits"], z[pre + "group_logits"]
    seed, profile = int(z["seed"]), str(z["profile"])
    sd2 = weights.synth_state_dict(weights.nlvr_param_spec(g, v), seed, profile)
    sd1 = weights.synth_state_dict(weights.retrieval_param_spec(g, v), seed + 1, profile)
    imgs = synthetic.scene_images(range(int(z["n_index"])), 224)
    keep = np.where(labels.any(1))[0]
    return dict(sd2=sd2, sd1=sd1, imgs=imgs, tok=synthetic.HashTokenizer(), refs=refs[keep], cand=cand[keep], groups=groups[keep],
                caps=[caps[q] for q in keep], ref=ref[keep], gref=gref[keep])


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "rank"
    args = [a for a in sys.argv[2:] if not a.startswith("q=")]
    max_q = next((int(a[2:]) for a in sys.argv[2:] if a.startswith("q=")), None)
    variants = args or ["x3", "x1f8", "x2a", "x1"]
    AR._lin0, AR._ln0 = AR.lin, AR.ln
    AR.lin, AR.ln = text_lin, ln_tagged
    fx = load_wide(which)
    # text side: values, weights and streams (the cross block's sum and LayerNorm included) exact at the sites - its Linears' arithmetic
    # is `text_lin`; ViT + cross-block operands fp16, ViT stream fp16 (= set_precision("text32"))
    text = lambda e, l, o, k: e in ("med", "nlvr") and (o not in AR.CROSS or k in ("s", "t") or (o in EXTRA_OPS and k in ("a", "w")))
    pol = AR.Policy(HF, HF, [(text, None)])
    rows = []
    for m in variants:
        MODE[0] = m
        t0 = time.time()
        st = AR.stats(*AR.run(fx, pol, max_q), fx)
        st.update(variant=m, seconds=round(time.time() - t0, 1))
        rows.append(st)
        print(f"{which:8s} {m:6s} max|d| {st['max_abs']:.2e} centred rms {st['rms_centred']:.2e} exact {st['exact']:.3f} tau {st['tau']:.4f} top10 {st['top10']:.3f} "
              f"({st['seconds']} s)", flush=True)
    for r in rows:
        r["variant"] += (("+" + "+".join(EXTRA_OPS)) if EXTRA_OPS else "") + ((" only:" + ",".join(ONLY_OPS)) if ONLY_OPS else "")
    variants = [r["variant"] for r in rows]
    out = os.path.join(ROOT, "profiles", f"r6_split8_probe_{which}.json")
    prev = json.load(open(out))["rows"] if os.path.exists(out) else []
    keep = [r for r in prev if r["variant"] not in variants]
    json.dump(dict(fixture=which + "_wide", queries=len(fx["refs"]) if max_q is None else max_q, a_exp=A_EXP, rows=keep + rows), open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
