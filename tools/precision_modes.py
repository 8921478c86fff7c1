"""Rank fidelity and throughput of every operand / residual-stream precision mode on one MI355X (rounds 4-5).
Round 5 adds the EXACT mode (fp32 on the f32-input MFMA) and the fixtures regrown to >= 16 scored queries
(tests/golden/rank224_wide.npz c100 / c200 / f50, outlier224_wide.npz):  python tools/precision_modes.py --wide > profiles/r5_precision_modes.json


For each mode: (i) the two rank fixtures that hold the REFERENCE's own outputs - tests/golden/outlier224.npz (outlier-channel
weights, K = 100 + 5) and tests/golden/rank224.npz c100 (separated logits, K = 100 + 5) - scored through
generate_cirr_val_predictions: max |dlogit|, fraction of sorted positions holding the reference's candidate, Kendall tau, top-10
overlap; (ii) the benchmark step (64 queries x 105 candidates from pixels) timed over 3 steps.

    python tools/precision_modes.py [--no-timing] > profiles/r4_precision_modes.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scipy.stats import kendalltau

from candidate_reranking_cir_amd import config, synthetic, validate_stage2 as V, weights
from candidate_reranking_cir_amd.blip_stage1 import BLIP_Retrieval
from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR
from tests import helpers as H

BF, HF, F32 = torch.bfloat16, torch.float16, torch.float32
# name -> (text operands, image operands (None = same), text stream (None = auto), ViT stream ("same" / None = auto / dtype))
MODES = {
    "bf16 | streams f16": (BF, None, HF, "same"),
    "bf16 | streams f32": (BF, None, F32, "same"),
    "mixed (ViT+cross bf16, text f16) | ViT stream f16, text f32": (HF, BF, F32, HF),
    "mixed (ViT+cross bf16, text f16) | streams f16": (HF, BF, HF, HF),
    "f16 | streams f16": (HF, None, HF, "same"),
    "f16 | streams f16, fusion layers 9-11 text stream f32": (HF, None, HF, "same", 9),
    "f16 | streams f16, fusion layers 6-11 text stream f32": (HF, None, HF, "same", 6),
    "f16 | streams f16, fusion layers 3-11 text stream f32": (HF, None, HF, "same", 3),
    "f16 | ViT stream f16, text f32": (HF, None, F32, HF),
    "f16 | streams f32": (HF, None, F32, "same"),
    "text32 (text side: fp32 rows as split8 - fp16 + 2 scaled-fp8 products, fp32 stream; ViT + cross block f16)": (F32, HF, F32, HF, None, 8),
    "text32x3 (round 5: text side as three fp16 products on [hi|lo|hi] rows)": (F32, HF, F32, HF, None, 3),
    "exact (fp32 everywhere, f32-input MFMA)": (F32, None, F32, "same"),
}


def order_stats(ours, ref):
    o, r = np.argsort(-ours, kind="stable"), np.argsort(-ref, kind="stable")
    return float((o == r).mean()), float(kendalltau(ours, ref).statistic), len(set(o[:10]) & set(r[:10])) / 10.0


def apply(m, mode):
    dt, idt, sdt, vsdt = MODES[mode][:4]
    if hasattr(m, "set_text_stream32_from"):                       # (the stage-I model has no two-branch encoder)
        m.set_text_stream32_from(MODES[mode][4] if len(MODES[mode]) > 4 else None)
    if dt == F32:
        arith = MODES[mode][5] if len(MODES[mode]) > 5 else 8
        return m.set_precision("exact" if idt is None else ("text32" if arith == 8 else "text32x3"))
    if m.compute_dtype == F32:
        m.set_precision("f16")
    m.set_compute_dtype(dt, idt)
    m.set_stream_dtype(sdt, vit=vsdt)
    return m


_MODELS, _BANKS = {}, {}


def build(g, v, seed, profile, mode, dev):
    """The model pair of a (seed, profile) is built ONCE (0.5 G random parameters on the host: ~40 s) and switched between modes."""
    key = (seed, profile)
    if key not in _MODELS:
        sd2, sd1 = H.state_dicts(g, v, seed, profile)
        m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
        m1 = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
        m2.load_state_dict(sd2); m1.load_state_dict(sd1)
        _MODELS[key] = (m2.to(dev).float().eval(), m1.to(dev).float().eval())
    m2, m1 = _MODELS[key]
    return apply(m2, mode), apply(m1, mode)


def fixture_stats(name, mode, dev):
    """`name`: outlier224 | rank224_c100 (the round-4 fixtures: 2 / 4 scored queries) | outlier224_wide | rank224_wide_c100 | _c200 | _f50
    (round 5: 16 scored queries each)."""
    g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
    fname, _, tag = name.partition("_c") if "_c" in name else name.partition("_f")
    tag = ("c" if "_c" in name else "f") + tag if tag else ""
    z = H.load(fname + ".npz")
    pre = tag + "_" if tag else ""
    refs, cand, labels, caps, ref = z[pre + "refs"], z[pre + "cand"], z[pre + "labels"], z[pre + "caps"], z[pre + "logits"]
    fiq = tag.startswith("f")
    groups, targets, gref = (None, None, None) if fiq else (z[pre + "groups"], z[pre + "targets"], z[pre + "group_logits"])
    m2, m1 = build(g, v, int(z["seed"]), str(z["profile"]), mode, dev)
    bkey = (int(z["seed"]), str(z["profile"]), int(z["n_index"]), mode)
    if bkey not in _BANKS:
        _BANKS.clear()
        _BANKS[bkey] = V.extract_index_features(synthetic.scene_images(range(int(z["n_index"])), 224), m2, batch_size=64)
    bank = _BANKS[bkey]
    if fiq:
        ds = V.RelativeValSet(ref_index=refs, cand_index=cand, labels=labels, captions=[V.fiq_caption(str(p[0]), str(p[1])) for p in caps])
        logits, gl = V.generate_fiq_val_predictions(m2, m1, ds, bank, query_batch=4).cpu().numpy(), None
    else:
        ds = V.RelativeValSet(ref_index=refs, cand_index=cand, labels=labels, captions=[str(c) for c in caps], group_index=groups, target_index=targets)
        lt, gt = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=4)
        logits, gl = lt.cpu().numpy(), gt.cpu().numpy()
    scored = labels.any(1)
    per_q = np.array([order_stats(logits[q], ref[q]) for q in np.where(scored)[0]])
    st = per_q.mean(0)
    e = logits[scored] - ref[scored]
    # pairs the reference separates by more than 4 x this run's own max error: how many there are, and whether all keep their order
    tol = float(np.abs(e).max())
    dec = flips = tot = 0
    for q in np.where(scored)[0]:
        iu = np.triu_indices(logits.shape[1], 1)
        dr, do = (ref[q][:, None] - ref[q][None, :])[iu], (logits[q][:, None] - logits[q][None, :])[iu]
        d = np.abs(dr) > 4 * tol
        dec += int(d.sum()); tot += len(dr); flips += int((np.sign(dr[d]) != np.sign(do[d])).sum())
    extra = {}
    if "logits_f64" in z.files:      # conditioning: the reference's own fp32-vs-fp64 deviation per query (oracle/make_golden.py wide outlier64)
        qs = np.where(scored)[0]
        noise = np.abs(z["logits_f64"][qs] - ref[qs].astype(np.float64)).max(1)
        well = noise < 2e-5
        extra = dict(well_conditioned_queries=int(well.sum()), reference_fp32_vs_fp64_noise_max=float(noise.max()),
                     well_max_abs=float(np.abs(e[well]).max()), well_exact=float(per_q[well, 0].mean()), well_tau=float(per_q[well, 1].mean()),
                     well_tau_worst=float(per_q[well, 1].min()), well_top10=float(per_q[well, 2].mean()))
    return dict(**extra, scored_queries=int(scored.sum()), max_abs=float(max(np.abs(e).max(), 0.0 if gl is None else np.abs(gl - gref).max())),
                rms_centred=float(np.sqrt(((e - e.mean(1, keepdims=True)) ** 2).mean())),
                logit_sigma=float(ref[scored].std(axis=1).mean()), exact=float(st[0]), tau=float(st[1]), top10=float(st[2]),
                tau_worst_query=float(per_q[:, 1].min()), top1_agree=float(np.mean([np.argmax(logits[q]) == np.argmax(ref[q]) for q in np.where(scored)[0]])),
                pairs_decided_at_4x_own_error=dec / tot, decided_pairs_flipped=flips)


def timing(mode, dev, q_n=64, k=105, steps=3):
    g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
    m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    m2.load_state_dict(weights.synth_state_dict(weights.nlvr_param_spec(g, v), 0, "test"))
    m1 = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    m1.load_state_dict(weights.synth_state_dict(weights.retrieval_param_spec(g, v), 1, "test"))
    m2, m1 = apply(m2.to(dev).eval(), mode), apply(m1.to(dev).eval(), mode)
    gen = torch.Generator(device=dev).manual_seed(1234)
    images = torch.randn((q_n + q_n * k, 3, 224, 224), generator=gen, device=dev, dtype=torch.float32).to(m2.token_dtype)
    ids = torch.stack([synthetic.caption_ids(q, 32) for q in range(q_n)]).to(dev)
    mask = torch.ones_like(ids)
    qidx = torch.arange(q_n, device=dev).repeat_interleave(k)

    def step():
        toks = m2.img_embed16(images)
        z = m1.z_t(toks[:q_n], ids, mask)
        return m2.score(z.last_hidden_state, ids, mask, toks[q_n:], qidx)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    return q_n * k * steps / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--modes", default="")
    ap.add_argument("--wide", action="store_true", help="round 5: the >= 16-query fixtures besides the two round-4 ones")
    args = ap.parse_args()
    dev = torch.device("cuda")
    rows = []
    for mode in MODES:
        if args.modes and not any(s in mode for s in args.modes.split(",")):
            continue
        row = dict(mode=mode)
        for fx in ("outlier224",) + (("outlier224_wide",) if args.wide else ()) + ("rank224_c100",) + (("rank224_wide_c100", "rank224_wide_c200", "rank224_wide_f50") if args.wide else ()):
            row[fx] = fixture_stats(fx, mode, dev)
        torch.cuda.empty_cache()
        if not args.no_timing:
            row["triplets_per_s"] = round(timing(mode, dev), 1)
        torch.cuda.empty_cache()
        rows.append(row)
        for fx in row:
            if fx.endswith(("_wide", "_c200", "_f50")) or fx == "rank224_wide_c100":
                w = row[fx]
                if "well_tau" in w:
                    print(f"   {fx:20s} well-conditioned {w['well_conditioned_queries']} q: max|d| {w['well_max_abs']:.2e}  exact {w['well_exact']:.3f}  tau {w['well_tau']:.4f} "
                          f"(worst {w['well_tau_worst']:.4f})  top10 {w['well_top10']:.3f}", file=sys.stderr, flush=True)
                print(f"   {fx:20s} {w['scored_queries']:2d} q  max|d| {w['max_abs']:.2e}  exact {w['exact']:.3f}  tau {w['tau']:.4f} (worst {w['tau_worst_query']:.4f})  top10 {w['top10']:.3f}  "
                      f"top1 {w['top1_agree']:.2f}  decided pairs {w['pairs_decided_at_4x_own_error']:.3f} flipped {w['decided_pairs_flipped']}", file=sys.stderr, flush=True)
        o, r = row["outlier224"], row["rank224_c100"]
        print(f"{mode:62s} outlier: max|d| {o['max_abs']:.2e} exact {o['exact']:.2f} tau {o['tau']:.3f} top10 {o['top10']:.2f} | rank224 c100: "
              f"max|d| {r['max_abs']:.2e} exact {r['exact']:.3f} tau {r['tau']:.4f} top10 {r['top10']:.2f} | {row.get('triplets_per_s', 0):.0f} triplets/s",
              file=sys.stderr, flush=True)
    print(json.dumps(dict(rows=rows, note="fixtures: the reference's own fp32 outputs; timing: 64 queries x 105 candidates from pixels, 3 steps"), indent=1))


if __name__ == "__main__":
    main()
