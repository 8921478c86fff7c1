"""Rank fidelity and throughput of every operand / residual-stream precision mode on one MI355X (round 4).

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
    "f16 | ViT stream f16, text f32": (HF, None, F32, HF),
    "f16 | streams f32": (HF, None, F32, "same"),
}


def order_stats(ours, ref):
    o, r = np.argsort(-ours, kind="stable"), np.argsort(-ref, kind="stable")
    return float((o == r).mean()), float(kendalltau(ours, ref).statistic), len(set(o[:10]) & set(r[:10])) / 10.0


def apply(m, mode):
    dt, idt, sdt, vsdt = MODES[mode]
    m.set_compute_dtype(dt, idt)
    m.set_stream_dtype(sdt, vit=vsdt)
    return m


def build(g, v, seed, profile, mode, dev):
    sd2, sd1 = H.state_dicts(g, v, seed, profile)
    m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    m1 = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
    m2.load_state_dict(sd2); m1.load_state_dict(sd1)
    return apply(m2.to(dev).float().eval(), mode), apply(m1.to(dev).float().eval(), mode)


def fixture_stats(name, mode, dev):
    g, v = H.geometry(H.FULL_BERT, dict(image_size=224))
    if name == "outlier224":
        z = H.load("outlier224.npz")
        refs, cand, labels, caps, groups, targets, ref, gref = (z["refs"], z["cand"], z["labels"], z["caps"], z["groups"], z["targets"],
                                                                z["logits"], z["group_logits"])
    else:
        z = H.load("rank224.npz")
        refs, cand, labels, caps, groups, targets, ref, gref = (z["c100_refs"], z["c100_cand"], z["c100_labels"], z["c100_caps"], z["c100_groups"],
                                                                z["c100_targets"], z["c100_logits"], z["c100_group_logits"])
    m2, m1 = build(g, v, int(z["seed"]), str(z["profile"]), mode, dev)
    imgs = synthetic.scene_images(range(int(z["n_index"])), 224)
    bank = V.extract_index_features(imgs, m2, batch_size=64)
    ds = V.RelativeValSet(ref_index=refs, cand_index=cand, labels=labels, captions=[str(c) for c in caps], group_index=groups, target_index=targets)
    lt, gt = V.generate_cirr_val_predictions(m2, m1, ds, bank, query_batch=4)
    logits, gl = lt.cpu().numpy(), gt.cpu().numpy()
    scored = labels.any(1)
    st = np.array([order_stats(logits[q], ref[q]) for q in np.where(scored)[0]]).mean(0)
    e = logits[scored] - ref[scored]
    return dict(max_abs=float(max(np.abs(e).max(), np.abs(gl - gref).max())), rms_centred=float(np.sqrt(((e - e.mean(1, keepdims=True)) ** 2).mean())),
                exact=float(st[0]), tau=float(st[1]), top10=float(st[2]))


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
    args = ap.parse_args()
    dev = torch.device("cuda")
    rows = []
    for mode in MODES:
        if args.modes and not any(s in mode for s in args.modes.split(",")):
            continue
        row = dict(mode=mode)
        for fx in ("outlier224", "rank224_c100"):
            row[fx] = fixture_stats(fx, mode, dev)
        torch.cuda.empty_cache()
        if not args.no_timing:
            row["triplets_per_s"] = round(timing(mode, dev), 1)
        torch.cuda.empty_cache()
        rows.append(row)
        o, r = row["outlier224"], row["rank224_c100"]
        print(f"{mode:62s} outlier: max|d| {o['max_abs']:.2e} exact {o['exact']:.2f} tau {o['tau']:.3f} top10 {o['top10']:.2f} | rank224 c100: "
              f"max|d| {r['max_abs']:.2e} exact {r['exact']:.3f} tau {r['tau']:.4f} top10 {r['top10']:.2f} | {row.get('triplets_per_s', 0):.0f} triplets/s",
              file=sys.stderr, flush=True)
    print(json.dumps(dict(rows=rows, note="fixtures: the reference's own fp32 outputs; timing: 64 queries x 105 candidates from pixels, 3 steps"), indent=1))


if __name__ == "__main__":
    main()
