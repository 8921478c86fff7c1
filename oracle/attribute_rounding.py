"""Where does the 16-bit rank error enter?  Per-site rounding emulation on the CPU (TEST INFRASTRUCTURE, like the rest of oracle/).

The reference's arithmetic (restated as in cir_oracle.py) with a rounding inserted at every place the HIP path stores or
consumes a 16-bit value; a POLICY decides the format of every such site, so that one piece (an encoder, a layer, an operator
class, the weights or the activations of it) can run with bf16 roundings while everything else runs with fp16 roundings.
Inputs are the two rank fixtures whose reference outputs are committed (tests/golden/outlier224.npz: outlier-channel
weights; tests/golden/rank224.npz c100: separated logits) - the errors are measured against the reference's own fp32 logits.

Sites: (engine, layer, op, kind) with engine in {vit, med, nlvr}; op in {patch, qkv, attn, proj, cq, ckv, cattn, cproj, fc1,
fc2, cls, tokens, zt}; kind in {"a": the 16-bit operand copy a GEMM reads, "w": its weight matrix, "o": a GEMM / attention
output stored in 16 bits (it IS the next operand), "p": softmax probabilities fed to P.V, "s": residual-stream storage}.

    python oracle/attribute_rounding.py [outlier|rank] [quick]      -> profiles/r4_precision_attribution_<fixture>.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from scipy.stats import kendalltau

from candidate_reranking_cir_amd import config, synthetic, weights
from candidate_reranking_cir_amd.blip_stage2 import encode_text
from oracle import cir_oracle as O

torch.set_num_threads(int(os.environ.get("CIR_THREADS", "8")))
BF, HF = torch.bfloat16, torch.float16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Policy:
    """format of a site = first matching rule, else `base`; a rule is (predicate(engine, layer, op, kind), dtype or None = exact)."""

    def __init__(self, base, stream, rules=()):
        self.base, self.stream, self.rules = base, stream, list(rules)

    def __call__(self, eng, layer, op, kind):
        for pred, dt in self.rules:
            if pred(eng, layer, op, kind):
                return dt
        return self.stream if kind in ("s", "t") else self.base


POL = [None]
_WCACHE = {}


def rq(x, site):
    dt = POL[0](*site)
    return x if dt is None else x.to(dt).float()


def wq(w, key, site):
    """rounded weight matrix, cached per (key, format): the same 600 tensors are re-used by every candidate batch"""
    dt = POL[0](*site)
    if dt is None:
        return w[key]
    ck = (id(w), key, dt)
    if ck not in _WCACHE:
        _WCACHE[ck] = w[key].to(dt).float()
    return _WCACHE[ck]


def lin(w, key, x, site_op):
    """x is already rounded where it was stored; the weight takes the format of its own site"""
    eng, layer, op = site_op
    return F.linear(x, wq(w, key + ".weight", (eng, layer, op, "w")), w[key + ".bias"])


def ln(w, key, x, eps):
    return F.layer_norm(x, (x.shape[-1],), w[key + ".weight"], w[key + ".bias"], eps)


FOLD_LN = [False]     # emulate the LayerNorm fold of the ViT: the consuming GEMM runs on the RAW fp16 stream with weights W o gamma


def ln_lin_fold(w, ln_key, lin_key, x, eps, site_op):
    """LN(x) W^T + b as rstd * (x W'^T - mu s) + b', W' = round16(W o gamma), s = row sums of W', b' = b + W beta (fp32):
    what a GEMM on the raw residual stream computes when the statistics arrive separately (LayerNorm fold)."""
    eng, layer, op = site_op
    dt = POL[0](eng, layer, op, "w")
    g, be = w[ln_key + ".weight"], w[ln_key + ".bias"]
    W = w[lin_key + ".weight"]
    Wp = W * g[None, :]
    Wp = Wp if dt is None else Wp.to(dt).float()
    s = Wp.sum(1)
    bp = w[lin_key + ".bias"] + W @ be
    mu = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + eps)
    return rstd * (x @ Wp.t() - mu * s) + bp


def vit(w, image, prefix="visual_encoder.", eps=1e-6):
    E = "vit"
    pw = w[prefix + "patch_embed.proj.weight"]
    d = pw.shape[0]
    nh = d // 64
    x = F.conv2d(rq(image, (E, -1, "patch", "a")), wq(w, prefix + "patch_embed.proj.weight", (E, -1, "patch", "w")),
                 w[prefix + "patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
    b = x.shape[0]
    x = rq(x, (E, -1, "patch", "s"))
    x = torch.cat([w[prefix + "cls_token"].expand(b, -1, -1), x], dim=1) + w[prefix + "pos_embed"][:, : x.shape[1] + 1, :]
    x = rq(x, (E, -1, "patch", "s"))
    for i in range(12):
        p = f"{prefix}blocks.{i}."
        if FOLD_LN[0]:
            qkv = rq(ln_lin_fold(w, p + "norm1", p + "attn.qkv", x, eps, (E, i, "qkv")), (E, i, "qkv", "o"))
        else:
            y = rq(ln(w, p + "norm1", x, eps), (E, i, "qkv", "a"))
            qkv = rq(lin(w, p + "attn.qkv", y, (E, i, "qkv")), (E, i, "qkv", "o"))
        n = qkv.shape[1]
        qkv = qkv.reshape(b, n, 3, nh, 64).permute(2, 0, 3, 1, 4)
        a = (qkv[0] @ qkv[1].transpose(-2, -1)) * 0.125
        a = rq(a.softmax(dim=-1), (E, i, "attn", "p"))
        y = rq((a @ qkv[2]).transpose(1, 2).reshape(b, n, d), (E, i, "attn", "o"))
        x = rq(x + lin(w, p + "attn.proj", y, (E, i, "proj")), (E, i, "proj", "s"))
        if FOLD_LN[0]:
            f = rq(F.gelu(ln_lin_fold(w, p + "norm2", p + "mlp.fc1", x, eps, (E, i, "fc1"))), (E, i, "fc1", "o"))
        else:
            y = rq(ln(w, p + "norm2", x, eps), (E, i, "fc1", "a"))
            f = rq(F.gelu(lin(w, p + "mlp.fc1", y, (E, i, "fc1"))), (E, i, "fc1", "o"))
        x = rq(x + lin(w, p + "mlp.fc2", f, (E, i, "fc2")), (E, i, "fc2", "s"))
    return ln(w, prefix + "norm", x, eps)            # fp32 tokens; their 16-bit copy is rounded where it is consumed ("tokens")


def sdpa(q, k, v, mask, nh, site):
    eng, layer, op = site
    qh, kh, vh = O._heads(q, nh), O._heads(k, nh), O._heads(v, nh)
    s = qh @ kh.transpose(-1, -2) / 8.0
    if mask is not None:
        s = s + mask
    p = rq(torch.softmax(s, -1), (eng, layer, op, "p"))
    ctx = p @ vh
    b, h, t, dh = ctx.shape
    return rq(ctx.transpose(1, 2).reshape(b, t, h * dh), (eng, layer, op, "o"))


def med(w, ids, mask, enc32, prefix="text_encoder.", eps=1e-12):
    E = "med"
    enc = rq(enc32, (E, -1, "tokens", "a"))
    hs = rq(O.bert_embeddings(w, ids, prefix, eps), (E, -1, "emb", "s"))
    sm = O.self_mask_additive(mask)
    for i in range(12):
        p = f"{prefix}encoder.layer.{i}."
        a = p + "attention.self."
        h = rq(hs, (E, i, "qkv", "a"))
        q, k, v = (rq(lin(w, a + n, h, (E, i, "qkv")), (E, i, "qkv", "o")) for n in ("query", "key", "value"))
        ctx = sdpa(q, k, v, sm, 12, (E, i, "attn"))
        t = rq(lin(w, p + "attention.output.dense", ctx, (E, i, "proj")) + hs, (E, i, "proj", "t"))
        hs = rq(ln(w, p + "attention.output.LayerNorm", t, eps), (E, i, "proj", "s"))
        c = p + "crossattention.self."
        h = rq(hs, (E, i, "cq", "a"))
        q = rq(lin(w, c + "query", h, (E, i, "cq")), (E, i, "cq", "o"))
        k, v = (rq(lin(w, c + n, enc, (E, i, "ckv")), (E, i, "ckv", "o")) for n in ("key", "value"))
        ctx = sdpa(q, k, v, None, 12, (E, i, "cattn"))
        t = rq(lin(w, p + "crossattention.output.dense", ctx, (E, i, "cproj")) + hs, (E, i, "cproj", "t"))
        hs = rq(ln(w, p + "crossattention.output.LayerNorm", t, eps), (E, i, "cproj", "s"))
        h = rq(hs, (E, i, "fc1", "a"))
        f = rq(F.gelu(lin(w, p + "intermediate.dense", h, (E, i, "fc1"))), (E, i, "fc1", "o"))
        t = rq(lin(w, p + "output.dense", f, (E, i, "fc2")) + hs, (E, i, "fc2", "t"))
        hs = rq(ln(w, p + "output.LayerNorm", t, eps), (E, i, "fc2", "s"))
    return hs                                          # z_t: the API hands fp32 of the stream copy on


def nlvr(w, ids, mask, z_t, cand32, prefix="text_encoder.", eps=1e-12):
    E = "nlvr"
    cand = rq(cand32, (E, -1, "tokens", "a"))
    emb = O.bert_embeddings(w, ids, prefix, eps)
    sm = O.self_mask_additive(mask)
    hs = [rq(z_t, (E, -1, "zt", "s")), rq(emb, (E, -1, "zt", "s"))]
    for i in range(12):
        p = f"{prefix}encoder.layer.{i}."
        att = []
        for b in (0, 1):
            s = f"{p}attention.self{b}."
            h = rq(hs[b], (E, i, "qkv", "a"))
            q, k, v = (rq(lin(w, s + n, h, (E, i, "qkv")), (E, i, "qkv", "o")) for n in ("query", "key", "value"))
            ctx = sdpa(q, k, v, sm, 12, (E, i, "attn"))
            t = rq(lin(w, f"{p}attention.output.dense{b}", ctx, (E, i, "proj")) + hs[b], (E, i, "proj", "t"))
            att.append(rq(ln(w, p + "attention.output.LayerNorm" + "AB"[b], t, eps), (E, i, "proj", "s")))
        dd = []
        for b in (0, 1):
            s = f"{p}crossattention.self{b}."
            h = rq(att[b], (E, i, "cq", "a"))
            q = rq(lin(w, s + "query", h, (E, i, "cq")), (E, i, "cq", "o"))
            k, v = (rq(lin(w, s + n, cand, (E, i, "ckv")), (E, i, "ckv", "o")) for n in ("key", "value"))
            ctx = sdpa(q, k, v, None, 12, (E, i, "cattn"))
            dd.append(lin(w, f"{p}crossattention.output.dense{b}", ctx, (E, i, "cproj")))     # folded with the merge: no store between
        mk = p + "crossattention.output.merge_layer"
        m = lin(w, mk, torch.cat(dd, -1), (E, i, "cproj")) if mk + ".weight" in w else (dd[0] + dd[1]) / 2
        m = rq(m, (E, i, "cproj", "t"))
        x = [rq(ln(w, p + "crossattention.output.LayerNorm" + "AB"[b], m + att[b], eps), (E, i, "cproj", "s")) for b in (0, 1)]
        for b in (0, 1):
            h = rq(x[b], (E, i, "fc1", "a"))
            f = rq(F.gelu(lin(w, p + "intermediate.dense", h, (E, i, "fc1"))), (E, i, "fc1", "o"))
            t = rq(lin(w, p + "output.dense", f, (E, i, "fc2")) + x[b], (E, i, "fc2", "t"))
            hs[b] = rq(ln(w, p + "output.LayerNorm", t, eps), (E, i, "fc2", "s"))
    hid = rq(torch.cat([hs[0][:, 0], hs[1][:, 0]], -1), (E, 12, "cls", "a"))
    y = rq(F.relu(lin(w, "cls_head.0", hid, (E, 12, "cls"))), (E, 12, "cls", "o"))
    return lin(w, "cls_head.2", y, (E, 12, "cls"))[:, 0]


# ------------------------------------------------------------------------------------------------ fixtures
def load_fixture(name):
    g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
    if name == "outlier":
        z = np.load(os.path.join(ROOT, "tests/golden/outlier224.npz"))
        refs, cand, groups, caps, labels = z["refs"], z["cand"], z["groups"], [str(c) for c in z["caps"]], z["labels"]
        ref, gref = z["logits"], z["group_logits"]
    else:
        z = np.load(os.path.join(ROOT, "tests/golden/rank224.npz"))
        refs, cand, groups, caps, labels = (z["c100_refs"], z["c100_cand"], z["c100_groups"], [str(c) for c in z["c100_caps"]], z["c100_labels"])
        ref, gref = z["c100_logits"], z["c100_group_logits"]
    seed, profile = int(z["seed"]), str(z["profile"])
    sd2 = weights.synth_state_dict(weights.nlvr_param_spec(g, v), seed, profile)
    sd1 = weights.synth_state_dict(weights.retrieval_param_spec(g, v), seed + 1, profile)
    imgs = synthetic.scene_images(range(int(z["n_index"])), 224)
    tok = synthetic.HashTokenizer()
    keep = np.where(labels.any(1))[0]                 # scored queries only
    return dict(sd2=sd2, sd1=sd1, imgs=imgs, tok=tok, refs=refs[keep], cand=cand[keep], groups=groups[keep], caps=[caps[q] for q in keep],
                ref=ref[keep], gref=gref[keep], bank_slice=z["bank_slice"])


def run(fx, pol, max_q=None):
    POL[0] = pol
    with torch.no_grad():
        used = sorted(set(fx["refs"].tolist()) | set(fx["cand"].ravel().tolist()) | set(fx["groups"].ravel().tolist()))
        row = {j: i for i, j in enumerate(used)}
        sig = (FOLD_LN[0],) + tuple(str(pol("vit", l, o, k)) for l in range(-1, 12) for o in ("patch", "qkv", "attn", "proj", "fc1", "fc2") for k in "awops")
        if sig not in fx.setdefault("_banks", {}):       # most runs differ only in the fusion part: one ViT pass per distinct ViT policy
            fx["_banks"][sig] = torch.cat([vit(fx["sd2"], fx["imgs"][used[i:i + 32]]) for i in range(0, len(used), 32)])
        bank = fx["_banks"][sig]
        out, gout = [], []
        for q in range(len(fx["refs"]) if max_q is None else max_q):
            ids, mask = encode_text(fx["tok"], [fx["caps"][q]], "cpu")          # [ENC] in slot 0, blip_stage2.py:114
            z = med(fx["sd1"], ids, mask, bank[row[int(fx["refs"][q])]][None])
            allc = np.concatenate([fx["cand"][q], fx["groups"][q]])
            c = bank[[row[int(j)] for j in allc]]
            k = len(allc)
            lg = nlvr(fx["sd2"], ids.expand(k, -1), mask.expand(k, -1), z.expand(k, -1, -1), c).numpy()
            out.append(lg[:len(fx["cand"][q])])
            gout.append(lg[len(fx["cand"][q]):])
    return np.stack(out), np.stack(gout)


def stats(out, gout, fx):
    n = len(out)
    ref, gref = fx["ref"][:n], fx["gref"][:n]
    ex, tau, top = [], [], []
    for q in range(n):
        o, r = np.argsort(-out[q], kind="stable"), np.argsort(-ref[q], kind="stable")
        ex.append(float((o == r).mean()))
        tau.append(float(kendalltau(out[q], ref[q]).statistic))
        top.append(len(set(o[:10]) & set(r[:10])) / 10.0)
    e = out - ref
    return dict(max_abs=float(max(np.abs(e).max(), np.abs(gout - gref).max())), rms=float(np.sqrt((e ** 2).mean())),
                rms_centred=float(np.sqrt(((e - e.mean(1, keepdims=True)) ** 2).mean())), exact=float(np.mean(ex)), tau=float(np.mean(tau)),
                top10=float(np.mean(top)), sigma=float(ref.std(1).mean()))


def piece(**kw):
    """predicate over sites: every given field must match (a value or a set of values)"""
    def pred(eng, layer, op, kind):
        for name, val in kw.items():
            x = dict(eng=eng, layer=layer, op=op, kind=kind)[name]
            if isinstance(val, (set, tuple, list)):
                if x not in val:
                    return False
            elif x != val:
                return False
        return True
    return pred


CROSS = ("cq", "ckv", "cattn", "cproj", "tokens")       # the image-facing block of the text encoders (engine.py: cross_dtype)


def candidates(go):
    """Mixed modes a build could run (every rule set is realisable with per-launch operand types: one type per GEMM / attention)."""
    text = lambda e, l, o, k: k != "s" and e in ("med", "nlvr") and o not in CROSS
    image = lambda e, l, o, k: k != "s" and (e == "vit" or o in CROSS)
    patch = lambda e, l, o, k: k != "s" and e == "vit" and o == "patch"
    cls = lambda e, l, o, k: k != "s" and o == "cls"
    tstream32 = lambda e, l, o, k: k == "s" and e in ("med", "nlvr")
    go("C0  bf16 everywhere, fp16 stream", Policy(BF, HF))
    go("C1  bf16 + patch embedding fp16", Policy(BF, HF, [(patch, HF)]))
    go("C2  bf16 + patch embedding fp16 + cls_head fp16", Policy(BF, HF, [(patch, HF), (cls, HF)]))
    go("C3  mixed: ViT + cross block bf16, text side fp16; streams fp16", Policy(BF, HF, [(text, HF)]))
    go("C4  C3 + patch embedding fp16", Policy(BF, HF, [(text, HF), (patch, HF)]))
    go("C5  C4 with the text-side stream fp32", Policy(BF, HF, [(text, HF), (patch, HF), (tstream32, None)]))
    go("C6  ViT fp16, cross block bf16, text side fp16; streams fp16", Policy(HF, HF, [(lambda e, l, o, k: k != "s" and e != "vit" and o in CROSS, BF)]))
    go("C7  fp16 everywhere, fp16 stream", Policy(HF, HF))
    go("C8  fp16 everywhere, ViT stream fp16, text-side stream fp32", Policy(HF, HF, [(tstream32, None)]))
    go("C9  fp16 everywhere, fp32 stream", Policy(HF, None))


def stream_sites(go):
    """fp16 operands everywhere; which text-side residual-stream sites need fp32 storage?"""
    go("S0  fp16 stream everywhere", Policy(HF, HF))
    txt = lambda ops: (lambda e, l, o, k: k == "s" and e in ("med", "nlvr") and o in ops)
    for name, ops in (("attention-out sum + LayerNorm (proj)", ("proj",)), ("cross/merge sum + LayerNorm (cproj)", ("cproj",)),
                      ("FFN-out sum + LayerNorm (fc2)", ("fc2",)), ("z_t / embeddings", ("zt", "emb")),
                      ("proj + cproj", ("proj", "cproj")), ("cproj + fc2", ("cproj", "fc2")), ("all text-side", ("proj", "cproj", "fc2", "zt", "emb"))):
        go(f"S   fp32 at: {name}", Policy(HF, HF, [(txt(ops), None)]))
    nl = lambda e, l, o, k: k == "s" and e == "nlvr"
    go("S   fp32 at: nlvr only (stage-I stream fp16)", Policy(HF, HF, [(nl, None)]))
    go("S   fp32 at: nlvr layers 6-11 only", Policy(HF, HF, [(lambda e, l, o, k: k == "s" and e == "nlvr" and l >= 6, None)]))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "outlier"
    quick = "quick" in sys.argv[2:]
    fx = load_fixture(which)
    max_q = 1 if quick else None
    rows = []
    if "fix68" in sys.argv[2:]:      # re-measure one row of the committed table (its ViT pass had been served from a stale cache entry)
        path = os.path.join(ROOT, "profiles", f"r4_precision_attribution_{which}.json")
        doc = json.load(open(path))
        base = next(r for r in doc["rows"] if r["name"].startswith("all fp16, fp16 stream"))
        st = stats(*run(fx, Policy(HF, HF, [(lambda e, l, o, k, p=piece(eng="vit", layer=(6, 7, 8)): k != "s" and p(e, l, o, k), BF)]), max_q), fx)
        row = next(r for r in doc["rows"] if r["name"] == "bf16 only: vit: layers 6-8")
        row.update(st)
        row["added_rms_centred"] = float(np.sqrt(max(st["rms_centred"] ** 2 - base["rms_centred"] ** 2, 0.0)))
        json.dump(doc, open(path, "w"), indent=1)
        return print(which, "vit layers 6-8:", st)
    if "sums" in sys.argv[2:]:       # text-side stream: is it the pre-LayerNorm SUMS (GEMM outputs) or the LayerNorm OUTPUTS that need fp32?
        def go(name, pol):
            st = stats(*run(fx, pol, max_q), fx)
            print(f"{name:64s} max|d| {st['max_abs']:.2e} centred rms {st['rms_centred']:.2e} exact {st['exact']:.3f} tau {st['tau']:.4f} top10 {st['top10']:.2f}", flush=True)
        txt = lambda e: e in ("med", "nlvr")
        go("fp16 operands, fp16 streams (default)", Policy(HF, HF))
        go("  text-side pre-LN sums fp32, LN outputs fp16", Policy(HF, HF, [(lambda e, l, o, k: txt(e) and k == "t", None)]))
        go("  text-side LN outputs fp32, pre-LN sums fp16", Policy(HF, HF, [(lambda e, l, o, k: txt(e) and k == "s", None)]))
        go("  both fp32 (split)", Policy(HF, HF, [(lambda e, l, o, k: txt(e) and k in ("s", "t"), None)]))
        return
    if "tail" in sys.argv[2:]:       # what would a more precise TAIL (cls_head, last layer) buy in the default mode?
        def go(name, pol):
            st = stats(*run(fx, pol, max_q), fx)
            print(f"{name:64s} max|d| {st['max_abs']:.2e} centred rms {st['rms_centred']:.2e} exact {st['exact']:.3f} tau {st['tau']:.4f} top10 {st['top10']:.2f}", flush=True)
        go("default: fp16 operands + fp16 streams", Policy(HF, HF))
        go("  + cls_head exact (fp32 operands)", Policy(HF, HF, [(lambda e, l, o, k: o == "cls", None)]))
        go("  + cls_head exact + nlvr layer 11 exact (operands and stream)", Policy(HF, HF, [(lambda e, l, o, k: e == "nlvr" and (o == "cls" or l == 11), None)]))
        go("  + cls_head exact + nlvr layers 10-11 stream fp32 only", Policy(HF, HF, [(lambda e, l, o, k: o == "cls" or (e == "nlvr" and l >= 10 and k == "s"), None)]))
        for l0 in (11, 10, 9, 8, 6):
            go(f"  + nlvr stream fp32 from layer {l0} on (operands fp16 everywhere)", Policy(HF, HF, [(lambda e, l, o, k, l0=l0: e == "nlvr" and l >= l0 and k == "s", None)]))
        return
    if "fold" in sys.argv[2:]:
        for fold in (False, True):
            FOLD_LN[0] = fold
            st = stats(*run(fx, Policy(HF, HF), max_q), fx)
            print(f"fp16 operands + fp16 streams, ViT LayerNorm fold {fold}: max|d| {st['max_abs']:.2e} centred rms {st['rms_centred']:.2e} exact {st['exact']:.3f} "
                  f"tau {st['tau']:.4f} top10 {st['top10']:.2f}", flush=True)
        return
    if "candidates" in sys.argv[2:] or "streams" in sys.argv[2:]:
        def go(name, pol):
            t0 = time.time()
            st = stats(*run(fx, pol, max_q), fx)
            st["name"] = name
            rows.append(st)
            print(f"{name:70s} max|d| {st['max_abs']:.2e} centred rms {st['rms_centred']:.2e}  exact {st['exact']:.2f} tau {st['tau']:.3f} "
                  f"top10 {st['top10']:.2f}  ({time.time() - t0:.0f} s)", flush=True)
        tag = "candidates" if "candidates" in sys.argv[2:] else "streams"
        (candidates if tag == "candidates" else stream_sites)(go)
        out = os.path.join(ROOT, "profiles", f"r4_precision_{tag}_{which}.json")
        json.dump(dict(fixture=which, rows=rows), open(out, "w"), indent=1)
        return print("wrote", out)

    def go(name, pol):
        t0 = time.time()
        s = stats(*run(fx, pol, max_q), fx)
        s["name"] = name
        s["seconds"] = round(time.time() - t0, 1)
        rows.append(s)
        print(f"{name:58s} max|d| {s['max_abs']:.2e} rms {s['rms']:.2e} centred {s['rms_centred']:.2e}  exact {s['exact']:.2f} tau {s['tau']:.3f} "
              f"top10 {s['top10']:.2f}  ({s['seconds']} s)", flush=True)
        return s

    go("exact fp32 (restatement vs the reference's own logits)", Policy(None, None))
    go("all bf16, fp16 stream (round-3 headline mode)", Policy(BF, HF))
    go("all bf16, fp32 stream", Policy(BF, None))
    go("all fp16, fp32 stream", Policy(HF, None))
    base = go("all fp16, fp16 stream (= the base every piece below sits on)", Policy(HF, HF))
    b2 = base["rms_centred"] ** 2

    def bf_piece(name, **kw):
        s = go("bf16 only: " + name, Policy(HF, HF, [(lambda e, l, o, k, p=piece(**kw): k != "s" and p(e, l, o, k), BF)]))
        s["added_rms_centred"] = float(np.sqrt(max(s["rms_centred"] ** 2 - b2, 0.0)))
        return s

    for eng in ("vit", "med", "nlvr"):
        bf_piece(f"{eng}: everything", eng=eng)
    for eng in ("vit", "nlvr"):
        bf_piece(f"{eng}: weights only", eng=eng, kind="w")
        bf_piece(f"{eng}: activations only (a, o, p)", eng=eng, kind=("a", "o", "p"))
    ops = dict(vit=["patch", "qkv", "attn", "proj", "fc1", "fc2"], nlvr=["tokens", "qkv", "attn", "proj", "cq", "ckv", "cattn", "cproj", "fc1", "fc2", "cls"])
    for eng, names in ops.items():
        for op in names:
            bf_piece(f"{eng}: op {op}", eng=eng, op=op)
    if not quick:
        for eng in ("vit", "nlvr"):
            for lo in range(0, 12, 3):
                bf_piece(f"{eng}: layers {lo}-{lo + 2}", eng=eng, layer=(lo, lo + 1, lo + 2))
    # candidate mixed modes (what a cheap fix would look like)
    go("mixed A: bf16 everywhere, fp16 for nlvr layers 9-11 + cls", Policy(BF, HF, [(lambda e, l, o, k: k != "s" and e == "nlvr" and l >= 9, HF)]))
    go("mixed B: vit fp16, med + nlvr bf16", Policy(BF, HF, [(lambda e, l, o, k: k != "s" and e == "vit", HF)]))
    go("mixed C: vit bf16, med + nlvr fp16", Policy(HF, HF, [(lambda e, l, o, k: k != "s" and e == "vit", BF)]))
    go("mixed D: bf16 weights, fp16 activations (not an MFMA mode: attribution only)", Policy(HF, HF, [(lambda e, l, o, k: k == "w", BF)]))
    out = os.path.join(ROOT, "profiles", f"r4_precision_attribution_{which}{'_quick' if quick else ''}.json")
    json.dump(dict(fixture=which, queries=len(fx["refs"]) if max_q is None else max_q, threads=torch.get_num_threads(), rows=rows), open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
