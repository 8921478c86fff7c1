"""Rounding-point emulation of the HIP path on the CPU (TEST INFRASTRUCTURE, like the rest of oracle/).

How much does the storage format of the residual stream add to the logit drift?  The reference's arithmetic (restated as in
cir_oracle.py) with 16-bit rounding inserted where the HIP path rounds: GEMM / attention operands and their 16-bit
outputs (`OP`), and - separately - the residual stream after every x + sublayer(x) and LayerNorm that feeds one (`RS`).
Benchmark geometry, `test` weights, 24 structured candidates.  Output recorded in
profiles/r2_secondary/residual_stream_emulation.json:

    python oracle/emulate_rounding.py
"""
import sys, math
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from candidate_reranking_cir_amd import config, synthetic, weights
from oracle import cir_oracle as O
torch.set_num_threads(8)

OP = {"dt": None}      # operand rounding dtype
RS = {"dt": None}      # residual-stream rounding dtype

def rq(x, dt):
    return x if dt is None else x.to(dt).float()

def lin(w, key, x):
    return F.linear(rq(x, OP["dt"]), rq(w[key + '.weight'], OP["dt"]), w[key + '.bias'])

def ln(w, key, x, eps):
    return F.layer_norm(x, (x.shape[-1],), w[key + ".weight"], w[key + ".bias"], eps)

def vit(w, image, prefix="visual_encoder.", eps=1e-6):
    pw = w[prefix + "patch_embed.proj.weight"]; d = pw.shape[0]; nh = d // 64
    x = F.conv2d(rq(image, OP["dt"]), rq(pw, OP["dt"]), w[prefix + "patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
    b = x.shape[0]
    x = torch.cat([w[prefix + "cls_token"].expand(b, -1, -1), x], dim=1) + w[prefix + "pos_embed"][:, : x.shape[1] + 1, :]
    x = rq(x, RS["dt"])
    for i in range(12):
        p = f"{prefix}blocks.{i}."
        y = ln(w, p + "norm1", x, eps)
        qkv = rq(lin(w, p + "attn.qkv", y), OP["dt"])
        n = qkv.shape[1]
        qkv = qkv.reshape(b, n, 3, nh, 64).permute(2, 0, 3, 1, 4)
        a = (qkv[0] @ qkv[1].transpose(-2, -1)) * 0.125
        a = rq(a.softmax(dim=-1), OP["dt"])
        y = rq((a @ qkv[2]).transpose(1, 2).reshape(b, n, d), OP["dt"])
        x = rq(x + lin(w, p + "attn.proj", y), RS["dt"])
        y = ln(w, p + "norm2", x, eps)
        f = rq(F.gelu(lin(w, p + "mlp.fc1", y)), OP["dt"])
        x = rq(x + lin(w, p + "mlp.fc2", f), RS["dt"])
    return ln(w, prefix + "norm", x, eps)

def sdpa(q, k, v, mask, nh):
    q, k, v = rq(q, OP["dt"]), rq(k, OP["dt"]), rq(v, OP["dt"])
    qh, kh, vh = O._heads(q, nh), O._heads(k, nh), O._heads(v, nh)
    s = qh @ kh.transpose(-1, -2) / 8.0
    if mask is not None: s = s + mask
    p = rq(torch.softmax(s, -1), OP["dt"])
    ctx = p @ vh
    b, h, t, dh = ctx.shape
    return rq(ctx.transpose(1, 2).reshape(b, t, h * dh), OP["dt"])

def med(w, ids, mask, enc, prefix="text_encoder.", eps=1e-12):
    h = rq(O.bert_embeddings(w, ids, prefix, eps), RS["dt"])
    sm = O.self_mask_additive(mask)
    for i in range(12):
        p = f"{prefix}encoder.layer.{i}."
        a = p + "attention.self."
        ctx = sdpa(lin(w, a + "query", h), lin(w, a + "key", h), lin(w, a + "value", h), sm, 12)
        h = rq(ln(w, p + "attention.output.LayerNorm", lin(w, p + "attention.output.dense", ctx) + h, eps), RS["dt"])
        c = p + "crossattention.self."
        ctx = sdpa(lin(w, c + "query", h), lin(w, c + "key", enc), lin(w, c + "value", enc), None, 12)
        h = rq(ln(w, p + "crossattention.output.LayerNorm", lin(w, p + "crossattention.output.dense", ctx) + h, eps), RS["dt"])
        f = rq(F.gelu(lin(w, p + "intermediate.dense", h)), OP["dt"])
        h = rq(ln(w, p + "output.LayerNorm", lin(w, p + "output.dense", f) + h, eps), RS["dt"])
    return h

def nlvr(w, ids, mask, z_t, cand, prefix="text_encoder.", eps=1e-12):
    emb = O.bert_embeddings(w, ids, prefix, eps)
    sm = O.self_mask_additive(mask)
    h = [rq(z_t, RS["dt"]), rq(emb, RS["dt"])]
    for i in range(12):
        p = f"{prefix}encoder.layer.{i}."
        att = []
        for b in (0, 1):
            s = f"{p}attention.self{b}."
            ctx = sdpa(lin(w, s + "query", h[b]), lin(w, s + "key", h[b]), lin(w, s + "value", h[b]), sm, 12)
            att.append(rq(ln(w, p + "attention.output.LayerNorm" + "AB"[b], lin(w, f"{p}attention.output.dense{b}", ctx) + h[b], eps), RS["dt"]))
        dd = []
        for b in (0, 1):
            s = f"{p}crossattention.self{b}."
            ctx = sdpa(lin(w, s + "query", att[b]), lin(w, s + "key", cand), lin(w, s + "value", cand), None, 12)
            dd.append(lin(w, f"{p}crossattention.output.dense{b}", ctx))
        mk = p + "crossattention.output.merge_layer"
        m = lin(w, mk, torch.cat(dd, -1)) if mk + ".weight" in w else (dd[0] + dd[1]) / 2
        m = rq(m, RS["dt"])
        x = [rq(ln(w, p + "crossattention.output.LayerNorm" + "AB"[b], m + att[b], eps), RS["dt"]) for b in (0, 1)]
        for b in (0, 1):
            f = rq(F.gelu(lin(w, p + "intermediate.dense", x[b])), OP["dt"])
            h[b] = rq(ln(w, p + "output.LayerNorm", lin(w, p + "output.dense", f) + x[b], eps), RS["dt"])
    hid = torch.cat([h[0][:, 0], h[1][:, 0]], -1)
    y = F.relu(lin(w, "cls_head.0", hid))
    return lin(w, "cls_head.2", y)[:, 0]

g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
sd2 = weights.synth_state_dict(weights.nlvr_param_spec(g, v), 21, "test")
sd1 = weights.synth_state_dict(weights.retrieval_param_spec(g, v), 22, "test")
k = 24
imgs = synthetic.scene_images(range(k + 1), 224)
ids = synthetic.caption_ids(0, 32)[None]; mask = torch.ones_like(ids)
def run(op, rs):
    OP["dt"], RS["dt"] = op, rs
    with torch.no_grad():
        f = vit(sd2, imgs)
        z = med(sd1, ids, mask, f[:1])
        return nlvr(sd2, ids.expand(k, -1), mask.expand(k, -1), z.expand(k, -1, -1), f[1:]).numpy()
ref = run(None, None)
print("sigma", ref.std())
for op, rs, name in [(torch.bfloat16, None, "bf16 operands, fp32 stream (current)"), (torch.bfloat16, torch.float16, "bf16 operands, fp16 stream"),
                     (torch.bfloat16, torch.bfloat16, "bf16 operands, bf16 stream"), (torch.float16, None, "fp16 operands, fp32 stream (current)"),
                     (torch.float16, torch.float16, "fp16 operands, fp16 stream")]:
    o = run(op, rs)
    e = o - ref
    print(f"{name:42s} max|d| {np.abs(e).max():.3e}  centred {np.abs(e - e.mean()).max():.3e}", flush=True)
