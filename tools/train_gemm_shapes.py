"""Per-shape timing of the dense products of ONE training step (B = 16, 384 px): forward / dgrad GEMMs (cir_gemm_bias_act), the grouped
weight gradients, the fused attention launches (HIP events around each launch; GPU box only).   python tools/train_gemm_shapes.py [f16|bf16]"""
import collections, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import config, ops, synthetic, train_ops as T, train as TR
from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR

dt = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
dev = torch.device("cuda")
b, l, n_tok, d = 16, 32, 577, 768
g, v = config.BertGeometry(), config.VitGeometry(image_size=384)
m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer()).to(dev)
m2.set_compute_dtype(dt)
for n, p in m2.named_parameters():
    p.requires_grad_(not n.startswith("visual_encoder."))
m2.train()
ids = torch.stack([synthetic.caption_ids(q, l) for q in range(b)]).to(dev)
mask = torch.ones_like(ids)
z = torch.randn((b, l, d), device=dev)
feats = torch.randn((b, n_tok, d), device=dev)
gt = torch.arange(b, device=dev)

def step():
    logits = TR.fusion_train(m2, z, feats, ids, mask)
    F.cross_entropy(logits, gt).backward()
    for p in m2.parameters():
        p.grad = None

recs = []
def wrap(mod, name, key_fn):
    orig = getattr(mod, name)
    def timed(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = orig(*a, **kw); e1.record()
        recs.append((key_fn(*a, **kw), e0, e1))
        return out
    setattr(mod, name, timed)

step(); step(); torch.cuda.synchronize()
wrap(ops, "gemm", lambda a, w, *r, **kw: ("gemm x%d" % (a.shape[0] if a.dim() == 3 else 1), a.shape[-2], w.shape[-2], a.shape[-1], str(kw.get("out_dtype") or a.dtype)[6:],
                                           kw.get("residual") is not None, 2.0 * (a.shape[0] if a.dim() == 3 else 1) * a.shape[-2] * w.shape[-2] * a.shape[-1]))
wrap(T, "wgrad_grouped", lambda probs: ("wgrad_grouped", len(probs), 0, 0, "float32", False, sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _ in probs)))
wrap(T, "attention_train_fwd", lambda q, k, *r, **kw: ("attn_fwd", q.shape[0] * q.shape[1], q.shape[2], k.shape[2], "", False, 4.0 * q.shape[0] * q.shape[1] * q.shape[2] * k.shape[2] * 64))
wrap(T, "attention_train_bwd", lambda q, k, *r, **kw: ("attn_bwd", q.shape[0] * q.shape[1], q.shape[2], k.shape[2], "", False, 14.0 * q.shape[0] * q.shape[1] * q.shape[2] * k.shape[2] * 64))
for fn in ("residual_layernorm_train", "layernorm_bwd_fused", "gelu_bwd16", "colsum16", "eltwise"):
    wrap(T, fn, (lambda fn: lambda *a, **kw: (fn, a[0].shape[0], a[0].shape[-1], 0, "", False, 0.0))(fn))
step(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in recs:
    a = agg.setdefault(key[:6], [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += key[6]
tot = sum(a[1] for a in agg.values())
print(f"{'op':24s} {'M':>7} {'N':>5} {'K':>5} {'out':8s} res  calls  ms_total  us/call   TF/s  share")
for (op, m, n, k, od, res), (c, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{op:24s} {m:>7} {n:>5} {k:>5} {od:8s} {int(res):>3} {c:>6} {t:>9.2f} {t / c * 1e3:>8.1f} {fl / t / 1e9 if fl else 0:>7.1f} {t / tot:>6.3f}")
print(f"total instrumented ms {tot:.2f}")
