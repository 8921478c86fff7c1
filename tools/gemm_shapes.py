"""Per-shape GEMM timing of one benchmark step (HIP events on the launch stream). GPU box only.
   Q=16 python tools/gemm_shapes.py [out.json]    (the JSON carries shape, kernel instantiation, us, TF/s, fraction of peak)"""
import collections, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import config, ops, synthetic, weights
from candidate_reranking_cir_amd.blip_stage1 import BLIP_Retrieval
from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR

q_n, k = int(os.environ.get("Q", 8)), 105
dev = torch.device("cuda")
g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer()).to(dev).eval()
m1 = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer()).to(dev).eval()
images = torch.randn((q_n + q_n * k, 3, 224, 224), device=dev).bfloat16()
ids = torch.stack([synthetic.caption_ids(q, 32) for q in range(q_n)]).to(dev)
mask = torch.ones_like(ids)
qidx = torch.arange(q_n, device=dev).repeat_interleave(k)

def step():
    toks = m2.img_embed16(images)
    z = m1.z_t(toks[:q_n], ids, mask)
    return m2.score(z.last_hidden_state, ids, mask, toks[q_n:], qidx)

orig = ops.gemm
recs = []
def timed(a, w, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(a, w, *args, **kw); e1.record()
    nb = a.shape[0] if a.dim() == 3 else 1
    res = kw.get("residual")
    recs.append(((nb, a.shape[-2], w.shape[-2], a.shape[-1], kw.get("act", 0), str(out.dtype)[6:], res is not None,
                  None if res is None else str(res.dtype)[6:]), e0, e1))
    return out
step(); step(); torch.cuda.synchronize()
ops.gemm = timed
import candidate_reranking_cir_amd.engine as E
step(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in recs:
    t = e0.elapsed_time(e1)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += t
tot = sum(a[1] for a in agg.values())
print(f"{'nb':>2} {'M':>7} {'N':>5} {'K':>5} act out      calls   ms_total  us/call   TF/s   share")
rows = []
for (nb, m, n, kk, act, od, res, rd), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * nb * m * n * kk * c
    print(f"{nb:>2} {m:>7} {n:>5} {kk:>5} {act:>3} {od:8s} {c:>5} {t:>9.2f} {t/c*1e3:>8.1f} {fl/t/1e9:>7.1f} {t/tot:>6.3f}")
    rows.append({"batch": nb, "M": m, "N": n, "K": kk, "act": act, "out": od, "residual": res, "calls_per_step": c,
                 "kernel": ops.gemm_kernel_name(m, n, kk, nb, res, act, getattr(torch, od), torch.bfloat16,
                                                res_dtype=None if rd is None else getattr(torch, rd)),
                 "us_per_call": round(t / c * 1e3, 1), "tflops": round(fl / t / 1e9, 1), "frac_of_2516.6": round(fl / t / 1e9 / 2516.6, 4),
                 "share_of_gemm_time": round(t / tot, 4)})
print("total gemm ms", tot)
if len(sys.argv) > 1:
    json.dump({"workload": f"{q_n} queries x {k} candidates per step, 224 px, bf16 (HIP events per launch, one instrumented step)",
               "total_gemm_ms_per_step": round(tot, 2), "shapes": rows}, open(sys.argv[1], "w"), indent=1)
