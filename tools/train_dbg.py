"""Per-parameter gradient error table of the training step (train768 fixture) against torch autograd through the CPU oracle."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from tests import helpers as H
from candidate_reranking_cir_amd import synthetic
from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR
from candidate_reranking_cir_amd.train import NlvrTrainer
from oracle import cir_oracle as O

dtype = torch.float16 if "fp16" in sys.argv else torch.bfloat16
z = H.load("train768.npz")
g, v = H.geometry(json.loads(str(z["bert_cfg"])), json.loads(str(z["vit_cfg"])))
sd2, _ = H.state_dicts(g, v, int(z["seed"]), str(z["profile"]))
torch.set_num_threads(8)
w = {k: t.clone().float() for k, t in sd2.items()}
keys = [k for k in w if k.startswith(("text_encoder.", "cls_head.")) and w[k].is_floating_point()]
for k in keys: w[k].requires_grad_(True)
ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
zt, feats = torch.from_numpy(z["z_t"]), torch.from_numpy(z["feats"])
m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer()); m2.load_state_dict(sd2)
m2 = m2.cuda().float().set_compute_dtype(dtype)
tr = NlvrTrainer(m2, 0.0, 0.0)
mine = tr.forward(zt.cuda(), feats.cuda(), ids.cuda(), mask.cuda())
logits = O.img_txt_fusion_train(w, zt, feats, ids, mask, relu_mask=None if "nomask" in sys.argv else tr.head_mask().cpu())
loss = F.cross_entropy(logits, torch.arange(4)); loss.backward()
dl = torch.autograd.grad(F.cross_entropy(logits.detach().requires_grad_(True), torch.arange(4)), [])if False else None
lg = logits.detach().clone().requires_grad_(True); F.cross_entropy(lg, torch.arange(4)).backward(); dl = lg.grad
grads = tr.backward(dl.cuda())
print("logits err", (mine.cpu() - logits.detach()).abs().max().item())
rows = []
for k in keys:
    if w[k].grad is None or w[k].grad.norm().item() < 1e-6: continue
    r = w[k].grad; got = grads[k].cpu().view_as(r)
    rows.append(((got - r).norm().item() / (r.norm().item() + 1e-30), r.norm().item(), got.norm().item(), k))
rows.sort(reverse=True)
print('mean rel', np.mean([r[0] for r in rows]), 'n', len(rows))
for e, rn, gn, k in rows[:12]: print(f"{e:9.3e} ref {rn:9.3e} got {gn:9.3e} {k}")
k = "text_encoder.embeddings.word_embeddings.weight"
print("word emb norms ref", w[k].grad.double().norm().item(), "got", grads[k].double().norm().item())
r = w[k].grad; got = grads[k].cpu()
d = (got - r).norm(dim=1); top = torch.argsort(d, descending=True)[:8]
for i in top: print("row", int(i), "diff", d[i].item(), "ref", r[i].norm().item(), "got", got[i].norm().item(), "count in ids", int((ids == int(i)).sum()))
print("---- top of the network, in backward order")
order = ["cls_head.2.weight", "cls_head.0.bias", "cls_head.0.weight"]
p = "text_encoder.encoder.layer.11."
order += [p + n for n in ("output.LayerNorm.bias", "output.LayerNorm.weight", "output.dense.bias", "output.dense.weight", "intermediate.dense.bias", "intermediate.dense.weight",
                          "crossattention.output.LayerNormA.bias", "crossattention.output.LayerNormA.weight", "crossattention.output.LayerNormB.bias",
                          "crossattention.output.merge_layer.bias", "crossattention.output.merge_layer.weight",
                          "crossattention.output.dense0.bias", "crossattention.output.dense0.weight", "crossattention.self0.value.bias", "crossattention.self0.value.weight",
                          "crossattention.self0.query.weight", "crossattention.self0.key.weight", "attention.output.LayerNormA.bias", "attention.output.LayerNormA.weight",
                          "attention.output.dense0.bias", "attention.output.dense0.weight", "attention.self0.value.bias", "attention.self0.value.weight",
                          "attention.self0.query.weight", "attention.self0.key.weight", "attention.self1.value.weight")]
order += ["text_encoder.encoder.layer.10.output.LayerNorm.bias", "text_encoder.encoder.layer.10.output.dense.weight"]
err = {k: e for e, rn, gn, k in rows}
for k in order: print(f"{err.get(k, float('nan')):9.3e} {k}")
print("---- ReLU mask of cls_head.0")
with torch.no_grad():
    wd = {k: t.detach() for k, t in w.items()}
    idsE = ids.clone(); idsE[:, 0] = 30523
    hid = torch.cat([O.nlvr_forward(wd, idsE[i:i + 1].expand(4, -1), mask[i:i + 1].expand(4, -1), zt[i:i + 1].expand(4, -1, -1), feats) for i in range(4)])
    z1 = F.linear(hid, wd["cls_head.0.weight"], wd["cls_head.0.bias"])
mz1 = tr.sv["z1"].cpu()
print("z1 sigma", z1.std().item(), "max err", (mz1 - z1).abs().max().item(), "rms err", (mz1 - z1).pow(2).mean().sqrt().item(),
      "sign flips", int(((mz1 > 0) != (z1 > 0)).sum()), "of", z1.numel(), "hid err", (tr.sv["hid16"].float().cpu() - hid).abs().max().item())
