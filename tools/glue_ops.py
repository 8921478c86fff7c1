"""Which torch operators run inside one benchmark step (GPU box only): counts of aten ops that launch device work."""
import os, sys, collections
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
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total) for e in prof.key_averages()]
for key, cnt, t in sorted(rows, key=lambda r: -r[1])[:40]:
    if key.startswith("aten::") or "copy" in key.lower() or "Memcpy" in key:
        print(f"{key:50s} calls {cnt:5d}  device us {t:10.1f}")
