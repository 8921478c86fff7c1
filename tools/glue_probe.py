"""Which torch ops (glue, not libcirrank kernels) still run inside one benchmark step: torch.profiler table grouped by
op with the Python stack of each memcpy / elementwise launch.  GPU box only."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import config, ops, synthetic
from candidate_reranking_cir_amd.blip_stage1 import BLIP_Retrieval
from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR

q_n, k = int(os.environ.get("Q", 16)), 105
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

step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60, max_src_column_width=90))
