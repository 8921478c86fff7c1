"""Throughput sweep over queries-per-step and ViT chunk size on one box (GPU box only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import config, ops, synthetic, weights, engine
from candidate_reranking_cir_amd.blip_stage1 import BLIP_Retrieval
from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR

dev = torch.device("cuda")
g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer()).to(dev).eval()
m1 = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer()).to(dev).eval()
k = 100
cases = [(16, 1616), (16, 808), (16, 404), (16, 539), (16, 270), (32, 1616), (32, 808)]
for rep in range(2):
    for q_n, chunk in cases:
        images = torch.randn((q_n + q_n * k, 3, 224, 224), device=dev).bfloat16()
        ids = torch.stack([synthetic.caption_ids(q, 32) for q in range(q_n)]).to(dev)
        mask = torch.ones_like(ids)
        qidx = torch.arange(q_n, device=dev).repeat_interleave(k)
        def step():
            toks = m2.engines()[0].forward(images, want32=False, chunk=chunk)[1]
            z = m1.z_t(toks[:q_n], ids, mask)
            return m2.score(z.last_hidden_state, ids, mask, toks[q_n:], qidx)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"Q={q_n:3d} chunk={chunk:5d}: {dt*1e3:8.2f} ms/step  {q_n*k/dt:9.1f} triplets/s", flush=True)
        del images
