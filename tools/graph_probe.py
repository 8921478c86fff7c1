"""Small-batch latency: one query x K=100 from pixels, eager launches vs the same step replayed from a hipGraph
(torch.cuda.CUDAGraph capture of the ctypes kernel launches on the capture stream).  python tools/graph_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import config, ops, synthetic
from candidate_reranking_cir_amd.blip_stage1 import BLIP_Retrieval
from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR

dev = torch.device("cuda")
g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer()).to(dev).eval()
m1 = BLIP_Retrieval(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer()).to(dev).eval()
k = 100
for q_n in (1, 2, 4):
    images = torch.randn((q_n + q_n * k, 3, 224, 224), device=dev).bfloat16()
    ids = torch.stack([synthetic.caption_ids(q, 32) for q in range(q_n)]).to(dev)
    mask = torch.ones_like(ids)
    qidx = torch.arange(q_n, device=dev).repeat_interleave(k)

    def step():
        toks = m2.img_embed16(images)
        z = m1.z_t(toks[:q_n], ids, mask)
        return m2.score(z.last_hidden_state, ids, mask, toks[q_n:], qidx)

    for _ in range(3): ref = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    line = f"Q={q_n}: eager {eager*1e3:7.2f} ms/step ({q_n*k/eager:8.1f} triplets/s)"
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): step()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = step()
        gr.replay(); torch.cuda.synchronize()
        same = torch.equal(out, ref)
        t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 10
        line += f" | hipGraph {graph*1e3:7.2f} ms/step ({q_n*k/graph:8.1f} triplets/s) identical={same}"
    except Exception as e:  # noqa: BLE001
        line += f" | capture failed: {type(e).__name__}: {str(e)[:200]}"
    print(line, flush=True)
