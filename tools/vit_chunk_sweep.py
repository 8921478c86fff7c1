"""ViT-B/16 forward over the benchmark's 6784 images at several chunk sizes (does a chunk whose residual stream fits the 256-MB
Infinity Cache make the LayerNorm / residual passes cheaper than the larger GEMM launches cost?).  One line per chunk size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from candidate_reranking_cir_amd import config, synthetic, weights
from candidate_reranking_cir_amd.blip_stage2 import BLIP_NLVR

dev = torch.device("cuda")
g, v = config.BertGeometry(), config.VitGeometry(image_size=224)
m2 = BLIP_NLVR(med_config=g, vit_geometry=v, tokenizer=synthetic.HashTokenizer())
m2.load_state_dict(weights.synth_state_dict(weights.nlvr_param_spec(g, v), 0, "test"))
m2 = m2.to(dev).eval().set_precision(sys.argv[1] if len(sys.argv) > 1 else "f16")
eng = m2.engines(text=False)[0]
n = 6784
images = torch.randn((n, 3, 224, 224), generator=torch.Generator(device=dev).manual_seed(1), device=dev).to(m2.token_dtype)
out = torch.empty((n, 197, 768), dtype=m2.token_dtype, device=dev)
for chunk in ([int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else (2048, 1132, 848, 680, 424, 340, 212)):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.forward(images, chunk=chunk, out16=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    parts = -(-n // chunk); size = -(-n // parts)
    print(f"chunk<= {chunk:5d} -> {parts:2d} parts of {size:4d} images (stream {size * 197 * 768 * 2 / 2**20:6.0f} MiB): {dt * 1e3:7.2f} ms  {n / dt:8.0f} img/s", flush=True)
