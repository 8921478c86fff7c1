"""Runs only the ViT attention shape (B images x 12 heads, 197 x 197) a few times: the program to put behind rocprofv3 --pmc."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import ops
B, N, d = int(os.environ.get("B", 808)), int(os.environ.get("N", 197)), 768
qkv = torch.randn((B, N, 3, d), device="cuda").bfloat16()
out = torch.empty((B, N, d), device="cuda", dtype=torch.bfloat16)
for _ in range(int(os.environ.get("REPS", 6))):
    ops.attention(qkv[:, :, 0].unsqueeze(1), qkv[:, :, 1].unsqueeze(1), qkv[:, :, 2].unsqueeze(1), out.unsqueeze(1), 0.125)
torch.cuda.synchronize()
print("done")
