"""Stand-alone timing of cir_attention on the path's four shapes (GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import lib, ops

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

d = 768
# ViT: B images, N=197, packed qkv
for B, N in [(1696, 197), (808, 197), (202, 197), (64, 577), (404, 577)]:
    qkv = torch.randn((B, N, 3, d), device="cuda").bfloat16()
    out = torch.empty((B, N, d), device="cuda", dtype=torch.bfloat16)
    for name, cap in (("shared", 608), ("stream", 32)):
        lib.set_tuning(lib.TUNE_ATTN_SHARED_MAX, cap)
        us = timeit(lambda: ops.attention(qkv[:, :, 0].unsqueeze(1), qkv[:, :, 1].unsqueeze(1), qkv[:, :, 2].unsqueeze(1), out.unsqueeze(1), 0.125))
        fl = 4.0 * B * 12 * N * N * 64
        gb = B * N * 4 * d * 2 / 1e9
        print(f"vit    B={B:4d} N={N} {name:15s}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {gb/us*1e3:6.2f} TB/s (q,k,v,out once)", flush=True)
lib.set_tuning(lib.TUNE_ATTN_SHARED_MAX, 0)
T, L, N = 1600, 32, 197
qb = torch.randn((2, T, L, d), device="cuda").bfloat16()
kv = torch.randn((T, N, 4, d), device="cuda").bfloat16()
cc = torch.empty((T, L, 2, d), device="cuda", dtype=torch.bfloat16)
us = timeit(lambda: ops.attention(qb.permute(1, 0, 2, 3), kv[:, :, 0::2].permute(0, 2, 1, 3), kv[:, :, 1::2].permute(0, 2, 1, 3), cc.permute(0, 2, 1, 3), 0.125))
print(f"cross  T={T} L={L} N={N}: {us:8.1f} us  {4.0*T*2*12*L*N*64/us/1e6:7.1f} TF/s")
qkv = torch.randn((2, T, L, 3 * d), device="cuda").bfloat16()
ctx = torch.empty((2, T, L, d), device="cuda", dtype=torch.bfloat16)
mask = torch.zeros((T, L), device="cuda")
us = timeit(lambda: ops.attention(qkv[..., :d], qkv[..., d:2*d], qkv[..., 2*d:], ctx, 0.125, mask.unsqueeze(0).expand(2, T, L)))
print(f"self   T={T} L={L}: {us:8.1f} us  {4.0*T*2*12*L*L*64/us/1e6:7.1f} TF/s")
