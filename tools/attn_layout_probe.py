"""Is the ViT attention bound by the access pattern of its Q / K / V reads?  Same kernels, same bytes, two layouts of the fused QKV
tensor: row-major (M, 2304) as the QKV GEMM writes it today (a head's K rows are 128-byte pieces 4608 bytes apart) and HEAD-MAJOR
(36, M, 64) (a head's K for one image = 25 KB contiguous).  The head-major case runs through the existing ABI by passing the 12
heads as the B0 axis with H = 1.  (Round 4 also ran a two-pass kernel through this probe: profiles/r4_secondary/attn_layout_probe.txt.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from candidate_reranking_cir_amd import ops

dev = torch.device("cuda")
b, n, h = 1696, 197, 12
m = b * n
for dt in (torch.float16, torch.bfloat16):
    rm = torch.randn((b, n, 3, h * 64), device=dev).to(dt)                       # row-major fused QKV
    hm = rm.view(m, 3 * h, 64).permute(1, 0, 2).contiguous()                    # (36, M, 64) head-major slabs
    ctx = torch.empty((b, n, h * 64), dtype=dt, device=dev)
    def run_rm():
        ops.attention(rm[:, :, 0].unsqueeze(1), rm[:, :, 1].unsqueeze(1), rm[:, :, 2].unsqueeze(1), ctx.unsqueeze(1), 0.125)
    hm4 = hm.view(3, h, b, n, 64)                                               # [q|k|v][head][image][token][64]
    q, k, v = (hm4[i].permute(1, 0, 2, 3) for i in range(3))                    # (image, head, token, 64) views
    ctx2 = torch.empty((b, n, h * 64), dtype=dt, device=dev)
    o = ctx2.view(b, n, h, 64).permute(0, 2, 1, 3)                              # (image, head, token, 64) view of the row-major ctx
    def run_hm():
        ops.attention(q, k, v, o, 0.125)
    for name, fn in (("row-major", run_rm), ("head-major", run_hm)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
        print(f"{str(dt)[6:]:9s} {name:10s} {us:7.1f} us  ({(m * 2304 * 2 + m * 768 * 2) / us / 1e6:.2f} TB/s)", flush=True)
    print("  max |row-major - head-major| =", (ctx.float() - ctx2.float()).abs().max().item())
