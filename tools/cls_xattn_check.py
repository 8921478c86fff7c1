"""cir_cls_cross_attention against an fp32 torch reference + timing (GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import ops
torch.manual_seed(0)
for (T, N, D, dt) in [(5, 197, 768, torch.bfloat16), (7, 17, 128, torch.bfloat16), (3, 577, 768, torch.float16), (6720, 197, 768, torch.bfloat16)]:
    x = torch.randn((T, N, D), device="cuda").to(dt)
    qp = (torch.randn((T, 32, D), device="cuda") * 0.3).to(dt)
    out = ops.cls_cross_attention(x, qp, 0.125)
    torch.cuda.synchronize()
    if T <= 16:
        s = torch.einsum("trd,tnd->trn", qp.float(), x.float()) * 0.125
        ref = torch.softmax(s, -1) @ x.float()
        err = (out.float() - ref).abs().max().item()
        print(f"T={T} N={N} D={D} {dt}: max|err| {err:.3e} (ref absmax {ref.abs().max().item():.2f}) nan {torch.isnan(out).sum().item()}")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): ops.cls_cross_attention(x, qp, 0.125, out=out)
        e0.record()
        for _ in range(10): ops.cls_cross_attention(x, qp, 0.125, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"T={T} N={N} D={D}: {us:.1f} us  {T * N * D * 2 / us / 1e6:.2f} TB/s of tokens")
