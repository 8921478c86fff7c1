import torch, sys
sys.path.insert(0, '/root/repo')
from candidate_reranking_cir_amd import ops, lib
torch.manual_seed(0)
for (m, n, k) in ((130, 144, 128), (64, 128, 64), (128, 128, 128)):
    a = torch.randint(-3, 4, (m, k)).to(torch.bfloat16).cuda()
    w = torch.randint(-3, 4, (n, k)).to(torch.bfloat16).cuda()
    bias = torch.randint(-5, 6, (n,)).float().cuda()
    lib.set_tuning(lib.TUNE_GEMM_TILE, 128)
    out = ops.gemm(a, w, bias, out_dtype=torch.float32)
    ref = a.float() @ w.float().T + bias
    d = (out - ref)
    bad = d.nonzero()
    print((m, n, k), "mismatches", len(bad), "of", out.numel())
    if len(bad):
        rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
        print(" rows", rows[:40], "cols", cols[:60])
        print(" diff sample", d[bad[0, 0], bad[0, 1]].item(), "bias there", bias[bad[0, 1]].item(), "vals", out[bad[0,0], bad[0,1]].item(), ref[bad[0,0], bad[0,1]].item())
        # is diff == -bias or some other bias?
        print(" diff == -bias[col]:", bool((d[bad[:,0], bad[:,1]] == -bias[bad[:,1]]).all()))
    out0 = ops.gemm(a, w, None, out_dtype=torch.float32)
    print("  no-bias mismatches", int(((out0 - a.float() @ w.float().T) != 0).sum()))
# attention two-pass vs reference
from candidate_reranking_cir_amd.lib import TUNE_ATTN_TWO_PASS
for dt in (torch.float16, torch.bfloat16):
    for lq in (197, 50, 224, 100):
        b = 5
        qkv = (torch.randn((b, lq, 3, 768)) * 1.0).to(dt).cuda()
        outs = {}
        for mode in (-1, 0, 1):
            lib.set_tuning(TUNE_ATTN_TWO_PASS, mode)
            ctx = torch.empty((b, lq, 768), dtype=dt, device="cuda")
            ops.attention(qkv[:, :, 0].unsqueeze(1), qkv[:, :, 1].unsqueeze(1), qkv[:, :, 2].unsqueeze(1), ctx.unsqueeze(1), 0.125)
            outs[mode] = ctx.float()
        lib.set_tuning(TUNE_ATTN_TWO_PASS, 0)
        q, k, v = (qkv[:, :, i].float().reshape(b, lq, 12, 64).transpose(1, 2) for i in range(3))
        ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(b, lq, 768)
        print(dt, lq, "online err", (outs[-1] - ref).abs().max().item(), "two-pass split", (outs[0] - ref).abs().max().item(),
              "two-pass single", (outs[1] - ref).abs().max().item(), "nan", bool(torch.isnan(outs[0]).any()))
