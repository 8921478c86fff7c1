"""Where do the integer-exact GEMM results differ? (debug aid, GPU box only)  python tools/gemm_dbg.py M N K [out32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import lib, ops
m, n, k = (int(x) for x in sys.argv[1:4])
out32 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
lib.set_tuning(lib.TUNE_GEMM_TILE, 256)
g = torch.Generator(device="cpu").manual_seed(m * 7 + n)
a = torch.randint(-3, 4, (m, k), generator=g).to(torch.bfloat16).cuda()
w = torch.randint(-3, 4, (n, k), generator=g).to(torch.bfloat16).cuda()
bias = torch.randint(-5, 6, (n,), generator=g).float().cuda()
ref = a.float() @ w.float().T + bias
for rep in range(3):
    out = ops.gemm(a, w, bias, out_dtype=torch.float32 if out32 else torch.bfloat16).float()
    torch.cuda.synchronize()
    bad = (out != ref).nonzero()
    print(f"rep {rep}: {bad.shape[0]} mismatches of {m * n}")
    if bad.shape[0]:
        rows, cols = bad[:, 0].unique().tolist(), bad[:, 1].unique().tolist()
        print("  rows", rows[:40], "..." if len(rows) > 40 else "")
        print("  cols", cols[:40], "..." if len(cols) > 40 else "", "n cols", len(cols))
        r, c = bad[0].tolist()
        print("  first", r, c, out[r, c].item(), ref[r, c].item())
