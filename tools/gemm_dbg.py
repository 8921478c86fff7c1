"""Where do the integer-exact GEMM results differ? (debug aid, GPU box only)  python tools/gemm_dbg.py M N K [out32] [act]
Operands in {-1, 0, 1}: every partial sum is an exactly representable integer also in a 16-bit output (|sum| <= 256)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import lib, ops
m, n, k = (int(x) for x in sys.argv[1:4])
out32 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
act = int(sys.argv[5]) if len(sys.argv) > 5 else 0
lib.set_tuning(lib.TUNE_GEMM_TILE, 256)
g = torch.Generator(device="cpu").manual_seed(m * 7 + n)
a = torch.randint(-1, 2, (m, k), generator=g).to(torch.bfloat16).cuda()
w = torch.randint(-1, 2, (n, k), generator=g).to(torch.bfloat16).cuda()
bias = torch.randint(-5, 6, (n,), generator=g).float().cuda()
ref = a.float() @ w.float().T + bias
if act == 2: ref = ref.clamp_min(0)
for rep in range(2):
    out = ops.gemm(a, w, bias, act=act, out_dtype=torch.float32 if out32 else torch.bfloat16).float()
    torch.cuda.synchronize()
    bad = (out != ref).nonzero()
    print(f"rep {rep}: {bad.shape[0]} mismatches of {m * n}")
    if bad.shape[0]:
        rows, cols = bad[:, 0].unique().tolist(), bad[:, 1].unique().tolist()
        print("  rows", rows[:48], "..." if len(rows) > 48 else "", "n rows", len(rows))
        print("  cols", cols[:48], "..." if len(cols) > 48 else "", "n cols", len(cols))
        r, c = bad[0].tolist()
        print("  first", r, c, out[r, c].item(), ref[r, c].item())
        # is it a permutation of rows? find where ref row r went
        for rr in rows[:4]:
            hit = (out == ref[rr]).all(1).nonzero().flatten().tolist()
            print(f"  ref row {rr} appears as out rows {hit[:6]}")
