"""Stand-alone timing of cir_gemm_bias_act on chosen shapes (GPU box only).
usage: python tools/gemm_bench.py [M N K [act out32 reps]]   (no args: the path's main shapes)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import lib, ops
TILE = int(os.environ.get("TILE", "0"))   # 0 auto | 128 | 256
lib.set_tuning(lib.TUNE_GEMM_TILE, TILE)

def run(m, n, k, act=0, out32=False, reps=20, res=False):
    a = (torch.randn((m, k), device="cuda")).bfloat16()
    w = (torch.randn((n, k), device="cuda") * 0.02).bfloat16()
    b = torch.randn((n,), device="cuda")
    r = torch.randn((m, n), device="cuda") if res else None
    out = torch.empty((m, n), device="cuda", dtype=torch.float32 if out32 else torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, b, residual=r, act=act, out_dtype=out.dtype, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(a, w, b, residual=r, act=act, out_dtype=out.dtype, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"M={m:7d} N={n:5d} K={k:5d} act={act} out32={int(out32)} res={int(res)} tile={TILE or 'auto'!s:>4s}: {us:9.1f} us  {2.0*m*n*k/us/1e6:8.1f} TF/s", flush=True)

if len(sys.argv) >= 4:
    m, n, k = map(int, sys.argv[1:4])
    act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    out32 = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
    res = bool(int(sys.argv[7])) if len(sys.argv) > 7 else out32
    run(m, n, k, act, out32, reps, res=res)
else:
    for shape in [(157600, 3072, 768, 0, False, False), (159176, 2304, 768, 0, False, False), (159176, 3072, 768, 1, False, False),
                  (159176, 768, 3072, 0, True, True), (159176, 768, 768, 0, True, True), (51200, 3072, 768, 1, False, False),
                  (51200, 768, 3072, 0, True, True), (25600, 768, 1536, 0, True, False), (8192, 8192, 8192, 0, False, False),
                  (4096, 4096, 4096, 0, False, False)]:
        m, n, k, act, o32, res = shape
        run(m, n, k, act, o32, 10, res)
