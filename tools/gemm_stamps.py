"""Diagnostic: per-tile phase shares of gemm256 from in-kernel s_memtime stamps.
Build the stamped library first: make -C candidate_reranking_cir_amd/csrc stamp
  python tools/gemm_stamps.py M N K [act] [residual]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CIR_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "candidate_reranking_cir_amd", "libcirrank_stamp.so"))
from candidate_reranking_cir_amd import ops, lib
m, n, k = (int(x) for x in sys.argv[1:4])
act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
res = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # 1: fp32 out + fp32 residual in place (proj / fc2 epilogue)
a = torch.randn((m, k), device="cuda").bfloat16(); w = (torch.randn((n, k), device="cuda") * 0.02).bfloat16(); b = torch.randn((n,), device="cuda")
out = torch.empty((m, n), device="cuda", dtype=torch.bfloat16)
x = torch.randn((m, n), device="cuda") if res else None
for _ in range(3):
    if res: ops.gemm(a, w, b, residual=x, out_dtype=torch.float32, out=x)
    else: ops.gemm(a, w, b, act=act, out=out)
torch.cuda.synchronize()
l = lib.load(); l.cir_debug_read_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros((2, 64, 8), dtype=np.uint64)
l.cir_debug_read_stamps(buf.ctypes.data)
names = ["wait_operands", "acc_init", "sync_start", "main_loop", "trail_sync", "setup+prologue_issue", "epilogue", "loop_back(next tile stamp0)"]
for grp in (0, 1):
    s = buf[grp].astype(np.int64)
    nt = int((s[:, 0] > 0).sum())
    d = np.zeros((nt - 1, 8))
    for t in range(nt - 1):
        for i in range(7): d[t, i] = s[t, i + 1] - s[t, i]
        d[t, 7] = s[t + 1, 0] - s[t, 7]
    print(f"wave group {grp}: {nt} tiles; mean cycles (100 MHz memtime ticks x?) per segment, tiles 2..{nt-2}:")
    mid = d[2:-1]
    tot = mid.sum(1).mean()
    for i, nm in enumerate(names): print(f"   {nm:32s} {mid[:, i].mean():9.0f}  {100*mid[:, i].mean()/tot:5.1f}%")
    print(f"   total per tile {tot:9.0f}")
