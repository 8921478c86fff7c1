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
import time
lnf = int(os.environ.get("LNF", "0"))                 # 1: LayerNorm folded in (cir_gemm_ln_bias_act), 2: the same shape through the plain fp16 kernel
if lnf:
    a, w, out = a.half(), w.half(), out.half()
    wg, cs, bb = ops.ln_fold_pack(w.float(), b, torch.ones(k, device="cuda"), torch.zeros(k, device="cuda"))
def launch():
    if lnf == 1: ops.gemm_ln(a, wg, cs, bb, 1e-6, act, out=out)
    elif res: ops.gemm(a, w, b, residual=x, out_dtype=torch.float32, out=x)
    else: ops.gemm(a, w, b, act=act, out=out)
# >= 2 s of back-to-back launches first: the stamps then show the clock the part HOLDS under this load (MI355X guide, DVFS item 6)
t_end = time.time() + float(os.environ.get("WARM_S", "2.5"))
n_launch = 0
while time.time() < t_end:
    for _ in range(20): launch()
    torch.cuda.synchronize(); n_launch += 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): launch()
e1.record(); torch.cuda.synchronize()
print(f"M={m} N={n} K={k} act={act} res={res}: {e0.elapsed_time(e1) * 100:.1f} us per launch (stamped build, after {n_launch} warm launches), {2.0 * m * n * k / (e0.elapsed_time(e1) * 1e-4) / 1e6:.1f} TF/s")
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
# per-workgroup entry / exit stamps: spread of the persistent workgroups (tail) and the in-kernel clock
l.cir_debug_read_blocks.argtypes = [ctypes.c_void_p]
blk = np.zeros((512, 4), dtype=np.uint64)
l.cir_debug_read_blocks(blk.ctypes.data)
b = blk[:256].astype(np.int64)
cyc, rt = b[:, 2] - b[:, 0], (b[:, 3] - b[:, 1]) / 100.0       # shader cycles, microseconds (100 MHz)
t0, t1 = (b[:, 1] - b[:, 1].min()) / 100.0, (b[:, 3] - b[:, 1].min()) / 100.0
print(f"workgroups: life us min/med/max {rt.min():.1f}/{np.median(rt):.1f}/{rt.max():.1f}; entry spread {t0.max():.1f} us; exit min/med/max {t1.min():.1f}/{np.median(t1):.1f}/{t1.max():.1f} us")
print(f"in-kernel clock GHz min/med/max {(cyc / rt / 1e3).min():.3f}/{np.median(cyc / rt / 1e3):.3f}/{(cyc / rt / 1e3).max():.3f}")
for x in range(8):
    sel = np.arange(256) % 8 == x
    print(f"   blocks = {x} mod 8: exit med {np.median(t1[sel]):.1f} max {t1[sel].max():.1f} us, life med {np.median(rt[sel]):.1f}")
# end of each K-tile pair relative to the start of the main loop (stamp 3) and to the end (stamp 4)
l.cir_debug_read_pairs.argtypes = [ctypes.c_void_p]
pr = np.zeros((2, 64, 8), dtype=np.uint64)
l.cir_debug_read_pairs(pr.ctypes.data)
for grp in (0, 1):
    s3, s4 = buf[grp][:, 3].astype(np.int64), buf[grp][:, 4].astype(np.int64)
    pp = pr[grp].astype(np.int64)
    nt = int((s3 > 0).sum())
    npair = int((pp[2] > 0).sum())
    edges = np.concatenate([s3[2:nt - 1, None], pp[2:nt - 1, :npair], s4[2:nt - 1, None]], axis=1)
    print(f"wave group {grp}: cycles per K-tile pair (mean over tiles): " + " ".join(f"{x:.0f}" for x in np.diff(edges, axis=1).mean(0)))
