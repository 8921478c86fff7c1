"""s_memtime stamps of the folded cross-attention kernel (diagnostic build `make -C candidate_reranking_cir_amd/csrc folddbg`, CIR_LIB=...folddbg8.so):
where a workgroup's cycles go - per wave 0 / wave 4 of workgroups 0-3."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CIR_LIB", os.path.join(ROOT, "candidate_reranking_cir_amd", "libcirrank_folddbg8.so"))
import numpy as np, torch
from candidate_reranking_cir_amd import ops, lib
T, L, N, D = 6720, 32, 197, 768
dt = torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda shape, s: (torch.randn(shape, generator=g, device="cuda") * s).to(dt)
q, x = r((2, T * L, D), 1.0), r((T, N, D), 1.0)
wk, wv = r((2, D, D), 0.03), r((2, D, D), 0.03)
bv = torch.randn((2, D), device="cuda")
wkt, wvp = ops.fold_pack_key(wk), ops.fold_pack_value(wv)
out = torch.empty((T, L, 2, D), dtype=dt, device="cuda")
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    ops.cross_attention_folded(q, x, wkt, wvp, bv, out, L, 0.125)
torch.cuda.synchronize()
c = lib.load()
buf = (ctypes.c_ulonglong * 512)()
c.cir_debug_fold_stamps.argtypes = [ctypes.c_void_p]
assert c.cir_debug_fold_stamps(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(4, 2, 64).astype(np.int64)
names = {0: "start(q staged)", 25: "phase1 end", 26: "softmax end", 51: "phase2 end"}
for k in range(4):
    names.update({1 + 6 * k: f"p1 c{k} top", 2 + 6 * k: f"p1 c{k} vmcnt0 done", 3 + 6 * k: f"p1 c{k} barrier done", 4 + 6 * k: f"p1 c{k} G1(h0) done",
                  5 + 6 * k: f"p1 c{k} G2(h0) done", 6 + 6 * k: f"p1 c{k} unit1 done"})
    names.update({27 + 6 * k: f"p2 c{k} top", 28 + 6 * k: f"p2 c{k} vmcnt0 done", 29 + 6 * k: f"p2 c{k} barrier done", 30 + 6 * k: f"p2 c{k} G3(h0) done",
                  31 + 6 * k: f"p2 c{k} unit0 done", 32 + 6 * k: f"p2 c{k} unit1 done"})
for wg in range(2):
    for wv_ in range(2):
        s = a[wg, wv_]
        print(f"--- workgroup {wg} wave {4 * wv_}: cycles since start (delta)")
        prev = s[0]
        for k in sorted(names):
            if s[k] == 0:
                continue
            print(f"   {names[k]:24s} {s[k] - s[0]:8d}  (+{s[k] - prev})")
            prev = s[k]
