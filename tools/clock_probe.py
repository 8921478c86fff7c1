"""Samples the shader clock and package power (rocm-smi) while a GEMM loop runs: vendor GEMM vs cir_gemm_bias_act.
Calibration only.  python tools/clock_probe.py"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candidate_reranking_cir_amd import ops  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10)
            out.append(r.stdout.strip().replace("\n", " | "))
        except Exception as e:  # noqa: BLE001
            out.append(f"err {e}")
        time.sleep(0.5)


def run(name, fn, seconds=4.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    el = time.time() - t0
    stop.set()
    th.join()
    print(f"== {name}: {n / el:.1f} calls/s", flush=True)
    for line in out[1:6]:
        print("   ", line[:400], flush=True)
    return n / el


def main():
    dt = torch.bfloat16
    for (m, n, k) in ((8192, 8192, 8192), (318352, 3072, 768)):
        a = (torch.randn((m, k), device="cuda") * 0.5).to(dt)
        w = (torch.randn((n, k), device="cuda") * 0.02).to(dt)
        out = torch.empty((m, n), dtype=dt, device="cuda")
        fl = 2.0 * m * n * k
        r1 = run(f"hipBLASLt {m}x{n}x{k}", lambda: torch.matmul(a, w.t(), out=out))
        r2 = run(f"cir       {m}x{n}x{k}", lambda: ops.gemm(a, w, None, out=out))
        print(f"TF/s: hipBLASLt {fl * r1 / 1e12:.0f}  cir {fl * r2 / 1e12:.0f}", flush=True)
    print(subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout[-1500:])


if __name__ == "__main__":
    main()
