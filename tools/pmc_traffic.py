"""Aggregates two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, separate runs of the same bench
command) into HBM-side bytes per GEMM launch -> profiles/<name>.json.  Units and the gfx950 correction follow
/opt/skills/guides/MI355X_MICROARCH.md: both counters are KiB; FETCH_SIZE reports half the bytes of wide coalesced reads.

  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_gemm_traffic.json"""
import csv
import glob
import json
import sys


def collect(d, counter):
    tot, n, by = 0.0, 0, {}
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "gemm" not in r["Kernel_Name"] or "cir" not in r["Kernel_Name"]:
                continue
            v = float(r["Counter_Value"]) * 1024.0
            tot += v
            n += 1
            k = "gemm256" if "gemm256" in r["Kernel_Name"] else "gemm128"
            e = by.setdefault(k, [0, 0.0])
            e[0] += 1
            e[1] += v
    return tot, n, by


def main():
    fdir, wdir, out = sys.argv[1:4]
    fb, fn, fby = collect(fdir, "FETCH_SIZE")
    wb, wn, wby = collect(wdir, "WRITE_SIZE")
    res = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 "
                  "--warmup 1 --no-cpu-baseline; all cir::gemm*_kernel launches (warm-up, timed and instrumented steps)",
        "launches_fetch_pass": fn, "launches_write_pass": wn,
        "fetch_bytes_per_launch_corrected_x2": 2.0 * fb / max(fn, 1),
        "write_bytes_per_launch": wb / max(wn, 1),
        "hbm_bytes_per_launch": 2.0 * fb / max(fn, 1) + wb / max(wn, 1),
        "by_kernel": {k: {"launches": v[0], "fetch_bytes_per_launch_x2": 2.0 * v[1] / v[0],
                          "write_bytes_per_launch": wby.get(k, [1, 0.0])[1] / max(wby.get(k, [1, 0.0])[0], 1)} for k, v in fby.items()},
        "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide "
                "coalesced reads); Infinity-Cache hits are counted",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
