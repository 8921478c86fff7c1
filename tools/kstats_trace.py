"""Per-step kernel statistics from a rocprofv3 --kernel-trace run of bench.py, counted over the TIMED steps only.

tools/kstats.py divides the whole run's totals (weight packing, warm-up, instrumented steps included) by a step count: fractional calls
per step and a total that is off by the launches outside the steps (round-5 review, weak 12).  Here the dispatches are cut into steps at
the step's first kernel (cir::patchify_kernel of the ViT's first chunk, i.e. a patchify launch whose predecessor is not part of a ViT
pass) and only steps [warmup, warmup + steps) are aggregated - the same launches bench.py's wall clock covers.

    python tools/kstats_trace.py <rocprof output dir> <warmup> <steps> [rows]
"""
import csv
import glob
import sys


def main():
    d, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    nrows = int(sys.argv[4]) if len(sys.argv) > 4 else 18
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
    # a step starts at a patchify launch that follows a kernel of the scoring tail (top-k of the previous step) or nothing of the path
    starts, in_vit = [], False
    for i, (_, _, name) in enumerate(rows):
        if "patchify_kernel" in name:
            if not in_vit:
                starts.append(i)
            in_vit = True
        elif "topk_desc_kernel" in name:
            in_vit = False
    if len(starts) < warmup + steps:
        raise SystemExit(f"found {len(starts)} steps in the trace, need {warmup + steps}")
    lo, hi = starts[warmup], (starts[warmup + steps] if len(starts) > warmup + steps else len(rows))
    # the timed region ends with the last step's top-k launches: cut the tail at the last topk before `hi`
    while hi > lo and "topk_desc_kernel" not in rows[hi - 1][2]:
        hi -= 1
    sel = rows[lo:hi]
    agg = {}
    for s, e, name in sel:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    span = sel[-1][1] - sel[0][0]
    print(f"timed steps {warmup}..{warmup + steps - 1}: {len(sel)} launches, kernel time per step {tot / steps / 1e6:.2f} ms, first-start-to-last-end span per step {span / steps / 1e6:.2f} ms")
    for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:nrows]:
        print(f"{name[:70]:70s} calls/step={n / steps:7.1f} avg_us={ns / n / 1e3:8.1f} ms/step={ns / steps / 1e6:7.2f} {100.0 * ns / tot:5.1f}%")


if __name__ == "__main__":
    main()
