"""Per-step kernel statistics of `bench.py --mode train` from a rocprofv3 --kernel-trace run: steps are cut at the optimizer's update kernel
(one cir::adamw_dev_kernel launch per step), the untimed legs-instrumented steps are left out, and for every step the GPU's busy time, its
idle time between launches and the per-kernel totals are listed.

    python tools/train_trace.py <rocprof output dir> <first step> <steps> [rows]
"""
import csv
import glob
import sys


def main():
    d, first, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    nrows = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
    ends = [i for i, r in enumerate(rows) if "adamw_dev_kernel" in r[2] or "adamw_kernel" in r[2]]
    if len(ends) < first + steps:
        raise SystemExit(f"found {len(ends)} optimizer steps in the trace, need {first + steps}")
    lo, hi = ends[first - 1] + 1, ends[first + steps - 1] + 1
    sel = rows[lo:hi]
    span = (sel[-1][1] - sel[0][0]) / steps
    busy, t_end, idle = 0, sel[0][0], 0
    agg = {}
    for s, e, name in sel:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
        if s > t_end:
            idle += s - t_end
        busy += e - max(s, t_end) if e > t_end else 0
        t_end = max(t_end, e)
    tot = sum(a[1] for a in agg.values())
    print(f"steps {first}..{first + steps - 1}: {len(sel) / steps:.0f} launches per step, span {span / 1e6:.2f} ms per step, kernel time {tot / steps / 1e6:.2f} ms, "
          f"GPU busy {busy / steps / 1e6:.2f} ms, idle between launches {idle / steps / 1e6:.2f} ms")
    for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:nrows]:
        print(f"{name[:110]:110s} calls/step={c / steps:7.1f} avg_us={t / c / 1e3:8.1f} ms/step={t / steps / 1e6:6.2f} {t / tot * 100:5.1f}%")


main()
