"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel totals normalised per benchmark step."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot/steps/1e6:.2f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    n = r["Name"]
    n = n[:70]
    print(f"{n:70s} calls/step={float(r['Calls'])/steps:7.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/steps/1e6:7.2f} {float(r['Percentage']):5.1f}%")
