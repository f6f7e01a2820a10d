"""Steady-state kernel summary from a rocprofv3 --kernel-trace CSV: keeps the dispatches between the first
two hiast::confusion_kernel markers that bench.py emits around its timed region, groups by kernel name.
    python tools/trace_summary.py <kernel_trace.csv> <steps> > profiles/<name>.csv"""
import csv
import sys
from collections import defaultdict


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "confusion_kernel" in r["Kernel_Name"]]
    assert len(marks) >= 2, "markers not found"
    sel = rows[marks[0] + 1:marks[1]]
    agg = defaultdict(lambda: [0, 0])
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += d
    total = sum(v[1] for v in agg.values())
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls_per_step", "avg_us", "ms_per_step", "percent_of_gpu_busy"])
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k[:150], "%.2f" % (n / steps), "%.1f" % (d / n / 1e3), "%.3f" % (d / steps / 1e6), "%.2f" % (100.0 * d / total)])
    w.writerow(["TOTAL_gpu_busy", "", "", "%.3f" % (total / steps / 1e6), "100"])
    w.writerow(["TIMED_REGION_span", "", "", "%.3f" % (span / steps / 1e6), ""])


if __name__ == "__main__":
    main()
