"""Steady-state kernel summary from a rocprofv3 --kernel-trace CSV: keeps the dispatches between the first
two hiast::confusion_kernel markers that bench.py emits around its timed region, groups by kernel name AND launch shape
(grid size in workgroups: a template instance that runs on layer3 and layer4 maps is two rows, not one mixed average).
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
        # rocprofv3 reports the grid in work-items: workgroups = grid / workgroup size per dimension
        try:
            wg = [max(1, int(r["Grid_Size_" + c]) // max(1, int(r["Workgroup_Size_" + c]))) for c in "XYZ"]
            shape = "x".join(str(v) for v in wg if v != 1) or "1"
        except (KeyError, ValueError):
            shape = "?"
        a = agg[(r["Kernel_Name"], shape)]
        a[0] += 1
        a[1] += d
    total = sum(v[1] for v in agg.values())
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "workgroups", "calls_per_step", "avg_us", "ms_per_step", "percent_of_gpu_busy"])
    for (k, shape), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k[:150], shape, "%.2f" % (n / steps), "%.1f" % (d / n / 1e3), "%.3f" % (d / steps / 1e6), "%.2f" % (100.0 * d / total)])
    w.writerow(["TOTAL_gpu_busy", "", "", "", "%.3f" % (total / steps / 1e6), "100"])
    w.writerow(["TIMED_REGION_span", "", "", "", "%.3f" % (span / steps / 1e6), ""])
    w.writerow(["DISPATCHES_per_step", "", "%.1f" % (len(sel) / steps), "", "", ""])


if __name__ == "__main__":
    main()
