"""per-shape durations of the weight-gradient kernels from a rocprofv3 --kernel-trace csv of tools/ab_wgrad.py"""
import collections
import csv
import sys

d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "wgrad" not in n:
        continue
    key = (n[:44], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""))
    d.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort()
    print("%-46s grid %6s %4s n=%3d med %7.1f min %7.1f" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
