"""median duration per kernel name from a rocprofv3 --kernel-trace csv: python tools/dbg/kstats.py <csv> [substring ...]"""
import collections
import csv
import sys

d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if len(sys.argv) > 2 and not any(k in n for k in sys.argv[2:]):
        continue
    d.setdefault(n[:110], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort()
    print("%-112s n=%4d med %8.1f min %8.1f" % (k, len(v), v[len(v) // 2], v[0]))
