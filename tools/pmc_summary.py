"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch.
    python tools/pmc_summary.py <counter_collection.csv> [name-filter]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
