#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python command; prints the per-kernel summary
#   tools/prof_cmd.sh <tag> <python script and args...>
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
tag="$1"; shift
out="gpurun_out/prof_${tag}"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rp" -o p -- python3 "$@" > "$out/stdout.txt" 2> "$out/stderr.txt"
stats=$(find "$out/rp" -name '*kernel_stats.csv' | head -1)
cp "$stats" "$out/kernel_stats.csv"
rm -rf "$out/rp"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print("%-90s calls %6s avg %9.1f us total %9.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
