#!/bin/bash
# HBM-side bytes of EVERY kernel of the bench step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one
# pass on gfx950) over the default bench run, summed per kernel name.   tools/pmc_bench.sh <tag>     (on the GPU box)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
tag="${1:-pmc}"
out="gpurun_out/pmcb_${tag}"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
# serial streams: a dispatch's counters are not polluted by a kernel running beside it
export HIAST_BENCH_SERIAL=1 HIAST_NO_SIDE_STREAM=1 HIAST_NO_WGRAD_STREAM=1 HIAST_EVAL_SPLIT=1 HIAST_BENCH_PL_STREAM=0
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/$c" -o p -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline > "$out/$c.json" 2> "$out/$c.err"
  f=$(find "$out/$c" -name "*counter_collection.csv" | head -1)
  cp "$f" "$out/$c.csv"
  rm -rf "$out/$c"
done
python3 tools/pmc_bench_summary.py "$out/FETCH_SIZE.csv" "$out/WRITE_SIZE.csv" 4 > "$out/summary.txt"
head -50 "$out/summary.txt"
