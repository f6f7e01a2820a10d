#!/bin/bash
# rocprofv3 kernel trace of the default bench.py run -> steady-state per-kernel table + critical-path summary
#   tools/prof_bench.sh <tag> [extra bench args]      (run on the GPU box from the repo root)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
tag="$1"; shift || true
# a tag that ends in "serial": every part of the step on ONE stream (true per-kernel durations, nothing co-running)
case "$tag" in *serial) export HIAST_BENCH_SERIAL=1 HIAST_NO_SIDE_STREAM=1 HIAST_NO_WGRAD_STREAM=1 HIAST_EVAL_SPLIT=1 HIAST_BENCH_PL_STREAM=0;; esac
out="gpurun_out/prof_${tag}"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
steps=8
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rp" -o bench -- python3 bench.py --steps $steps --warmup 3 --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
csv=$(find "$out/rp" -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py "$csv" $steps > "$out/steady_kernels.csv"
python3 tools/trace_critical.py "$csv" $steps > "$out/critical_path.txt" 2>&1 || true
stats=$(find "$out/rp" -name '*kernel_stats.csv' | head -1)
cp "$stats" "$out/kernel_stats_incl_warmup.csv"
rm -rf "$out/rp"
head -40 "$out/steady_kernels.csv"
cut -c1-300 "$out/bench.json" | head -3
