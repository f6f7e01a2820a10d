#!/bin/bash
# separate rocprofv3 --pmc passes per shape and counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)
#   tools/pmc_igemm.sh [shape ...]        (run on the GPU box)
set -euo pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
out="gpurun_out/pmc_igemm"
rm -rf "$out"; mkdir -p "$out"
shapes=("$@")
if [ ${#shapes[@]} -eq 0 ]; then shapes=(l3conv2_pl2 l4conv2_pl2 l3conv3_pl2 l3conv1_pl2 l3conv2_pl1); fi
for s in "${shapes[@]}"; do
  case "$s" in wgrad_group*) filter=wgrad_group_kernel;; wgrad*) filter=wgrad_tn;; l3conv3_pl1) filter=xconv_kernel;; l3conv3_pl2) filter=xconv2_kernel;; *) filter=igemm_bn_act;; esac
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo "$c" | tr ' ' '_' | cut -c1-20)
    rocprofv3 --pmc $c --output-format csv -d "$out/${s}_$tag" -o p -- python3 tools/pmc_igemm.py "$s" > "$out/${s}_$tag.log" 2>&1
    f=$(find "$out/${s}_$tag" -name "*counter_collection.csv" | head -1)
    echo "== $s | $c" >> "$out/summary.txt"
    python3 tools/pmc_summary.py "$f" "$filter" >> "$out/summary.txt" 2>&1
    rm -rf "$out/${s}_$tag"
  done
  grep "^shape" "$out/${s}_FETCH_SIZE.log" >> "$out/summary.txt"
done
cat "$out/summary.txt"
