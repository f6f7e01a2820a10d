#!/bin/bash
# busy / stall counters of the weight-gradient group kernel beside the tile kernel: separate rocprofv3 --pmc
# passes per counter group (no trace domains beside them), the program itself after `--`
#   tools/pmc_busy.sh [shape ...]        (run on the GPU box)
set -uo pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
out="gpurun_out/pmc_busy"
rm -rf "$out"; mkdir -p "$out"
shapes=("$@")
if [ ${#shapes[@]} -eq 0 ]; then shapes=(wgrad_group_l3 l3conv2_pl1 wgrad_l3conv2); fi
for s in "${shapes[@]}"; do
  case "$s" in wgrad_group*) filter=wgrad_group_kernel;; wgrad*) filter=wgrad_tn;; *) filter=igemm_bn_act;; esac
  for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_INSTS_SMEM"; do
    tag=$(echo "$c" | tr ' ' '_' | cut -c1-40)
    echo "== $s | $c" >> "$out/summary.txt"
    if timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d "$out/${s}_$tag" -o p -- python3 tools/pmc_igemm.py "$s" > "$out/${s}_$tag.log" 2>&1; then
      f=$(find "$out/${s}_$tag" -name "*counter_collection.csv" | head -1)
      python3 tools/pmc_summary.py "$f" "$filter" >> "$out/summary.txt" 2>&1
    else
      echo "   (pass failed: $(tail -1 "$out/${s}_$tag.log" | cut -c1-160))" >> "$out/summary.txt"
    fi
    rm -rf "$out/${s}_$tag"
  done
done
cat "$out/summary.txt"
