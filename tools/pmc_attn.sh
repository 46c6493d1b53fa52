#!/bin/bash
# Issue-side counters of the attention kernel at two shapes (one counter pass per run; SQ block only):
#   bash tools/pmc_attn.sh   -> gpurun_out/pmc_attn_*.txt
set -e
export TMPDIR=/tmp
for shape in "8 384" "8 768"; do
  tag=$(echo $shape | tr ' ' 'x')
  rm -rf gpurun_out/pmc_attn_$tag
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_attn_$tag -o p -- python3 tools/attn_one.py $shape 6
  python3 tools/pmc_summary.py gpurun_out/pmc_attn_$tag > gpurun_out/pmc_attn_$tag.txt
  cat gpurun_out/pmc_attn_$tag.txt
done
