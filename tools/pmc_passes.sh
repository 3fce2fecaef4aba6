#!/bin/bash
# Separate rocprofv3 --pmc passes over tools/prof_driver.py (one 2^18 G1 scalar-mul batch, one 2^16 pairing batch):
# issue / wait / instruction-cache / memory-instruction counters for the stall analysis in DESIGN.md.
# Usage (on the GPU box): bash tools/pmc_passes.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc_stall}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQ_INSTS_SALU SQ_INSTS_BRANCH" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU"
do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$ROOT/$OUT/pass$i" -o p -- python3 "$ROOT/tools/prof_driver.py" both > "$ROOT/$OUT/pass$i.log" 2>&1
  echo "pass $i done"
done
