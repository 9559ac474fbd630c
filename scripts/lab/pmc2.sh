#!/bin/bash
# usage: scripts/pmc2.sh <outdir> -- <python args>  : extra SQ counters (instruction mix / fetch / scalar)
set -u
out=gpurun_out/$1; shift; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p $out
i=0
for pmc in "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_WAVES" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $pmc --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1
done
python3 scripts/pmc_summary.py "$out"
