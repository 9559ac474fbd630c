#!/bin/bash
# usage: scripts/pmc.sh <outdir-under-gpurun_out> -- <python args...>   (runs separate --pmc passes; counters only, no traces)
set -u
out=gpurun_out/$1; shift; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p $out
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT TCC_MISS" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $pmc --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1
done
python3 scripts/pmc_summary.py "$out"
