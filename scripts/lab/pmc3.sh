#!/bin/bash
# usage: scripts/pmc3.sh <outdir> -- <python args>  : memory-pipeline stall counters (TA / TCP / TD / LDS FIFOs), counters only
set -u
out=gpurun_out/$1; shift; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p $out
i=0
# (TA_* / TD_* / TCP_*_sum sets made rocprofv3 7.2 abort with signal 6 on this pool and then hang until the timeout: left out)
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $pmc --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1
done
python3 scripts/pmc_summary.py "$out"
