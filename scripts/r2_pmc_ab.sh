#!/bin/bash
# PMC comparison of the fp32 product kernels on the flagship: dynamic instruction mix and wait breakdown
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  out=gpurun_out/r2/pmc_$name; mkdir -p $out
  i=0
  for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" ; do
    i=$((i+1))
    ( export $(echo $envs | tr ',' ' '); timeout 150 rocprofv3 --pmc $pmc --output-format csv -d $out/p$i -- python3 bench.py --steps 30 --warmup 5 --settle-ms 0 --no-cpu-baseline > $out/p$i.log 2>&1 )
  done
  echo "=== $name"; python3 scripts/pmc_summary.py "$out" | grep -v "fixup\|tail_copy"
done
