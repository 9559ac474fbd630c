#!/bin/bash
# usage: scripts/r2_h16_pmc.sh [bench args]: where the 16-bit flagship kernel waits -- counters only, one small group per pass
# (L2 hits / misses and requests from the L1s, L1 -> L2 read traffic, issue / wait cycles of the waves, busy time of the address and data units)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/r2/h16_pmc
mkdir -p $out
i=0
for grp in "TCC_HIT TCC_MISS TCC_REQ" "TCC_EA_RDREQ TCC_EA_RDREQ_32B TCC_EA_WRREQ TCC_EA_WRREQ_64B" "TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCC_WRITE_REQ" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "TA_BUSY TD_BUSY TCP_PENDING_STALL_CYCLES" "TCC_BUSY TCC_TAG_STALL TCC_EA_RDREQ_DRAM_CREDIT_STALL" \
           "TCP_TA_TCP_STATE_READ TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 bench.py --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -1 $out/p$i.log | cut -c1-160)"
done
python3 scripts/pmc_summary.py "$out"
