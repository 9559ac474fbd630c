#!/bin/bash
# lab: launches and times of the small real products at N = 128 with the fused reduction on / off, fresh and prepared B
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for fr in 1 0; do
  export SPARTA_SP_FUSED_REDUCE=$fr
  python scripts/lab/r4_real_n128.py 2>&1 | grep -v "Warning\|amdgpu.ids" | sed "s/^/fused_reduce=$fr  /"
done
