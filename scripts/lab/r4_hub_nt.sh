#!/bin/bash
out=gpurun_out/r4/hub_nt.txt
mkdir -p gpurun_out/r4; : > $out
run() { echo "+ $*" >> $out; timeout -k 10 200 "$@" 2>&1 | grep -v MISMATCH >> $out; if grep -q "Memory access fault" $out; then echo "GPU fault"; cat $out; exit 1; fi; }
export HUB_LDB=1048640
for r in 1 2; do
for H in scripts/ubench/hub_gemm scripts/ubench/hub_gemm_nt; do
  run $H 36 1048576 512 10 1 64 256 3
  run $H 36 1048576 256 10 1 64 256 3
done
done
cat $out
