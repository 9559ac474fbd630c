#!/bin/bash
out=gpurun_out/r4/hub_ldb2.txt
mkdir -p gpurun_out/r4; : > $out
run() { echo "+ $* (HUB_LDB=$HUB_LDB)" >> $out; timeout -k 10 200 "$@" 2>&1 | grep -v MISMATCH >> $out; if grep -q "Memory access fault" $out; then echo "GPU fault"; cat $out; exit 1; fi; }
H=scripts/ubench/hub_gemm
for v in 10 0; do
  if [ $v = 10 ]; then T=36; else T=72; fi
  HUB_LDB=1048640 run $H $T 262144 512 $v 1 32 256 3
  HUB_LDB=1050624 run $H $T 262144 512 $v 1 32 256 3
  HUB_LDB=1048640 run $H $T 1048576 512 $v 1 64 256 3
  run $H $T 1048576 512 $v 1 64 256 3
done
cat $out
