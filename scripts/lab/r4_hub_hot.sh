#!/bin/bash
out=gpurun_out/r4/hub_hot.txt
mkdir -p gpurun_out/r4; : > $out
run() { echo "+ $*" >> $out; timeout -k 10 120 "$@" 2>&1 | grep -v MISMATCH >> $out; if grep -q "Memory access fault" $out; then echo "GPU fault"; exit 1; fi; }
H=scripts/ubench/hub_gemm
for v in 0 2; do
  run $H 72 262144 256 $v 1 32 256 3
  HUB_SAME_A=1 run $H 72 262144 256 $v 1 32 256 3
  HUB_SAME_B=1 run $H 72 262144 256 $v 1 32 256 3
  HUB_SAME_A=1 HUB_SAME_B=1 run $H 72 262144 256 $v 1 32 256 3
  HUB_SAME_A=1 HUB_SAME_B=1 run ${H}_p1 72 262144 256 $v 1 32 256 3
  HUB_SAME_A=1 HUB_SAME_B=1 run ${H}_p2 72 262144 256 $v 1 32 256 3
done
cat $out
