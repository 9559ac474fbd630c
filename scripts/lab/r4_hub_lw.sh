#!/bin/bash
out=gpurun_out/r4/hub_lw.txt
mkdir -p gpurun_out/r4; : > $out
run() { echo "+ $*" >> $out; timeout -k 10 120 "$@" 2>&1 | grep -v MISMATCH >> $out; if grep -q "Memory access fault" $out; then echo "GPU fault"; cat $out; exit 1; fi; }
H=scripts/ubench/hub_gemm
for v in 4 5; do run $H 4 4096 256 $v 0 8 8 2; run $H 3 8192 512 $v 1 4 16 2; run $H 64 2048 256 $v 0 8 64 2; done
if grep -q WRONG $out; then echo "WRONG results"; cat $out; exit 1; fi
for v in 0 4 5; do
  run $H 72 262144 256 $v 1 32 256 3
  run $H 72 262144 512 $v 1 32 256 3
  run $H 1024 4096 512 $v 0 8 256 3
  run $H 16 1048576 512 $v 0 8 256 3
  run $H 16 1048576 512 $v 1 128 256 3
  HUB_SAME_A=1 HUB_SAME_B=1 run $H 72 262144 256 $v 1 32 256 3
done
cat $out
