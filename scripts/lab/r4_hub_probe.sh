#!/bin/bash
# lab: which stream bounds the hub kernel (timing-only probe builds of scripts/ubench/hub_gemm.hip)
out=gpurun_out/r4/hub_probe.txt
mkdir -p gpurun_out/r4; : > $out
run() { echo "+ $*" >> $out; timeout -k 10 120 "$@" 2>&1 | grep -v MISMATCH >> $out; if grep -q "Memory access fault" $out; then echo "GPU fault"; exit 1; fi; }
H=scripts/ubench/hub_gemm
# correctness on small shapes first
for v in 0 1 2 3; do run $H 4 4096 256 $v 0 8 8 2; run $H 3 8192 512 $v 1 4 16 2; run $H 64 2048 256 $v 0 8 64 2; done
if grep -q WRONG $out; then echo "WRONG results"; grep -B1 WRONG $out | head; exit 1; fi
for v in 0 1; do
  for pb in 1 2 3; do run ${H}_p$pb 72 262144 256 $v 1 32 256 3; done
  run $H 72 262144 256 $v 1 32 256 3
  run $H 72 262144 512 $v 1 32 256 3
  run $H 72 262144 256 $v 0 8 256 3
  run $H 1024 4096 256 $v 0 8 256 3
  run $H 1024 4096 512 $v 0 8 256 3
  run $H 16 1048576 512 $v 0 8 256 3
done
run $H 72 262144 256 1 1 64 512 3
run $H 72 262144 512 1 1 64 512 3
cat $out
