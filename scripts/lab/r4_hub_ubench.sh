#!/bin/bash
# lab: the hub kernel's variants and schedules on a dense hub (scripts/ubench/hub_gemm.hip); run on the GPU box from the repo root
set -o pipefail
out=gpurun_out/r4/hub_ubench.txt
mkdir -p gpurun_out/r4
: > $out
H=scripts/ubench/hub_gemm
run() { echo "+ $*" >> $out; timeout -k 10 120 $H "$@" >> $out 2>&1 || echo "  (exit $?)" >> $out; if grep -q "Memory access fault" $out; then echo "GPU fault: stopping"; tail -5 $out; exit 1; fi; }
# correctness first: small shapes, every variant, both schedules, split and whole tiles
for v in 0 1 2 3; do
  run 4 4096 256 $v 0 8 8 2
  run 3 8192 512 $v 1 4 16 2
  run 64 2048 256 $v 0 8 64 2
done
tail -n 12 $out
# performance: T tiles of 128 rows, K columns
for v in 0 1 2 3; do
  run 72 262144 512 $v 0 8 256 3
  run 72 262144 512 $v 1 32 256 3
  run 72 262144 256 $v 0 8 256 3
  run 72 262144 256 $v 1 32 256 3
  run 72 262144 256 $v 1 64 512 3
done
run 16 1048576 512 0 0 8 256 3
run 16 1048576 512 0 1 128 256 3
run 16 1048576 512 1 0 8 256 3
run 16 1048576 512 1 1 128 256 3
grep -c "ok" $out; grep "WRONG\|exit\|error" $out | head
