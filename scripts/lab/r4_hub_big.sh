#!/bin/bash
out=gpurun_out/r4/hub_big.txt
mkdir -p gpurun_out/r4; : > $out
run() { echo "+ $* (HUB_LDB=$HUB_LDB)" >> $out; timeout -k 10 200 "$@" 2>&1 | grep -v MISMATCH >> $out; if grep -q "Memory access fault" $out; then echo "GPU fault"; cat $out; exit 1; fi; }
H=scripts/ubench/hub_gemm
export HUB_LDB=1048640
# the real hub shape: 36 group tiles of 256 rows, K = 1 M, N = 512, columns of B not a power of two apart
for v in 10 8 9; do
  run $H 36 1048576 512 $v 1 7 256 3
  run $H 36 1048576 512 $v 1 64 256 3
  run $H 36 1048576 512 $v 0 8 256 3
  run $H 36 1048576 256 $v 1 7 256 3
done
cat $out
