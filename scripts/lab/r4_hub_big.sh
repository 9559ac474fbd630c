#!/bin/bash
out=gpurun_out/r4/hub_big.txt
mkdir -p gpurun_out/r4; : > $out
run() { echo "+ $*" >> $out; timeout -k 10 200 "$@" 2>&1 | grep -v MISMATCH >> $out; if grep -q "Memory access fault" $out; then echo "GPU fault"; cat $out; exit 1; fi; }
H=scripts/ubench/hub_gemm
# the real hub shape: 36 group tiles of 256 rows (72 of 128), K = 1 M, N = 512: B is 1 GB (beyond the Infinity Cache)
for v in 10 8 9; do
  run $H 36 1048576 512 $v 1 7 256 3
  run $H 36 1048576 512 $v 1 64 256 3
  run $H 36 1048576 512 $v 0 8 256 3
done
run $H 72 1048576 512 0 1 7 256 3
run $H 72 1048576 512 0 1 64 256 3
cat $out
