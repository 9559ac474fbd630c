#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_spmm_gpu.py tests/test_real_matrices.py tests/test_configs_gpu.py -m gpu -x -q -k "16bit or real or config1 or config2" > gpurun_out/r3/pair_tests.log 2>&1; tail -4 gpurun_out/r3/pair_tests.log
for n in 128 256 512; do
BENCH_ARGS="--dtype f16 --ncols $n" bash scripts/lab/r2_h16_ahead_ab.sh pair_n$n:X=1 nopair_n$n:SPARTA_H16_PAIR=0
done
BENCH_ARGS="--dtype bf16 --ncols 128" bash scripts/lab/r2_h16_ahead_ab.sh pair_bf16:X=1 nopair_bf16:SPARTA_H16_PAIR=0
