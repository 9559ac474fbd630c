#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_spmm_gpu.py tests/test_real_matrices.py -m gpu -x -q -k "16bit or real" > gpurun_out/r3/pair_tests.log 2>&1; tail -3 gpurun_out/r3/pair_tests.log
BENCH_ARGS="--dtype f16 --col-block 64 --row-block 32" bash scripts/lab/r2_h16_ahead_ab.sh pair_w64:X=1 nopair_w64:SPARTA_H16_PAIR=0 pair_w64b:X=1 nopair_w64b:SPARTA_H16_PAIR=0
