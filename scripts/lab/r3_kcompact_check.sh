#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py tests/test_spmm_gpu.py -m gpu -x -q -k "config1 or mfma or oracle or layouts or ragged or tail or ring or accumulate or plan" > gpurun_out/r3/kc_tests.log 2>&1; tail -4 gpurun_out/r3/kc_tests.log
for i in 1 2; do python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-suite 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'value', j['value'])"; done
