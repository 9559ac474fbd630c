#!/bin/bash
# R-MAT 2^18 / 2^20, fp32 N = 128 (the suite's shape): chunk width x rows per workgroup of the column-major sparse-row kernel
mkdir -p gpurun_out/r2
for sc in 18 20; do for v in 4 2 1; do for cm in 16 32; do
  SPARTA_SP_VEC=$v SPARTA_SP_CMROWS=$cm python bench.py --workload rmat --rmat-scale $sc --ncols 128 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2/n128.json 2> gpurun_out/r2/n128.err
  python - $sc $v $cm <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r2/n128.json').read().strip().splitlines()[-1])
    print("scale %s vec %s cm %s: %.4f ms  %s" % (*sys.argv[1:4], d['ms_per_step'], d['roofline'].get('kernels_ms')))
except Exception as e: print(sys.argv[1:], 'ERR', e)
PY
done; done; done
