#!/bin/bash
# which chunk width for which size of B: R-MAT scale x vec, fp32 N=256 and bf16 N=512 (ms per product)
mkdir -p gpurun_out/r2
for sc in 14 16 18; do for dt in "f32 256" "bf16 512"; do set -- $dt; for v in 4 2 1; do for cm in 32 16; do
  SPARTA_SP_VEC=$v SPARTA_SP_CMROWS=$cm python bench.py --workload rmat --rmat-scale $sc --dtype $1 --ncols $2 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2/vec.json 2> gpurun_out/r2/vec.err
  python - $sc $1 $2 $v $cm <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r2/vec.json').read().strip().splitlines()[-1])
    print("scale %s %s N=%s vec %s cm %s: %.4f ms" % (*sys.argv[1:6], d['ms_per_step']))
except Exception as e: print(sys.argv[1:], 'ERR', e)
PY
done; done; done; done
