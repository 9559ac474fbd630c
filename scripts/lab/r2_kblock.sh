#!/bin/bash
# the per-block threshold of sparta_vbs_create_from_csr (SPARTA_SPARSE_K_BLOCK) on the power-law configs, both arms
mkdir -p gpurun_out/r2
one() { name=$1; shift; python bench.py "$@" --no-cpu-baseline > gpurun_out/r2/kb.json 2> gpurun_out/r2/kb.err; python - "$name" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r2/kb.json').read().strip().splitlines()[-1]); r=d['roofline']; di=d['config'].get('device_image',{})
    print("%-34s %.3f ms %s tiles_area %s blocks %s sp_nnz %s" % (sys.argv[1], d['ms_per_step'], r.get('kernels_ms'), di.get('mfma_tile_area'), di.get('mfma_blocks'), di.get('sparse_nnz')))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
}
for kb in ${KBS:-1e30 240 120 60}; do
  export SPARTA_SPARSE_K_BLOCK=$kb
  one "kb=$kb rmat1e-4 on" --workload rmat --rmat-scale 20 --rmat-density 0.0001 --dtype bf16 --ncols 512 --steps 10 --warmup 2
  one "kb=$kb rmat1e-4 off" --workload rmat --rmat-scale 20 --rmat-density 0.0001 --dtype bf16 --ncols 512 --steps 10 --warmup 2 --fixed-height 64
  one "kb=$kb ogbn on" --workload ogbn-like --steps 10 --warmup 2
done
