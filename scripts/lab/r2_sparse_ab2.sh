#!/bin/bash
# like r2_sparse_ab.sh plus the ogbn-like config: name:ENV=..,ENV=..
mkdir -p gpurun_out/r2
run() { name=$1; wl=$2; shift 2; env "$@" python bench.py $WLARGS --no-cpu-baseline > gpurun_out/r2/sp_${name}_$wl.json 2> gpurun_out/r2/sp_${name}_$wl.err; python - "$name" "$wl" <<'PY'
import json,sys
n,wl=sys.argv[1:3]
try:
    d=json.loads(open('gpurun_out/r2/sp_%s_%s.json'%(n,wl)).read().strip().splitlines()[-1]); r=d['roofline']
    print("%-20s %-12s ms/step %.4f value %.1f gather %.4f err %.2e" % (n, wl, d['ms_per_step'], d['value'], r.get('gather_frac',0), d['config'].get('parity_spot_check',{}).get('max_err_over_sum_abs',-1)))
except Exception as e: print(n, wl, 'ERR', e)
PY
}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  WLARGS="--workload rmat --rmat-scale 20 --dtype bf16 --ncols 512 --steps 20 --warmup 3" run $name bf16n512 $(echo $envs | tr ',' ' ')
  WLARGS="--workload rmat --rmat-scale 20 --ncols 256 --steps 20 --warmup 3" run $name f32n256 $(echo $envs | tr ',' ' ')
  [ -n "$OGBN" ] && WLARGS="--workload ogbn-like --steps 10 --warmup 2" run $name ogbn $(echo $envs | tr ',' ' ')
done
true
