#!/bin/bash
# A/B of the fp32 product kernels on the flagship (bench default workload), one line per variant
mkdir -p gpurun_out/r2
run() { name=$1; shift; env "$@" python bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/r2/ab_$name.json 2> gpurun_out/r2/ab_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/r2/ab_%s.json'%n).read().strip().splitlines()[-1]); r=d['roofline']
    print("%-28s ms/step %.5f kernel_ms %.5f frac %.4f mixed %.4f clock %s" % (n, d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('mixed_roofline_frac',0), r.get('shader_clock_mhz')))
except Exception as e: print(n, 'ERR', e)
PY
}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  run $name $(echo $envs | tr ',' ' ')
done
