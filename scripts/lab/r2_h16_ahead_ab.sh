#!/bin/bash
# A/B of the 16-bit no-barrier kernel on the flagship: usage r2_h16_ahead_ab.sh name:ENV=val,ENV=val ...   (bench --dtype f16, one line per variant)
mkdir -p gpurun_out/r2; ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
args=${BENCH_ARGS:---dtype f16}
for spec in "$@"; do
  name=${spec%%:*}; envs=$(echo ${spec#*:} | tr ',' ' ')
  env $envs python bench.py $args --steps 500 --warmup 50 --no-cpu-baseline --no-suite > gpurun_out/r2/h16ab_$name.json 2> gpurun_out/r2/h16ab_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/r2/h16ab_%s.json'%n).read().strip().splitlines()[-1]); r=d['roofline']
    print("%-24s ms/step %.5f kernel_ms %.5f frac %.4f parity %s" % (n, d['ms_per_step'], r['kernel_ms'], r['frac'], d.get('parity')))
except Exception as e: print(n, 'ERR', e)
PY
done
