#!/bin/bash
# A/B of developer builds of the fp32 no-barrier kernel: usage r3_f32_variants.sh name:lib[:ENV=val,...] ...   (lib = default or a file under sparta_amd/)
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
for spec in "$@"; do
  IFS=: read -r name lib envs <<< "$spec"
  if [ "$lib" = "default" ]; then unset SPARTA_AMD_LIB; else export SPARTA_AMD_LIB=$ROOT/sparta_amd/$lib; fi
  env $(echo $envs | tr ',' ' ') python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-suite 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'ms', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'], 'value', j['value'])"
done
