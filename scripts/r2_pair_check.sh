#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
python -m pytest tests/test_spmm_gpu.py -x -q -m gpu > gpurun_out/r2/gpu_tests_pair.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r2/gpu_tests_pair.log
python bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/r2/bench_pair.json 2> gpurun_out/r2/bench_pair.err; echo "pair rc=$?"
SPARTA_F32_PLAN=legacy python bench.py --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/r2/bench_legacy.json 2> gpurun_out/r2/bench_legacy.err; echo "legacy rc=$?"
python - <<'PY'
import json
for f in ['bench_pair','bench_legacy']:
    try:
        d=json.loads(open('gpurun_out/r2/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('mixed_roofline_frac'), r.get('shader_clock_mhz'), d['config']['tiles'])
    except Exception as e: print(f, 'ERR', e)
PY
