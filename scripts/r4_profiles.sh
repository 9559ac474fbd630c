#!/bin/bash
# the round's profile record: rocprofv3 kernel stats + PMC passes of the default bench command (fp32) and of --dtype f16
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
bash scripts/profile_bench.sh r4_default > gpurun_out/prof_r4_default.log 2>&1; tail -2 gpurun_out/prof_r4_default.log | cut -c1-300
bash scripts/profile_bench.sh r4_f16 --dtype f16 > gpurun_out/prof_r4_f16.log 2>&1; tail -2 gpurun_out/prof_r4_f16.log | cut -c1-300
ls gpurun_out/prof_r4_default gpurun_out/prof_r4_f16
