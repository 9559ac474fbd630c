#!/bin/bash
# GPU parity suite + one full-size part of configs[4] with the host-builder trace
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/gputest.log 2>&1 || { tail -40 gpurun_out/r3/gputest.log; exit 1; }
tail -3 gpurun_out/r3/gputest.log
SPARTA_BUILD_TRACE=1 timeout -k 10 600 python bench.py --workload rmat-part --slabs 8 --slab-sample 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3/parts_one_full.json 2> gpurun_out/r3/parts_one_full.err || { tail -30 gpurun_out/r3/parts_one_full.err; exit 1; }
grep "sparta build" gpurun_out/r3/parts_one_full.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3/parts_one_full.json"))
print(j["config"]["parts_detail"][0])
PY
