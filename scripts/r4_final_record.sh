#!/bin/bash
# the round's final record on one box: GPU tests, the driver's command (with the suite in its line), the default command, the suite sweep, rocprofv3 profiles
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/r4
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputest.log 2>&1 || { tail -40 gpurun_out/r4/gputest.log; exit 1; }
  tail -2 gpurun_out/r4/gputest.log
fi
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_driver_cmd.json 2> gpurun_out/r4/bench_driver_cmd.err || { tail -20 gpurun_out/r4/bench_driver_cmd.err; exit 1; }
timeout -k 10 600 python bench.py --no-suite > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err || { tail -20 gpurun_out/r4/bench_default.err; exit 1; }
timeout -k 10 600 python bench.py --no-suite --dtype f16 --no-cpu-baseline > gpurun_out/r4/bench_f16.json 2> gpurun_out/r4/bench_f16.err || { tail -20 gpurun_out/r4/bench_f16.err; exit 1; }
python scripts/suite_sweep.py gpurun_out/r4/suite.json > gpurun_out/r4/suite.md 2> gpurun_out/r4/suite.err || { tail -20 gpurun_out/r4/suite.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_driver_cmd", "bench_default", "bench_f16"):
    j = json.loads(open("gpurun_out/r4/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "mixed", j["roofline"].get("mixed_roofline_frac"), "cpu", (j.get("cpu_baseline") or {}).get("value"))
PY
tail -14 gpurun_out/r4/suite.md | cut -c1-200
[ -n "${SKIP_PROFILES:-}" ] || bash scripts/r4_profiles.sh
