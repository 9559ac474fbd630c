#!/bin/bash
# the round's final record on one box (gpurun): the driver's command (with the suite in its line), the default command, the 16-bit line, the suite sweep, the two-rank gloo line,
# with the rocprofv3 profiles around them (fp32 + f16 flagships first: scripts/profile_bench.sh; the column-compacted tile kernel last: scripts/r5_union_profile.sh).  GPU tests: SKIP_TESTS= to include them.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/r5
if [ -n "${WITH_TESTS:-}" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest.log 2>&1 || { tail -40 gpurun_out/r5/gputest.log; exit 1; }
  tail -2 gpurun_out/r5/gputest.log
fi
# the rocprofv3 passes FIRST: their HBM-traffic figure (profiles/traffic*.json, tagged with the kernel revision) is what the bench lines below quote as roofline.traffic
if [ -z "${SKIP_PROFILES:-}" ]; then
  bash scripts/profile_bench.sh r5_default > gpurun_out/prof_r5_default.log 2>&1; tail -2 gpurun_out/prof_r5_default.log | cut -c1-300
  bash scripts/profile_bench.sh r5_f16 --dtype f16 > gpurun_out/prof_r5_f16.log 2>&1; tail -2 gpurun_out/prof_r5_f16.log | cut -c1-300
  [ -f gpurun_out/prof_r5_default/traffic.json ] && cp gpurun_out/prof_r5_default/traffic.json profiles/traffic.json
  [ -f gpurun_out/prof_r5_f16/traffic.json ] && cp gpurun_out/prof_r5_f16/traffic.json profiles/traffic_f16.json
fi
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_driver_cmd.json 2> gpurun_out/r5/bench_driver_cmd.err || { tail -20 gpurun_out/r5/bench_driver_cmd.err; exit 1; }
timeout -k 10 600 python bench.py --no-suite > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err || { tail -20 gpurun_out/r5/bench_default.err; exit 1; }
timeout -k 10 600 python bench.py --no-suite --dtype f16 --no-cpu-baseline > gpurun_out/r5/bench_f16.json 2> gpurun_out/r5/bench_f16.err || { tail -20 gpurun_out/r5/bench_f16.err; exit 1; }
timeout -k 10 600 python scripts/suite_sweep.py gpurun_out/r5/suite.json > gpurun_out/r5/suite.md 2> gpurun_out/r5/suite.err || { tail -20 gpurun_out/r5/suite.err; exit 1; }
# configs[4]'s code path end to end with two ranks on this one GPU over gloo (the only N > 1 bench.py record a one-GPU box can give)
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --rmat-scale 16 --rmat-density 1e-3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5/parts_gloo2.json 2> gpurun_out/r5/parts_gloo2.err || { tail -30 gpurun_out/r5/parts_gloo2.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_driver_cmd", "bench_default", "bench_f16", "parts_gloo2"):
    j = json.loads(open("gpurun_out/r5/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "mixed", j["roofline"].get("mixed_roofline_frac"), "cpu", (j.get("cpu_baseline") or {}).get("value"), "n_gpus", j.get("n_gpus"))
PY
tail -30 gpurun_out/r5/suite.md | cut -c1-220
if [ -z "${SKIP_PROFILES:-}" ]; then
  bash scripts/r5_union_profile.sh 128 > gpurun_out/prof_r5_union.log 2>&1; tail -12 gpurun_out/prof_r5_union.log | cut -c1-200
fi
