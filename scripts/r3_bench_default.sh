#!/bin/bash
# the driver's command (+ the default command) with the benchmark set in the line, and the two new full-density GPU tests
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "stated_density or full_size_slab" > gpurun_out/r3/newtests2.log 2>&1 || { tail -30 gpurun_out/r3/newtests2.log; exit 1; }
tail -2 gpurun_out/r3/newtests2.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_driver_cmd.json 2> gpurun_out/r3/bench_driver_cmd.err || { tail -20 gpurun_out/r3/bench_driver_cmd.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r3/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "mixed", j["roofline"].get("mixed_roofline_frac"), "cpu", j["cpu_baseline"]["value"])
s=j["config"]["suite"]
print("suite seconds", s.get("seconds"), "min", s.get("min_frac_8d"), "median", s.get("median_frac_8d"))
for r in s.get("matrices", []): print("  ", r.get("name","")[:44].ljust(44), r.get("ms"), r.get("useful_gflops"), r.get("frac_8d"), r.get("gather_gbs"), r.get("carried_by"), r.get("error",""))
PY
