#!/bin/bash
# per-kernel times of one part of configs[4] (rocprofv3 --kernel-trace --stats over scripts/lab/r3_sparse_pmc.py): usage r3_part_trace.sh part
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/r3/part_trace_$1; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/lab/r3_sparse_pmc.py $1 > $out/run.log 2>&1
grep "^part" $out/run.log
python3 - $out <<'PY'
import sys, glob, csv
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("sparse", "vbs_", "row_major")):
            print("%-90s calls %4s avg %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
