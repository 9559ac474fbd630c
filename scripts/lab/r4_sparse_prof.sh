#!/bin/bash
# lab: per-kernel times of the sparse-row leg on a part (rocprofv3 --kernel-trace --stats)
cfg=${1:-c4}; part=${2:-5}; shift; shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
export HUB_PART=$part
for v in "$@"; do export $v; done
out=gpurun_out/r4/sp_prof
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/lab/r4_hub_parts.py $cfg,only > $out.log 2>&1
python3 - $out <<'PY'
import sys, glob, csv
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print("%-90s calls %5s avg %10.1f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $out
