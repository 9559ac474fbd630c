#!/bin/bash
# lab: FETCH_SIZE of the hub kernel on a hub part under the environment given as NAME=VALUE arguments
cfg=${1:-5}; shift
for v in "$@"; do export $v; done
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/r4/hub_fetch_part
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 scripts/lab/r4_hub_parts.py $cfg,only > $out/p1.log 2>&1
python3 - $out <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hub_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in agg.items(): print(c, len(v), sum(v) / len(v))
print("fetch GB per launch (x2 corrected): %.1f" % (2 * sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"]) * 1024 / 1e9))
PY
grep "hub G=4" $out/p1.log | cut -c1-120
rm -rf $out
