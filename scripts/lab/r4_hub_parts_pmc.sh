#!/bin/bash
# lab: PMC counters of the hub kernel on part 0 of configs[3] at 5 % (one pass per counter group)
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/r4/hub_parts_pmc
mkdir -p $out
cfg=${1:-5}
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 scripts/lab/r4_hub_parts.py $cfg,only > $out/p1.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/p2 -- python3 scripts/lab/r4_hub_parts.py $cfg,only > $out/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/p3 -- python3 scripts/lab/r4_hub_parts.py $cfg,only > $out/p3.log 2>&1
python3 - $out <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(list); dur = []
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hub_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("hub kernel dispatch mean us", sum(dur) / max(len(dur), 1) / 1e3, "n", len(dur))
for c, v in sorted(agg.items()): print("   %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
grep -h "hub G=4" $out/p1.log | cut -c1-200
