#!/bin/bash
# the sparse-row kernels' cache counters on one part of configs[4] (argument: the part, default 5): FETCH_SIZE alone in its pass, then the L2 hit / miss counts
# (bench.py --workload rmat-part itself dies under --pmc inside torch's generator kernels; the lab script builds the same part with the same library calls)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/r4/sp_pmc
rm -rf $out; mkdir -p $out
export HUB_PART=${1:-5}
CMD="python3 scripts/lab/r4_hub_parts.py c4,only"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- $CMD > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/p2 -- $CMD > $out/p2.log 2>&1 || { tail -5 $out/p2.log; exit 1; }
python3 - $out <<'PY'
import sys, glob, csv, collections, json, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(sparse_\w+|vbs_\w+|b_to_row_major_kernel)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"command": "scripts/r4_sparse_pmc.sh (rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE | --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- python3 scripts/lab/r4_hub_parts.py c4,only with HUB_PART=the part)",
       "note": "mean per launch on that part; FETCH_SIZE in KB as reported -- x 2 on gfx950 for bytes (MI355X_MICROARCH.md, HBM section)", "kernels": {}}
for k, d in agg.items():
    res["kernels"][k] = {c: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in d.items()}
    if "FETCH_SIZE" in d: res["kernels"][k]["fetch_gb_per_launch_corrected"] = 2.0 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024 / 1e9
    if "TCC_HIT_sum" in d and "TCC_REQ_sum" in d: res["kernels"][k]["l2_hit_rate"] = sum(d["TCC_HIT_sum"]) / max(sum(d["TCC_REQ_sum"]), 1)
import os
res["part"] = int(os.environ.get("HUB_PART", "5"))
json.dump(res, open(out + "_part%d.json" % res["part"], "w"), indent=1)
for k, v in res["kernels"].items(): print(k, {a: (round(b, 3) if not isinstance(b, dict) else round(b["mean"], 1)) for a, b in v.items()})
PY
rm -rf $out/p1 $out/p2
