#!/bin/bash
# HBM fetch of the sparse-row kernels on one part of configs[4]: usage r3_sparse_pmc.sh [part]
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/r3/sparse_pmc; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc -- python3 scripts/lab/r3_sparse_pmc.py ${1:-3} > $out/run.log 2>&1
tail -2 $out/run.log
python3 - $out <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "sparse" in k or "vbs_" in k or "row_major" in k:
        print(k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()}, "fetch x2 GB: %.3f" % (2 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024 / 1e9))
PY
