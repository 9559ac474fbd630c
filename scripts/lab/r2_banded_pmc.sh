#!/bin/bash
# HBM traffic of the banded 200k product (short tiles): variable-height tiles vs fixed 32-row tiles, counters only (FETCH_SIZE and WRITE_SIZE in separate passes)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
out=$ROOT/gpurun_out/r2/banded_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in var fixed; do
  scr=$ROOT/scripts/lab/r2_banded.py; [ $v = fixed ] && scr=$ROOT/scripts/lab/r2_banded_fixed.py
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$v/fetch -- python3 $scr $v > $out/$v.fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $out/$v/write -- python3 $scr $v > $out/$v.write.log 2>&1
  python3 - $out/$v <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vbs_spmm_f32_direct" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1].split("/")[-1], {k: round(sum(v)/len(v),1) for k,v in agg.items()}, "launches", len(agg.get("FETCH_SIZE",[])))
PY
done
