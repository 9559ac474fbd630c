#!/bin/bash
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
out=$ROOT/gpurun_out/r3/small_prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in ${@:-ca-HepPh bcsstk18}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$m -- python3 $ROOT/scripts/lab/r3_small_real.py $m > $out/$m.log 2>&1
  f=$(find $out/$m -name "*kernel_stats.csv" | head -1)
  echo "== $m"; tail -1 $out/$m.log; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-90s calls %6s avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $out/$m; done
