#!/bin/bash
# rocprofv3 kernel table of the banded 200k product (fp32 N = 128; LDS ring of the no-barrier kernel): profiles/r2/banded_r2b_kernel_stats.csv
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
out=$ROOT/gpurun_out/r2/banded_prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $ROOT/scripts/lab/r2_banded.py prof > $out/run.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]: print("%-70s calls %6s avg_us %9.2f" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
tail -1 $out/run.log | cut -c1-200
