#!/bin/bash
# kernel table of the R-MAT 2^20 bf16 N=512 product for one env variant: scripts/r2_sparse_prof.sh name ENV=1 ...
name=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
out=$ROOT/gpurun_out/r2/prof_$name; mkdir -p $out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $ROOT/bench.py --workload rmat --rmat-scale 20 --dtype bf16 --ncols 512 --steps 10 --warmup 2 --no-cpu-baseline > $out/bench.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]: print("%-60s calls %5s avg_us %10.1f" % (r['Name'].replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
