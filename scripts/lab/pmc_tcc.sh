#!/bin/bash
# usage: scripts/pmc_tcc.sh <tag> [bench args]: L2 hit/miss + fetch size of the default bench kernel (counters only)
set -u
tag=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/tcc_$tag
mkdir -p $out
timeout 90 rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $out/p1 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/p1.log 2>&1
python3 scripts/pmc_summary.py "$out"
