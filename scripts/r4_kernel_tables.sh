#!/bin/bash
# per-kernel tables (rocprofv3 --kernel-trace --stats) of the power-law configs on one GPU: configs[3] at 0.1 % (8 parts) and configs[4] (8 parts)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out/r4
run() { name=$1; shift; rm -rf gpurun_out/r4/kt_$name
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/kt_$name -- python3 bench.py "$@" --no-cpu-baseline > gpurun_out/r4/kt_$name.log 2>&1
  f=$(find gpurun_out/r4/kt_$name -name "*kernel_stats.csv" | head -1)
  # the library's kernels only (the generator's torch kernels and copies are not the product), then drop the trace (tens of MB)
  [ -n "$f" ] && { head -1 "$f"; grep "sparse_\|vbs_\|b_to_row_major\|pack_blocks" "$f"; } > gpurun_out/r4/kt_${name}_kernel_stats.csv
  rm -rf gpurun_out/r4/kt_$name; cut -c1-200 gpurun_out/r4/kt_${name}_kernel_stats.csv; tail -c 400 gpurun_out/r4/kt_$name.log | head -c 400; echo; }
for job in "$@"; do
  case $job in
    c3_1) run c3_1pct --workload rmat-part --rmat-scale 20 --rmat-density 1e-2 --slabs 16 --steps 5 --warmup 2 ;;
    c3_01) run c3_0p1pct --workload rmat-part --rmat-scale 20 --rmat-density 1e-3 --slabs 8 --steps 10 --warmup 3 ;;
    c4) run c4 --workload rmat-part --slabs 8 --steps 10 --warmup 3 ;;
  esac
done
