#!/bin/bash
# lab (round 5): per-kernel table (rocprofv3 --kernel-trace --stats) of the `clustered` family through the column-compacted tiles: scripts/lab/r5_union.py <N ...>
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
tag=${TAG:-union}; out=gpurun_out/r5/kt_$tag; rm -rf $out; mkdir -p gpurun_out/r5
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/lab/r5_union.py "$@" > gpurun_out/r5/kt_$tag.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { head -1 "$f"; grep "sparse_\|vbs_\|b_to_row_major\|colres" "$f"; } > gpurun_out/r5/kt_${tag}_kernel_stats.csv
rm -rf $out; cut -c1-220 gpurun_out/r5/kt_${tag}_kernel_stats.csv; grep "^{" gpurun_out/r5/kt_$tag.log | cut -c1-600
