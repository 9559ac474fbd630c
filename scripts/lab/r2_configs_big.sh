#!/bin/bash
# round-2: BASELINE configs[2] and configs[3] at their stated sizes on one MI355X (reorder on AND off); one JSON line per run
set -o pipefail
mkdir -p gpurun_out/r2
run() { name=$1; shift; echo "== $name: $* ($(date +%T))"; timeout -k 10 ${TMO:-900} "$@" > gpurun_out/r2/$name.json 2> gpurun_out/r2/$name.err; rc=$?; echo "rc=$rc ($(date +%T))"; tail -c 300 gpurun_out/r2/$name.json; echo; [ $rc -ne 0 ] && tail -5 gpurun_out/r2/$name.err; return $rc; }
case "$1" in
  ogbn)  run c2_ogbn_on python bench.py --workload ogbn-like --steps 20 --warmup 3 && run c2_ogbn_off python bench.py --workload ogbn-like --steps 20 --warmup 3 --fixed-height 64 ;;
  rmat_lo) run c3_rmat20_d0001_on python bench.py --workload rmat --rmat-scale 20 --rmat-density 0.0001 --dtype bf16 --ncols 512 --steps 10 --warmup 2 && run c3_rmat20_d0001_off python bench.py --workload rmat --rmat-scale 20 --rmat-density 0.0001 --dtype bf16 --ncols 512 --steps 10 --warmup 2 --fixed-height 64 ;;
  rmat_01) TMO=1100 run c3_rmat20_d001_$2 python bench.py --workload rmat --rmat-scale 20 --rmat-density 0.001 --dtype bf16 --ncols 512 --steps 3 --warmup 1 --settle-ms 0 $3 $4 ;;
esac
