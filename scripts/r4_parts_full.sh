#!/bin/bash
# one-GPU full-size runs of the rmat-part workload (round 4: hub plan, padded B)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/r4
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 1000 python bench.py "$@" > gpurun_out/r4/$name.json 2> gpurun_out/r4/$name.err || { tail -30 gpurun_out/r4/$name.err; return 1; }; tail -c 600 gpurun_out/r4/$name.json; echo; }
for job in "$@"; do
  case $job in
    c4) run slabs8_scale23 --workload rmat-part --slabs 8 --steps 10 --warmup 3 || exit 1 ;;
    c3_01) run c3_0p1pct_off --workload rmat-part --rmat-scale 20 --rmat-density 1e-3 --slabs 8 --steps 10 --warmup 3 || exit 1 ;;
    c3_1) run c3_1pct_off --workload rmat-part --rmat-scale 20 --rmat-density 1e-2 --slabs 16 --steps 5 --warmup 2 --no-cpu-baseline || exit 1 ;;
    c3_01_on) SPARTA_MINHASH_VERBOSE=1 run c3_0p1pct_on --workload rmat-part --rmat-scale 20 --rmat-density 1e-3 --slabs 8 --steps 10 --warmup 3 --reorder on --no-cpu-baseline || exit 1; grep minhash gpurun_out/r4/c3_0p1pct_on.err | tail -12 ;;
    c3_01_auto) run c3_0p1pct_auto --workload rmat-part --rmat-scale 20 --rmat-density 1e-3 --slabs 8 --steps 10 --warmup 3 --reorder auto --no-cpu-baseline || exit 1 ;;
    c3_5) run c3_5pct_off --workload rmat-part --rmat-scale 20 --rmat-density 5e-2 --slabs 64 --slab-sample 6 --steps 5 --warmup 2 --no-cpu-baseline || exit 1 ;;
  esac
done
