#!/bin/bash
# (1) the N = 2 share of configs[4] on one GPU (3.5e9 nonzeros in one handle: 64-bit paths), (2) two ranks sharing the GPU over gloo with --ag-chunks 2,
# (3) configs[3] at 0.1 % with the clustering (reorder on) and with reorder auto
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/r3
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 1100 python bench.py "$@" > gpurun_out/r3/$name.json 2> gpurun_out/r3/$name.err || { tail -30 gpurun_out/r3/$name.err; return 1; }; tail -c 400 gpurun_out/r3/$name.json; echo; }
for job in "$@"; do
  case $job in
    half) run half_of_c4 --workload rmat-part --slabs 2 --slab-sample 1 --steps 3 --warmup 1 --no-cpu-baseline || exit 1 ;;
    gloo) run gloo2_chunks --gpus 2 --backend gloo --rmat-scale 16 --rmat-density 1e-3 --steps 3 --warmup 1 --no-cpu-baseline --ag-chunks 2 || exit 1 ;;
    on) SPARTA_MINHASH_VERBOSE=1 run c3_0p1pct_on --workload rmat-part --rmat-scale 20 --rmat-density 1e-3 --slabs 8 --steps 10 --warmup 3 --reorder on --no-cpu-baseline || exit 1; grep "minhash: seed\|minhash: bands" gpurun_out/r3/c3_0p1pct_on.err | tail -8 ;;
    auto) run c3_0p1pct_auto --workload rmat-part --rmat-scale 20 --rmat-density 1e-3 --slabs 8 --steps 10 --warmup 3 --reorder auto --no-cpu-baseline || exit 1 ;;
  esac
done
