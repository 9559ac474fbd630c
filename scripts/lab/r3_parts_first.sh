#!/bin/bash
# first GPU run of the rmat-part workload: box facts, small-scale self-checks (one GPU streamed; 2 ranks over gloo sharing the GPU), one full-size part
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/r3
(nproc; free -g; rocm-smi --showmeminfo vram | head -8) > gpurun_out/r3/box.txt 2>&1
python bench.py --workload rmat-part --rmat-scale 16 --rmat-density 1e-3 --slabs 4 --steps 5 --warmup 2 --ncols 256 > gpurun_out/r3/parts_small.json 2> gpurun_out/r3/parts_small.err || { tail -20 gpurun_out/r3/parts_small.err; exit 1; }
echo small ok
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --rmat-scale 16 --rmat-density 1e-3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3/parts_gloo2.json 2> gpurun_out/r3/parts_gloo2.err || { tail -30 gpurun_out/r3/parts_gloo2.err; exit 1; }
echo gloo2 ok
timeout -k 10 600 python bench.py --workload rmat-part --slabs 8 --slab-sample 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3/parts_one_full.json 2> gpurun_out/r3/parts_one_full.err || { tail -30 gpurun_out/r3/parts_one_full.err; exit 1; }
echo full-part ok
tail -c 3000 gpurun_out/r3/parts_one_full.json
