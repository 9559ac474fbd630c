#!/bin/bash
# round-2 functional pass over the bench workloads at small sizes (one GPU box): every line must be valid JSON
set -o pipefail
mkdir -p gpurun_out/r2
run() { name=$1; shift; echo "== $name: $*"; "$@" > gpurun_out/r2/$name.json 2> gpurun_out/r2/$name.err; rc=$?; echo "rc=$rc"; tail -c 400 gpurun_out/r2/$name.json; echo; [ $rc -ne 0 ] && tail -5 gpurun_out/r2/$name.err; return 0; }
run s_default python bench.py --steps 20 --warmup 5
run s_rmat16 python bench.py --workload rmat --rmat-scale 16 --steps 20 --warmup 5
run s_rmat18_on python bench.py --workload rmat --rmat-scale 18 --rmat-density 0.0002 --dtype bf16 --ncols 512 --steps 10 --warmup 3
run s_rmat18_off python bench.py --workload rmat --rmat-scale 18 --rmat-density 0.0002 --dtype bf16 --ncols 512 --steps 10 --warmup 3 --fixed-height 64
run s_infeasible python bench.py --workload rmat --rmat-scale 20 --rmat-density 0.05 --dtype bf16 --ncols 512
run s_dist2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --workload rmat --rmat-scale 16 --steps 5 --warmup 2
run s_dist1 python bench.py --workload rmat --rmat-scale 16 --steps 5 --warmup 2 --dist-path
run s_fem2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2
