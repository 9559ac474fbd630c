#!/bin/bash
# round-2 measurement record: rocprofv3 of the default bench + one bench line per BASELINE config / arm that runs in about a minute
# (the 0.1 % density arms of configs[3] take ~4 minutes each: scripts/r2_configs_big.sh rmat_01 on|off)
set -u
mkdir -p gpurun_out/r2
bash scripts/profile_bench.sh r2_default > gpurun_out/r2/profile_default.log 2>&1; tail -2 gpurun_out/r2/profile_default.log | cut -c1-200
run() { name=$1; shift; echo "== $name ($(date +%T))"; timeout -k 10 600 "$@" > gpurun_out/r2/$name.json 2> gpurun_out/r2/$name.err; echo "rc=$?"; tail -c 200 gpurun_out/r2/$name.json; echo; }
run c1_cant_driver python bench.py --steps 20 --warmup 5
run c1_cant_default python bench.py
run c1_cant_f16 python bench.py --dtype f16
run c2_ogbn_on python bench.py --workload ogbn-like --steps 20 --warmup 3
run c2_ogbn_off python bench.py --workload ogbn-like --steps 20 --warmup 3 --fixed-height 64
run c3_rmat20_d0001_on python bench.py --workload rmat --rmat-scale 20 --rmat-density 0.0001 --dtype bf16 --ncols 512 --steps 10 --warmup 2
run c3_rmat20_d0001_off python bench.py --workload rmat --rmat-scale 20 --rmat-density 0.0001 --dtype bf16 --ncols 512 --steps 10 --warmup 2 --fixed-height 64
run c3_rmat20_d001_infeasible_1pct python bench.py --workload rmat --rmat-scale 20 --rmat-density 0.01 --dtype bf16 --ncols 512
run c3_rmat20_d005_infeasible python bench.py --workload rmat --rmat-scale 20 --rmat-density 0.05 --dtype bf16 --ncols 512
run c4_rmat_r1mini_f32 python bench.py --workload rmat --rmat-scale 20 --ncols 256 --steps 50 --warmup 5
run c5_dist2_gloo_one_gpu python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --backend gloo --workload rmat --rmat-scale 18 --ncols 256 --dtype f16 --steps 5 --warmup 2
python scripts/suite_sweep.py 128 gpurun_out/r2/suite.json > gpurun_out/r2/suite.md 2> gpurun_out/r2/suite.err; tail -8 gpurun_out/r2/suite.md | cut -c1-220
