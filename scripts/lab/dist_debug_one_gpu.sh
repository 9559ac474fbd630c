#!/bin/bash
# usage (one-GPU box): scripts/dist_debug_one_gpu.sh  -- the multi-rank bench flow with N ranks sharing ONE GPU over gloo (collectives staged
# through the host: bench.py --backend gloo).  Checks the logic end to end (plan exchange, pack, splits, self-check against the all-gather,
# JSON line); the timings mean nothing.
set -u
port=29600
for cfg in "2 auto f32" "4 auto f32" "3 blocks f16" "2 allgather f32" "8 auto f32"; do
  set -- $cfg
  port=$((port + 1))
  echo "== N=$1 exchange=$2 dtype=$3"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus $1 --steps 30 --warmup 5 \
      --backend gloo --exchange $2 --dtype $3 --no-cpu-baseline 2> gpurun_out/dist_debug_$1_$2_$3.err | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
j = json.loads(l)
print(j['n_gpus'], j['dtype'], j['ms_per_step'], j['value'], '|', j['config']['parallelism'][:150])"
  grep -i -E "FAILED|raised|Traceback|Error" gpurun_out/dist_debug_$1_$2_$3.err | head -5
done
