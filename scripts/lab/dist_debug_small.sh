set -u
port=29700
for cfg in "2 auto f32" "3 blocks f16" "4 auto f32" "2 allgather bf16"; do
  set -- $cfg
  port=$((port + 1))
  echo "== N=$1 exchange=$2 dtype=$3"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus $1 --steps 30 --warmup 5 \
      --backend gloo --exchange $2 --dtype $3 --no-cpu-baseline 2> gpurun_out/dist_debug_$1_$2_$3.err | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
j = json.loads(l)
print(j['n_gpus'], j['dtype'], j['ms_per_step'], j['value'], '|', j['config']['parallelism'][:150])" || exit 1
  grep -i -E "FAILED|raised|Traceback|Error" gpurun_out/dist_debug_$1_$2_$3.err | head -5
done
