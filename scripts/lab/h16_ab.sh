set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
o=gpurun_out/h16_ab.txt; : > $o
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
for depth in 2 4; do
  for cfg in "--dtype f16" "--dtype f16 --col-block 64 --row-block 64" "--dtype bf16" "--dtype f16 --ncols 256" "--dtype f16 --workload rmat --rmat-scale 18 --ncols 256"; do
    echo "depth=$depth $cfg" >> $o
    SPARTA_H16_DEPTH=$depth timeout -k 10 200 python bench.py $cfg --steps 500 --warmup 50 --no-cpu-baseline 2>>$o | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline']['achieved'], j['roofline']['frac'], j['roofline'].get('kernels_ms'))" >> $o || exit 1
  done
done
cat $o
