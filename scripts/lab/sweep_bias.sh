for b in 0 0.08 0.15 0.22 0.3; do
  export SPARTA_SLOT_BIAS=$b
  echo "== bias $b"
  SPARTA_PATH=stream python scripts/gpu_long.py 2>&1 | grep TF | head -1
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('flagship', d['value'], d['ms_per_step'], r['achieved'], r['frac'])"
  python bench.py --no-cpu-baseline --row-block 64 --col-block 64 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('64x64   ', d['value'], d['ms_per_step'], r['achieved'], r['frac'])"
done
