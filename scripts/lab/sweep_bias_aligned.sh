for b in 0 0.1 0.15 0.2; do
  for cfg in "" "--row-block 64 --col-block 64" "--col-block 64" "--row-block 64"; do
    SPARTA_STREAM_ALIGN=1 SPARTA_SLOT_BIAS=$b python bench.py --no-cpu-baseline $cfg 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bias $b [$cfg]', d['value'], d['ms_per_step'], r['achieved'], r['frac'])"
  done
done
