run() { desc="$1"; shift; env "$@" python bench.py --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$desc', d['value'], d['ms_per_step'], r['path'], r['achieved'], r['frac'], r['kernels_ms'], r.get('shader_clock_mhz'))"; }
ARGS=""
run "default      " X=1
run "interleave   " SPARTA_STREAM_INTERLEAVE=1
run "3 per CU     " SPARTA_WORKERS_PER_CU=3
run "split plan   " SPARTA_STREAM_ALIGN=0
ARGS="--col-block 64"
run "w64 auto     " X=1
run "w64 class    " SPARTA_PATH=class
run "w64 stream   " SPARTA_PATH=stream
ARGS="--row-block 64 --col-block 64"
run "64x64 auto   " X=1
ARGS="--row-block 64 --col-block 32"
run "64x32 auto   " X=1
