#!/bin/bash
# lab (round 5): the fp32 flagship (bench.py default workload, no suite, no CPU baseline) on compile-time variants of vbs_spmm_f32_direct_kernel, interleaved on ONE box:
#   base | SPARTA_DIRECT_TWOACC=1 (two accumulator chains) | SPARTA_DIRECT_ORDER=1 | SPARTA_DIRECT_FBPRE=1 | SPARTA_DIRECT_WAVES=3
# the variant libraries are built beside the product library by the caller (sparta_amd/libsparta_amd_<name>.so), loaded through SPARTA_AMD_LIB
for rep in 1 2; do
  for v in base twoacc order1 fbpre w3; do
    lib=sparta_amd/libsparta_amd.so; [ $v != base ] && lib=sparta_amd/libsparta_amd_$v.so
    [ -f $lib ] || continue
    SPARTA_AMD_LIB=$PWD/$lib python bench.py --no-suite --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v rep $rep: ms_per_step %.5f  value %.0f GFLOP/s  roofline.frac %.4f  kernel us %s' % (j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline'].get('kernel_us')))"
  done
done
