#!/bin/bash
# lab (round 5): LDS stages x workgroups per CU of vbs_union_h16_kernel on the clustered family (bf16, N = 128 and 512); variant libraries built beside the product library
for v in base:3 s2w3:3 s2w4:4 s3w4:4 s4w3:3; do
  n=${v%%:*}; w=${v#*:}
  lib=sparta_amd/libsparta_amd.so; [ $n != base ] && lib=sparta_amd/libsparta_amd_$n.so
  [ -f $lib ] || continue
  echo "== $n (plan: $w workgroups per CU)"
  SPARTA_AMD_LIB=$PWD/$lib SPARTA_UNION_WPC=$w ONLY=bf16 python scripts/lab/r5_union16.py 128 512 2>&1 | grep '"union": "1"' | grep bf16 | cut -c1-160
done
