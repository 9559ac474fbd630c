#!/bin/bash
# lab (round 5): variants of vbs_union_f32_kernel built beside the product library (libsparta_amd_<name>.so = k_union.hip compiled with -DSPARTA_UNION_TAILPIPE=1 (tp),
# -DSPARTA_UNION_STATS=1 (st), both (tpst), the other objects as in the product library; selected with SPARTA_AMD_LIB) on 2000 true clusters x 48 rows: full product / without the tails
for lib in libsparta_amd.so libsparta_amd_tp.so; do
  [ -f sparta_amd/$lib ] || continue
  echo "== $lib"
  SPARTA_AMD_LIB=$PWD/sparta_amd/$lib COLS=${COLS:-60000} PROBES=0,8 python scripts/lab/r5_union_l2.py 2>&1 | grep columns | cut -c1-200
done
for lib in libsparta_amd_st.so libsparta_amd_tpst.so; do
  [ -f sparta_amd/$lib ] || continue
  echo "== $lib"
  SPARTA_AMD_LIB=$PWD/sparta_amd/$lib COLS=${COLS:-60000} PROBES=0,8 python scripts/lab/r5_union_stats.py 2>&1 | grep -v Warn
done
