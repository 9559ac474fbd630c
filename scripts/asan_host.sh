#!/bin/bash
# Host-side sanitizer run (CPU container; GPU AddressSanitizer is not available on the pool): the host C++ of libsparta_amd.so
# (reorder engine incl. algorithm 7, VBS / hybrid builders, I/O, C-ABI glue) is built with -fsanitize=address,undefined into a
# scratch library and the CPU test files that exercise it run against it through SPARTA_AMD_LIB.
#   scripts/asan_host.sh        -> prints every sanitizer report (none expected) and the pytest summary
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/sparta_asan
mkdir -p "$OUT"
for f in capi reorder vbs_build io; do
  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I"$ROOT/include" -I"$ROOT/sparta_amd/csrc" -c "$ROOT/sparta_amd/csrc/$f.cpp" -o "$OUT/$f.o"
done
/opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -I"$ROOT/include" -I"$ROOT/sparta_amd/csrc" --offload-arch=gfx950 -c "$ROOT/sparta_amd/csrc/vbs_spmm.hip" -o "$OUT/vbs_spmm.o"
g++ -shared -fPIC -fsanitize=address,undefined -o "$OUT/libsparta_amd_asan.so" "$OUT"/{capi,reorder,vbs_build,io,vbs_spmm}.o -L/opt/rocm/lib -lamdhip64 -lpthread
cd "$ROOT"
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 SPARTA_AMD_LIB="$OUT/libsparta_amd_asan.so" \
  python -m pytest tests/test_host_golden.py tests/test_capi.py tests/test_io.py tests/test_oracle_vs_ref.py -q -s -p no:cacheprovider 2>&1 \
  | grep -E "runtime error|AddressSanitizer|SUMMARY|passed|failed" || true
