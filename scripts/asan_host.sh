#!/bin/bash
# Host-side sanitizer run (CPU container; GPU AddressSanitizer is not available on the pool): the host C++ of libsparta_amd.so
# (reorder engine incl. algorithm 7, VBS / hybrid builders, I/O, C-ABI glue) is built with -fsanitize=address,undefined into a
# scratch library and the CPU test files that exercise it run against it through SPARTA_AMD_LIB.  (tests/test_oracle_vs_ref.py is left out: it compares
# the oracle with the compiled REFERENCE, whose multiply reads C before writing it -- under the sanitizer's allocator that shows as NaN in the reference's output.)
#   scripts/asan_host.sh        -> prints every sanitizer report (none expected) and the pytest summary
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/sparta_asan
mkdir -p "$OUT"
CXX=/opt/rocm/lib/llvm/bin/clang++                              # (g++ 11 has no _Float16 in C++: vbs_plan.cpp / vbs_capi.cpp need clang)
for f in capi reorder vbs_build io vbs_plan vbs_union vbs_capi; do
  $CXX -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I"$ROOT/include" -I"$ROOT/sparta_amd/csrc" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
      -c "$ROOT/sparta_amd/csrc/$f.cpp" -o "$OUT/$f.o"
done
make -s -C "$ROOT/sparta_amd/csrc"                             # the kernel objects (k_*.o, not instrumented: device code)
$CXX -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "$OUT/libsparta_amd_asan.so" "$OUT"/{capi,reorder,vbs_build,io,vbs_plan,vbs_union,vbs_capi}.o "$ROOT"/sparta_amd/csrc/k_*.o \
    -L/opt/rocm/lib -lamdhip64 -lpthread
cd "$ROOT"
LD_PRELOAD=$($CXX -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 SPARTA_AMD_LIB="$OUT/libsparta_amd_asan.so" \
  python -m pytest tests/test_host_golden.py tests/test_capi.py tests/test_io.py tests/test_union_host.py -q -s -p no:cacheprovider 2>&1 \
  | grep -E "runtime error|AddressSanitizer|SUMMARY|passed|failed" || true
# the hybrid builder of sparta_vbs_create_from_csr (per-block tile / sparse-row split) runs on the host before the first device call: without a GPU the
# call ends in SPARTA_ERR_NO_DEVICE -- after the builder has done all its work under the sanitizers
LD_PRELOAD=$($CXX -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 SPARTA_AMD_LIB="$OUT/libsparta_amd_asan.so" \
  SPARTA_SPARSE_MIN_STEPS=0 SPARTA_PLAN_DEBUG=1 SPARTA_LAUNCH_NNZ=0 python - <<'PY' 2>&1 | grep -E "runtime error|AddressSanitizer|SUMMARY|builder" || true
# (SPARTA_PLAN_DEBUG: without a GPU create goes on through the tile lists and the stream plans -- pair tiles, the k-compacted fragment image, the 16-bit slices --
#  before it fails with NO_DEVICE)
import numpy as np, sparta_amd as sa
for kb in ("1", "8", "60", "1e30"):
    import os
    os.environ["SPARTA_SPARSE_K_BLOCK"] = kb
    for seed in range(3):
        m = sa.gen.rmat(12, 40000, seed=seed, symmetrize=True, pattern_only=False)
        for g, rbs, ff in ((sa.BlockingEngine(tau=0.4, col_block_size=32, blocking_algo=7).GetGrouping(m), 0, False), (np.arange(m.rows) // 64, 64, True)):
            for dt in (sa.F32, sa.F16, sa.BF16):
                try:
                    sa.DeviceVBS.from_csr(m, g, 32, rbs, ff, device=0, dtype=dt)
                except sa.SpartaError as e:
                    pass
# well-filled matrices: every block a tile (plans of 32-row and 64-row tiles, pair tiles of the 16-bit handles, split and whole-tile cuts)
for w, hgt in ((32, 32), (32, 20), (64, 64), (64, 32)):
    m = sa.gen.fem3d(5, 5, 12, 3, seed=3)
    g = np.arange(m.rows) // hgt
    for dt in (sa.F32, sa.F16):
        for al in ("0", "1"):
            os.environ["SPARTA_STREAM_ALIGN"] = al
            try:
                sa.VBR().fill_from_CSR_inplace(m, g, w).to_device(0, dtype=dt)
            except sa.SpartaError:
                pass
os.environ.pop("SPARTA_STREAM_ALIGN", None)
print("hybrid builder exercised")
PY

