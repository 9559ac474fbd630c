#!/bin/bash
# lab: L2 fill traffic of the hub kernel on the ubench at the real shape, one slab against two, schedules
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/r4/hub_fetch
rm -rf $out; mkdir -p $out
H=scripts/ubench/hub_gemm
# built here from the source beside it (no binaries in the tree): the numbers belong to this hub_gemm.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I sparta_amd/csrc scripts/ubench/hub_gemm.hip -o $H || exit 1
export HUB_LDB=1048640
i=0
for args in "36 1048576 256 10 1 8 256 2" "36 1048576 512 10 1 8 256 2" "36 1048576 512 10 1 64 256 2" "36 1048576 512 10 0 8 256 2"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/p$i -- $H $args > $out/p$i.log 2>&1
  echo "== $args"; grep -h TFLOP $out/p$i.log | cut -c1-160
  python3 - $out/p$i <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hub_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   " + "  ".join("%s %.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg.items())))
PY
done
rm -rf $out/p*/
