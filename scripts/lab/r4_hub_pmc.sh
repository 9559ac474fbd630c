#!/bin/bash
# lab: PMC counters of the hub kernel on the ubench (one pass per counter group; counters only)
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/r4/hub_pmc
mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
H=scripts/ubench/hub_gemm
# built here from the source beside it (no binaries in the tree): the numbers belong to this hub_gemm.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I sparta_amd/csrc scripts/ubench/hub_gemm.hip -o $H || exit 1
for cfg in "0 36x2" "10 36"; do
  set -- $cfg; v=$1
  if [ $v = 0 ]; then T=72; else T=36; fi
  args="$T 262144 512 $v 1 32 256 2"
  tag=v$v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- $H $args > $out/${tag}_trace.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/${tag}_p1 -- $H $args > $out/${tag}_p1.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/${tag}_p2 -- $H $args > $out/${tag}_p2.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_p3 -- $H $args > $out/${tag}_p3.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/${tag}_p4 -- $H $args > $out/${tag}_p4.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE TCP_TCC_READ_REQ_sum --output-format csv -d $out/${tag}_p5 -- $H $args > $out/${tag}_p5.log 2>&1
done
python3 - $out <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for tag in ("v0", "v10"):
    agg = collections.defaultdict(list); dur = []
    for f in glob.glob(out + "/%s_p*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "hub_kernel" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(tag, "dispatch mean us", sum(dur) / max(len(dur), 1) / 1e3)
    for c, v in sorted(agg.items()): print("   %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
    for f in glob.glob(out + "/%s_trace/**/*kernel_stats.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "hub_kernel" in r["Name"]: print("   trace:", r["Calls"], "calls avg ns", r["AverageNs"])
PY
grep -h "TFLOP" $out/*_trace.log | cut -c1-200
rm -rf $out/*_trace/*/*.db 2>/dev/null; du -sh $out
