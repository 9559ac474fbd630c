#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/profile_bench.sh <tag> [extra bench.py arguments, e.g. --dtype f16]
# 1) rocprofv3 --kernel-trace --stats of the default bench command, 2) separate --pmc passes (counters only) for HBM traffic.
set -u
tag=$1
shift
extra="$*"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $extra --no-cpu-baseline --no-suite > $out/bench_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch -- python3 bench.py $extra --steps 20 --warmup 5 --no-cpu-baseline --no-suite > $out/bench_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $out/pmc_write -- python3 bench.py $extra --steps 20 --warmup 5 --no-cpu-baseline --no-suite > $out/bench_pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/pmc_sq -- python3 bench.py $extra --steps 20 --warmup 5 --no-cpu-baseline --no-suite > $out/bench_pmc_sq.log 2>&1
python3 - "$out" <<'PY'
import sys, glob, csv, collections, json, re
out = sys.argv[1]
res = {"kernel_stats": [], "pmc": {}}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vbs_" in r["Name"] or "sparse_" in r["Name"] or "b_to_row_major" in r["Name"]:
            res["kernel_stats"].append({"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])})
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vbs_" not in r["Kernel_Name"]: continue
        k = re.search(r"vbs_\w+(<[^>]*>)?", r["Kernel_Name"]).group(0)
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    res["pmc"][k] = {c: sum(v) / len(v) for c, v in d.items()}
    p = res["pmc"][k]
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        # MI355X_MICROARCH.md section HBM: FETCH_SIZE (KB) reports exactly 1/2 of a wide coalesced read on gfx950 -> x2; WRITE_SIZE (KB) exact
        p["hbm_bytes_per_launch_corrected"] = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
# traffic.json for bench.py: HBM bytes per launch of the kernel with the largest total time in the trace, tied to the workload
try:
    line = [l for l in open(out + "/bench_trace.log") if l.startswith("{")][-1]
    bj = json.loads(line)
    dom = max(res["kernel_stats"], key=lambda k: k["calls"] * k["avg_ns"])
    key = re.search(r"vbs_\w+(<[^>]*>)?", dom["name"]).group(0)
    p = res["pmc"][key]
    json.dump({
        "note": "HBM bytes per launch of the dominant kernel of the default bench.py workload, from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, counters only). Correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE (KB) reports 1/2 of a wide coalesced read on gfx950 -> doubled; WRITE_SIZE (KB) exact.",
        "command": "scripts/profile_bench.sh <tag>  (rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python3 bench.py ...; rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS -- python3 bench.py ...)",
        "workload": {"vbs_area": bj["config"]["vbs_area"], "n_cols": bj["config"]["n_cols"], "kernel_path": bj["config"]["kernel_path"], "dtype": bj["dtype"],
                     "kernel_rev": bj["config"].get("kernel_rev", "")},
        "kernel": key, "kernel_avg_ns_in_trace": dom["avg_ns"],
        "FETCH_SIZE_KB": p["FETCH_SIZE"], "WRITE_SIZE_KB": p["WRITE_SIZE"], "hbm_bytes_per_launch": p["hbm_bytes_per_launch_corrected"],
    }, open(out + "/traffic.json", "w"), indent=1)
except Exception as e:
    print("traffic.json not written:", e)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
tail -1 $out/bench_trace.log | cut -c1-300
