#!/bin/bash
# rocprofv3 record of the column-compacted tile kernel (k_union.hip) on the benchmark set's clustered family: kernel trace + stats, then counter-only passes
# (HBM bytes, MFMA, LDS, L2), one pass per counter group.  usage (on the GPU box, via gpurun): scripts/r5_union_profile.sh [N]
set -u
n=${1:-128}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/prof_union
rm -rf $out; mkdir -p $out gpurun_out/r5
run="python3 scripts/lab/r5_union_run.py $n"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $run 500 > $out/run_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch -- $run 20 > $out/run_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write -- $run 20 > $out/run_pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq -- $run 20 > $out/run_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM --output-format csv -d $out/pmc_mfma -- $run 20 > $out/run_pmc_mfma.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $out/pmc_lds -- $run 20 > $out/run_pmc_lds.log 2>&1
python3 - "$out" "$n" <<'PY'
import sys, glob, csv, collections, json
out, n = sys.argv[1], int(sys.argv[2])
res = {"workload": "bench_suite clustered 2000 x 48 rows, 300 shared cols", "n_cols": n, "kernel_stats": [], "pmc_union_kernel": {}}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "union" in r["Name"] or "vbs_" in r["Name"] or "sparse_" in r["Name"] or "b_to_row" in r["Name"]:
            res["kernel_stats"].append({"name": r["Name"][:120], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])})
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "union" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
p = {c: sum(v) / len(v) for c, v in agg.items()}
if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE (KB) reports 1/2 of a wide coalesced read on gfx950 -> doubled; WRITE_SIZE (KB) exact
    p["hbm_bytes_per_launch_corrected"] = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
res["pmc_union_kernel"] = p
for tag in ("trace", "pmc_fetch"):
    try:
        res["line_" + tag] = json.loads([l for l in open(out + "/run_%s.log" % tag) if l.startswith("{")][-1])
    except Exception as e:
        res["line_" + tag] = repr(e)
k = [x for x in res["kernel_stats"] if "union" in x["name"]]
if k and isinstance(res.get("line_trace"), dict):
    t = k[0]["avg_ns"] * 1e-9
    L = res["line_trace"]
    res["roofline_union_kernel"] = {"bound": "mfma", "achieved_tflops_on_stored_elements": L["union_kernel_flops_on_stored_elements"] / t / 1e12,
                                    "executed_tflops": L["union_kernel_flops_executed"] / t / 1e12, "peak_tflops": 157.3,
                                    "frac": L["union_kernel_flops_on_stored_elements"] / t / 1e12 / 157.3,
                                    "algorithmic_gbs": L["union_kernel_algorithmic_bytes"] / t / 1e9, "traffic_bytes": p.get("hbm_bytes_per_launch_corrected"),
                                    "traffic_over_algorithmic": (p.get("hbm_bytes_per_launch_corrected") or 0) / L["union_kernel_algorithmic_bytes"]}
json.dump(res, open("gpurun_out/r5/union_profile_clustered_n%d.json" % n, "w"), indent=1)
print(json.dumps(res, indent=1)[:3500])
PY
for f in $out/trace/*/*kernel_stats.csv; do { head -1 $f; grep "union\|vbs_\|sparse_\|b_to_row" $f; } > gpurun_out/r5/union_profile_clustered_n${n}_kernel_stats.csv; done
