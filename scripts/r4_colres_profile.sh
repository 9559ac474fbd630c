#!/bin/bash
# rocprofv3 record of the resident-column kernel (k_colres.hip) on one of the reference's real matrices: kernel trace + stats, then counter-only passes (HBM bytes, LDS, L2)
# usage (on the GPU box, via gpurun): scripts/r4_colres_profile.sh [matrix file] [N]
set -u
mat=${1:-bcsstk18_r.el}
n=${2:-8192}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/prof_colres
mkdir -p $out gpurun_out/r4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/lab/r4_colres_run.py $mat $n 200 > $out/run_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fetch -- python3 scripts/lab/r4_colres_run.py $mat $n 20 > $out/run_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $out/pmc_write -- python3 scripts/lab/r4_colres_run.py $mat $n 20 > $out/run_pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/pmc_sq -- python3 scripts/lab/r4_colres_run.py $mat $n 20 > $out/run_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_lds -- python3 scripts/lab/r4_colres_run.py $mat $n 20 > $out/run_pmc_lds.log 2>&1
python3 - "$out" "$mat" "$n" <<'PY'
import sys, glob, csv, collections, json
out, mat, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
res = {"matrix": mat, "n_cols": n, "kernel_stats": [], "pmc": {}}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "colres" in r["Name"] or "vbs_" in r["Name"] or "sparse_" in r["Name"]:
            res["kernel_stats"].append({"name": r["Name"][:120], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])})
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "colres" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
p = {c: sum(v) / len(v) for c, v in agg.items()}
if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE (KB) reports 1/2 of a wide coalesced read on gfx950 -> doubled; WRITE_SIZE (KB) exact
    p["hbm_bytes_per_launch_corrected"] = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
res["pmc"] = p
for tag in ("trace", "pmc_fetch"):
    try:
        res["line_" + tag] = json.loads([l for l in open(out + "/run_%s.log" % tag) if l.startswith("{")][-1])
    except Exception as e:
        res["line_" + tag] = repr(e)
json.dump(res, open("gpurun_out/r4/colres_profile_%s_n%d.json" % (mat.split(".")[0], n), "w"), indent=1)
print(json.dumps(res, indent=1)[:2500])
PY
for f in $out/trace/*/*kernel_stats.csv; do cp $f gpurun_out/r4/colres_profile_${mat%%.*}_n${n}_kernel_stats.csv; done
