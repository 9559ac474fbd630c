#!/bin/bash
# usage (GPU box, via gpurun): scripts/pmc_rmat.sh  -- HBM traffic of the sparse-row kernels on the R-MAT workload (counters only, separate passes)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
out=gpurun_out/pmc_rmat
mkdir -p $out
timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/fetch -- python3 bench.py --workload rmat --ncols 256 --steps 5 --warmup 2 --no-cpu-baseline > $out/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $out/write -- python3 bench.py --workload rmat --ncols 256 --steps 5 --warmup 2 --no-cpu-baseline > $out/write.log 2>&1
python3 - "$out" <<'PY'
import sys, glob, csv, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = next((k for k in ("sparse_rows_kernel", "sparse_segments_kernel", "sparse_reduce_kernel", "b_to_row_major_kernel", "sparse_c_scatter_kernel") if k in n), None)
        if key: agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
tot = 0.0
for k, d in agg.items():
    p = {c: sum(v) / len(v) for c, v in d.items()}
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE (KB) reports 1/2 of a wide coalesced read on gfx950 -> doubled; WRITE_SIZE (KB) exact
    p["hbm_bytes_per_launch"] = (2.0 * p.get("FETCH_SIZE", 0.0) + p.get("WRITE_SIZE", 0.0)) * 1024.0
    res[k] = p
    tot += p["hbm_bytes_per_launch"]
res["all_sparse_row_kernels_hbm_bytes_per_step"] = tot
json.dump(res, open(out + "/rmat_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
