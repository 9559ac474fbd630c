"""lab (round 5): the benchmark set's `clustered` family through the column-compacted tiles (k_union.hip), on and off, per-kernel times.
   python scripts/lab/r5_union.py [N ...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
import bench_suite as bs

Ns = [int(x) for x in sys.argv[1:]] or [128]
name, kind, make, kw, w = [c for c in bs.cases(sa) if c[0].startswith("clustered")][0]
t0 = time.time(); m = make(); print("generated in %.1f s: %d x %d, %d nnz" % (time.time() - t0, m.rows, m.cols, m.nztot()), flush=True)
for N in Ns:
    for union in ("1", "0"):
        os.environ["SPARTA_UNION"] = union
        r = bs.run_one(sa, torch, name, kind, None, kw, w, N=N, m=m)
        print(json.dumps({k: r[k] for k in ("n_cols", "ms", "ms_prepared_b", "useful_gflops", "frac_8d", "carried_by", "kernels_ms", "host_seconds", "check_max_err", "mfma_tile_area", "sparse_nnz") if k in r} | {"union": union, "union_info": r.get("union_info")}), flush=True)
