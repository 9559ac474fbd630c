"""lab (round 5): what bounds the column-compacted tile kernel -- the same tile shapes (2000 clusters x 48 rows, ~300 list columns + 6 tail columns per row) with B small enough
to live in every L2 (the clusters draw their columns from 2 000 instead of 60 000: |B| = 1 MB at N = 128) against the benchmark set's family (|B| = 31 MB).  python scripts/lab/r5_union_l2.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
import bench_suite as bs
N = 128
for cols in [int(x) for x in os.environ.get('COLS', '60000,2000').split(',')]:
    m = bs._clustered(sa, 2000, 48, cols, 300, 6, 5)
    # the true clusters (the generator scatters the rows; group id = smallest row of the cluster): the LSH is not the subject here
    rng = np.random.Generator(np.random.PCG64(5)); order = rng.permutation(2000 * 48)
    g = np.empty(m.rows, np.int64)
    for gi in range(2000):
        rows = order[gi * 48:(gi + 1) * 48]; g[rows] = rows.min()
    d = sa.DeviceVBS.from_csr(m, g, 32, device=0)
    B = torch.rand(d.cols * N, device="cuda") - 0.5
    C = torch.zeros(d.rows * N, device="cuda")
    Bp = d.prepare_b(B, N)
    for probe in [int(x) for x in os.environ.get('PROBES', '0,8,4').split(',')]:
        os.environ["SPARTA_UNION_PROBE"] = str(probe)
        for _ in range(20): d.spmm_prepared(Bp, C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): d.spmm_prepared(Bp, C)
        e1.record(); torch.cuda.synchronize()
        print("columns %6d (|B| = %5.1f MB) probe %d: %.1f us per product   %s" % (cols, cols * N * 4 / 1e6, probe, e0.elapsed_time(e1) / 200 * 1e3, d.union_info() if probe == 0 else ""), flush=True)
    os.environ["SPARTA_UNION_PROBE"] = "0"
    Bp.close(); d.close()
