"""lab (round 5): does cutting a few tiles into row-tile pieces (vbs_union.cpp, SPARTA_UNION_SPLIT) pay where whole tiles leave some CUs a tile above the others?
n clusters x 48 rows (true clusters, prepared B, N = 128), n around the number of CUs.   python scripts/lab/r5_union_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
import bench_suite as bs
N = 128
for n_cl in (200, 300, 400, 600, 900, 1300):
    m = bs._clustered(sa, n_cl, 48, 60000, 300, 6, 5)
    rng = np.random.Generator(np.random.PCG64(5)); order = rng.permutation(n_cl * 48)
    g = np.empty(m.rows, np.int64)
    for gi in range(n_cl):
        rows = order[gi * 48:(gi + 1) * 48]; g[rows] = rows.min()
    out = []
    for split in ("0", "1", "2"):
        os.environ["SPARTA_UNION_SPLIT"] = split
        os.environ["SPARTA_SPARSE_MIN_STEPS"] = "0"; os.environ["SPARTA_LAUNCH_NNZ"] = "0"
        d = sa.DeviceVBS.from_csr(m, g, 32, device=0)
        B = torch.rand(d.cols * N, device="cuda") - 0.5
        C = torch.zeros(d.rows * N, device="cuda")
        Bp = d.prepare_b(B, N)
        for _ in range(20): d.spmm_prepared(Bp, C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): d.spmm_prepared(Bp, C)
        e1.record(); torch.cuda.synchronize()
        ui = d.union_info()
        out.append("split=%s %.1f us (steps %d)" % (split, e0.elapsed_time(e1) / 200 * 1e3, ui["steps32"] + ui["steps64"]))
        Bp.close(); d.close()
    print("%5d clusters: %s" % (n_cl, "   ".join(out)), flush=True)
