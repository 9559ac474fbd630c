"""Developer script: the path over a small set of matrix families (FEM, banded, clustered, uniform, power-law), one MI355X, fp32,
N = 128: reorder -> handle from the CSR -> product.  Reports useful GFLOP/s and the fraction of the better of the two rooflines a
matrix can be held to: the per-block-row mixed HBM/MFMA bound of its VBS (bench.py: mixed_roofline_seconds) and the HBM bound of
its nonzeros as sparse rows (nnz * (N*4 + 8) + rows * N * 4 bytes at 8 TB/s).
    python scripts/suite_sweep.py [n_cols=128] [out.json]
Besides the markdown table on stdout it writes one JSON record per matrix (seeds, blocking, fill, time, useful / executed rates, which
kernels carried it, fractions of both bounds) -- the committed record is profiles/r2/suite.json."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparta_amd as sa
from bench import mixed_roofline_seconds, PEAK_HBM_GBS

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
OUT = sys.argv[2] if len(sys.argv) > 2 else None
import json
records = []


def clustered(n_groups, rows_per, cols, shared, own, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    rr, cc = [], []
    order = rng.permutation(n_groups * rows_per)                 # rows of a group are scattered: the reorder has to find them
    for gi in range(n_groups):
        base = rng.choice(cols, shared, replace=False)
        for k in range(rows_per):
            c = np.union1d(base[rng.random(shared) < 0.8], rng.choice(cols, own, replace=False))
            rr.append(np.full(len(c), order[gi * rows_per + k])); cc.append(c)
    r, c = np.concatenate(rr), np.concatenate(cc)
    o = np.lexsort((c, r)); r, c = r[o], c[o]
    n = n_groups * rows_per
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))])
    return sa.CSR(n, cols, rowptr, c.astype(np.int32), rng.uniform(-1, 1, len(c)).astype(np.float32))


cases = [
    ("FEM 3D 9x9x257 x3 (cant-like)", lambda: sa.gen.cant_like(), dict(blocking_algo=5, tau=0.6, row_block_size=32, force_fixed_size=True), 32),
    ("FEM 3D 20x20x50 x3", lambda: sa.gen.fem3d(20, 20, 50, 3, 2), dict(blocking_algo=5, tau=0.6, row_block_size=32, force_fixed_size=True), 32),
    ("banded 200k, +-16, 50 %", lambda: sa.gen.banded(200000, 16, density=0.5, seed=4), dict(blocking_algo=7, tau=0.5, minhash_max_rows=32), 32),
    ("clustered 2000 x 48 rows, 300 shared cols", lambda: clustered(2000, 48, 60000, 300, 6, 5), dict(blocking_algo=7, tau=0.6), 32),
    ("uniform 100k x 100k, 2 M nnz", lambda: sa.gen.uniform_random(100000, 100000, 2000000, seed=1), dict(blocking_algo=7, tau=0.6), 64),
    ("R-MAT 2^18, 5.2 M nnz", lambda: sa.gen.rmat(18, 10 << 18, seed=3, symmetrize=True, pattern_only=False), dict(blocking_algo=7, tau=0.4), 64),
    ("R-MAT 2^20, 20 M nnz", lambda: sa.gen.rmat(20, 10 << 20, seed=3, symmetrize=True, pattern_only=False), dict(blocking_algo=7, tau=0.4), 64),
]
print("| matrix | rows | nnz | reorder | block-rows / fill | ms | useful GFLOP/s | carried by | fraction of the better roofline |")
print("|---|---|---|---|---|---|---|---|---|")
for name, make, eng_kw, w in cases:
    m = make()
    t0 = time.time()
    eng = sa.BlockingEngine(col_block_size=w, **eng_kw)
    g = eng.GetGrouping(m)
    t_r = time.time() - t0
    rbs, ff = eng_kw.get("row_block_size", 0), eng_kw.get("force_fixed_size", False)
    d = sa.DeviceVBS.from_csr(m, g, w, rbs, ff, device=0)
    B = torch.rand(d.cols * N, device="cuda") - 0.5
    C = torch.zeros(d.rows * N, device="cuda")
    t_pre = time.time()                                   # untimed pre-roll: a freshly started process runs its first milliseconds below the steady clock (as bench.py)
    while time.time() - t_pre < 0.3:
        for _ in range(20):
            d.spmm(B, C, N)
        torch.cuda.synchronize()
    reps = 200 if m.nztot() < 8e6 else 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        d.spmm(B, C, N)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    d.set_class_timing(True); d.spmm(B, C, N); ct = d.class_times(); d.set_class_timing(False)
    sp = d.sparse_info()
    # rooflines: (a) the VBS as MFMA tiles (needs its nzcount: from the blocking statistics), (b) the nonzeros as sparse rows
    nb_tot, area = None, None
    st = np.zeros(3, np.int64); avg = sa._lib.C.c_float(0)
    gg = np.ascontiguousarray(sa.get_fixed_size_grouping(g, rbs) if ff else g, np.int64)
    sa._lib.check(sa._lib.lib.sparta_blocking_info(m.rows, m.cols, sa.host._p64(np.ascontiguousarray(m.rowptr, np.int64)), sa.host._p32(np.ascontiguousarray(m.colidx, np.int32)),
                                                   sa.host._p64(gg), w, sa.host._p64(st), sa._lib.C.byref(avg)))
    area, nblocks = float(st[0]), float(st[1])
    h_mean = float(avg.value) if avg.value > 0 else 1.0
    br = max(1.0, m.rows / h_mean)
    # mixed bound with the mean block-row (exact per-block-row data would need the VBS itself; this is the same formula on averages)
    t_mfma, _, _ = mixed_roofline_seconds(np.arange(int(br) + 1) * h_mean, np.full(int(br), nblocks / br), w, N, m.cols)
    t_sparse = (m.nztot() * (N * 4.0 + 8.0) + m.rows * N * 4.0) / (PEAK_HBM_GBS * 1e9)
    t_lb = min(t_mfma, t_sparse)
    carried = "sparse rows %.0f %%" % (100.0 * sp["nnz"] / max(m.nztot(), 1)) if sp["nnz"] * 2 > m.nztot() else "MFMA tiles %.0f %%" % (100.0 * (1 - sp["nnz"] / max(m.nztot(), 1)))
    print("| %s | %d | %d | %s, %.2f s | %d / %.3f | %.3f | %.0f | %s | %.2f (%s bound) |" % (
        name, m.rows, m.nztot(), {5: "Keeper 32", 7: "LSH"}[eng_kw["blocking_algo"]] + " tau %.1f w %d" % (eng_kw["tau"], w), t_r, int(br), m.nztot() / max(area, 1.0), ms,
        2.0 * m.nztot() * N / ms / 1e6, carried, t_lb / (ms * 1e-3), "MFMA/HBM mixed" if t_mfma <= t_sparse else "sparse-row HBM"), flush=True)
    exec_area = float(d.info()["nztot"])
    records.append({"matrix": name, "rows": int(m.rows), "cols": int(m.cols), "nnz": int(m.nztot()), "n_cols": N, "dtype": "f32",
                    "blocking": dict(eng_kw, col_block_size=w), "reorder_host_s": round(t_r, 3), "block_rows": int(br), "vbs_area": int(area),
                    "nonzero_blocks": int(nblocks), "fill": round(m.nztot() / max(area, 1.0), 5), "ms": round(ms, 5),
                    "useful_gflops": round(2.0 * m.nztot() * N / ms / 1e6, 1), "executed_gflops_mfma_part": round(2.0 * exec_area * N / ms / 1e6, 1),
                    "kernels_ms": {k: round(float(v_), 5) for k, v_ in ct.items()}, "sparse_nnz": int(sp["nnz"]), "sparse_rows": int(sp["rows"]),
                    "carried_by": carried, "bound_mixed_s": t_mfma, "bound_sparse_rows_s": t_sparse,
                    "frac_of_mixed_bound": round(t_mfma / (ms * 1e-3), 4), "frac_of_sparse_row_bound": round(t_sparse / (ms * 1e-3), 4),
                    "frac_of_better_bound": round(t_lb / (ms * 1e-3), 4), "kernel_rev": sa.KERNEL_REV,
                    "f32_plan": os.environ.get("SPARTA_F32_PLAN", "default")})
    d.close()
if OUT:
    json.dump({"n_cols": N, "dtype": "f32", "device": torch.cuda.get_device_name(0), "records": records}, open(OUT, "w"), indent=1)
