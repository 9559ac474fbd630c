"""lab (round 5): the benchmark set's `clustered` family on a 16-bit handle (column-compacted tiles through vbs_union_h16_kernel), per-kernel times; fp32 beside it.
   python scripts/lab/r5_union16.py [N ...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
import bench_suite as bs
Ns = [int(x) for x in sys.argv[1:]] or [128, 512]
name, kind, make, kw, w = [c for c in bs.cases(sa) if c[0].startswith("clustered")][0]
m = make()
g = sa.BlockingEngine(col_block_size=w, **kw).GetGrouping(m)
perm = sa.get_permutation(g)
for dt, tdt, nm in ((sa.BF16, torch.bfloat16, "bf16"), (sa.F16, torch.float16, "f16"), (sa.F32, torch.float32, "f32")):
    for union in ("1", "0"):
        os.environ["SPARTA_UNION"] = union
        d = sa.DeviceVBS.from_csr(m, g, w, device=0, dtype=dt)
        for N in Ns:
            ldb = (d.cols + 7) // 8 * 8
            B = (torch.rand(ldb * N, device="cuda") - 0.5).to(tdt)
            C = torch.zeros(d.rows * N, device="cuda")
            for _ in range(5): d.spmm(B, C, N, ldb=ldb)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): d.spmm(B, C, N, ldb=ldb)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 100
            # spot check of a few rows against float64 on the values the handle holds
            worst = 0.0
            for r in np.random.default_rng(1).integers(0, m.rows, 6):
                i = perm[r]; ci = m.colidx[m.rowptr[i]:m.rowptr[i + 1]]
                a = torch.from_numpy(m.vals[m.rowptr[i]:m.rowptr[i + 1]]).to(tdt).double().numpy()
                bb = B.view(N, ldb)[:, torch.from_numpy(ci.astype(np.int64)).cuda()].double().cpu().numpy()
                got = C.view(N, -1)[:, int(r)].double().cpu().numpy()
                worst = max(worst, float((np.abs(got - bb @ a) / (np.abs(bb) @ np.abs(a) + 1e-30)).max()))
            d.set_class_timing(True); d.spmm(B, C, N, ldb=ldb); ct = d.class_times(); d.set_class_timing(False)
            print(json.dumps({"dtype": nm, "union": union, "n_cols": N, "ms": round(ms, 5), "useful_tflops": round(2.0 * m.nztot() * N / ms / 1e9, 1), "kernels_ms": {k: round(v, 4) for k, v in ct.items()},
                              "union_info": d.union_info() if union == "1" else None, "sparse_nnz": d.sparse_info()["nnz"], "check": worst}), flush=True)
        d.close()
