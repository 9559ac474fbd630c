"""lab (round 5): the benchmark set's small hyper-sparse real matrices (social_location, ia-wikiquote: 4-11 nonzeros per row, sparse rows 100 %) at N = 128: ms per product and,
under rocprofv3 --kernel-trace --stats, the kernels behind it.   python scripts/lab/r5_sparse_small.py [reps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
import bench_suite as bs
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = 128
for name, kind, make, kw, w in bs.cases(sa):
    if not any(name.startswith(x) for x in os.environ.get("ONLY", "social_location,ia-wikiquote,uniform").split(",")):
        continue
    m = make()
    g = sa.BlockingEngine(col_block_size=w, **kw).GetGrouping(m)
    d = sa.DeviceVBS.from_csr(m, g, w, device=0)
    B = torch.rand(d.cols * N, device="cuda") - 0.5
    C = torch.zeros(d.rows * N, device="cuda")
    Bp = d.prepare_b(B, N)
    out = {}
    for label, fn in (("product", lambda: d.spmm(B, C, N)), ("prepared", lambda: d.spmm_prepared(Bp, C))):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out[label] = round(e0.elapsed_time(e1) / reps * 1e3, 2)
    print(json.dumps({"matrix": name[:40], "rows": m.rows, "nnz": int(m.nztot()), "us": out, "sparse_info": d.sparse_info()}), flush=True)
    Bp.close(); d.close()
