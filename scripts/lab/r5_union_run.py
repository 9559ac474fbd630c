"""lab (round 5): the benchmark set's `clustered` family (or its small form) through the library's default path, `reps` products; prints one JSON line.
   python scripts/lab/r5_union_run.py N reps"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
import bench_suite as bs
N, reps = int(sys.argv[1]), int(sys.argv[2])
name, kind, make, kw, w = [c for c in bs.cases(sa) if c[0].startswith("clustered")][0]
m = make()
g = sa.BlockingEngine(col_block_size=w, **kw).GetGrouping(m)
d = sa.DeviceVBS.from_csr(m, g, w, device=0)
B = torch.rand(d.cols * N, device="cuda") - 0.5
C = torch.zeros(d.rows * N, device="cuda")
for _ in range(5): d.spmm(B, C, N)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): d.spmm(B, C, N)
e1.record(); torch.cuda.synchronize()
ui, sp = d.union_info(), d.sparse_info()
# algorithmic bytes of the union kernel per launch: the tiles' stored elements (A once) + list entries + tails (8 bytes each) + its rows of C once + B once
alg = ui["area"] * 4 + ui["list_entries"] * 4 + ui["tail_nnz"] * 8 + ui["rows"] * N * 4 + d.cols * N * 4
flops_stored = 2.0 * ui["area"] * N
flops_exec = 2.0 * ui["exec_area"] * N          # (steps x 32 x the tile's rows rounded up to whole MFMA row tiles)
print(json.dumps({"matrix": name, "n_cols": N, "reps": reps, "ms_per_product": e0.elapsed_time(e1) / reps, "union_info": ui, "sparse_info": sp, "nnz": int(m.nztot()),
                  "union_kernel_algorithmic_bytes": alg, "union_kernel_flops_on_stored_elements": flops_stored, "union_kernel_flops_executed": flops_exec}))
