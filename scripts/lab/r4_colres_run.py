"""One matrix, one width, a fixed number of products of the resident-column kernel: the program scripts/r4_colres_profile.sh puts under rocprofv3.
usage: python3 scripts/lab/r4_colres_run.py <file under tests/golden/ref_data/minitest> <N> <products> [arm: fixed|cluster]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import sparta_amd as sa  # noqa: E402
import bench_suite as S  # noqa: E402

f, n, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
arm = sys.argv[4] if len(sys.argv) > 4 else "fixed"
m = S._sorted_rows(sa, sa.CSR.read_from_edgelist(os.path.join(S.DATA, f), pattern_only=True))
if arm == "fixed":
    g = np.arange(m.rows, dtype=np.int64) // 64
else:
    g = sa.BlockingEngine(blocking_algo=5, tau=0.5, col_block_size=64, row_block_size=64, force_fixed_size=True).GetGrouping(m)
d = sa.DeviceVBS.from_csr(m, g, 64, 64, True, device=0)
B = torch.rand(d.cols * n, device="cuda") - 0.5
C = torch.zeros(d.rows * n, device="cuda")
for _ in range(5):
    d.spmm(B, C, n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    d.spmm(B, C, n)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
cr = d.colres_info()
bound_ms = (d.rows + d.cols) * n * 4.0 / 8e12 * 1e3
print(json.dumps({"matrix": f, "arm": arm, "rows": d.rows, "cols": d.cols, "nnz": m.nztot(), "n_cols": n, "products": reps, "ms": round(ms, 5), "colres": cr,
                  "algorithmic_bytes": (d.rows + d.cols) * n * 4, "bound_ms_at_8TBs": round(bound_ms, 5), "frac_8d": round(bound_ms / ms, 4),
                  "image_bytes_streamed_per_product": cr["entries"] * (2 if cr["unit"] else 6) * -(-n // max(cr["nc"], 1)), "kernel_rev": sa.KERNEL_REV}))
