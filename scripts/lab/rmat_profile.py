"""Developer script: one R-MAT configuration for rocprofv3 (sparse-row path kernels)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lay = sys.argv[3] if len(sys.argv) > 3 else "col"
m = sa.gen.rmat(scale, 10 << scale, seed=3, symmetrize=True, pattern_only=False)
g = sa.BlockingEngine(blocking_algo="minhash", tau=0.4, col_block_size=64).GetGrouping(m)
vb = sa.VBR().fill_from_CSR_inplace(m, g, 64)
d = vb.to_device(0)
print(d.info())
L = sa.COL_MAJOR if lay == "col" else sa.ROW_MAJOR
B = torch.rand(m.cols * N, device="cuda") - 0.5
C = torch.zeros(vb.rows * N, device="cuda")
for _ in range(3):
    d.spmm(B, C, N, b_layout=L, c_layout=L)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    d.spmm(B, C, N, b_layout=L, c_layout=L)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("%s-major B and C, N=%d: %.3f ms, useful %.1f GFLOP/s, %.1f GB/s of nnz * N * 4 bytes" % (lay, N, ms, 2.0 * m.nztot() * N / ms / 1e6, m.nztot() * N * 4.0 / ms / 1e6))
