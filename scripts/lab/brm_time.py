"""Developer script: flagship SpMM with B column-major vs row-major (C column-major / row-major)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
m = sa.gen.cant_like()
N, w = 128, 32
g = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=32, force_fixed_size=True).GetGrouping(m)
vb = sa.VBR().fill_from_CSR_inplace(m, g, w, 32, True)
d = vb.to_device(0)
B = torch.from_numpy(sa.gen.dense_rhs(vb.cols, N, seed=3)).cuda()
Br = B.view(N, vb.cols).t().contiguous().view(-1)
C = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
for name, b, bl, cl in (("B col C col", B, sa.COL_MAJOR, sa.COL_MAJOR), ("B row C col", Br, sa.ROW_MAJOR, sa.COL_MAJOR), ("B row C row", Br, sa.ROW_MAJOR, sa.ROW_MAJOR)):
    for _ in range(50):
        d.spmm(b, C, N, b_layout=bl, c_layout=cl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000):
        d.spmm(b, C, N, b_layout=bl, c_layout=cl)
    e1.record(); torch.cuda.synchronize()
    print("%s: %.2f us" % (name, e0.elapsed_time(e1)))
