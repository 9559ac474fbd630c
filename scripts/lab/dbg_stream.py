"""Developer check: stream path vs oracle, per block-row error report."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
from oracle import oracle as O
rows, cols, nnz, w, n = [int(x) for x in os.environ.get('DBG_SHAPE', '1500,1500,60000,64,128').split(',')]
m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols + w + n)
g = sa.BlockingEngine(tau=0.5, col_block_size=w).GetGrouping(m) if not os.environ.get('DBG_FIXED') else np.arange(rows) // int(os.environ['DBG_FIXED'])
v = sa.VBR().fill_from_CSR_inplace(m, g, w)
B = sa.gen.dense_rhs(v.cols, n, seed=3)
Co = O.vbr_multiply(v.rows, v.cols, v.block_col_size, v.row_part, v.nzcount, v.jab, v.mab, B, n, None)
d = v.to_device(0)
Ct = torch.full((v.rows * n,), 5.0, dtype=torch.float32, device="cuda")
d.spmm(torch.from_numpy(B).cuda(), Ct, n)
torch.cuda.synchronize()
print(d.info())
got = Ct.cpu().numpy().reshape(n, v.rows)
ref = np.asarray(Co).reshape(n, v.rows)
err = np.abs(got - ref).max(axis=0)
h = np.diff(v.row_part)
bad = 0
for ib in range(v.block_rows):
    e = err[v.row_part[ib]:v.row_part[ib + 1]]
    if e.max() > 1e-3:
        bad += 1
        if bad < 25:
            print("block-row", ib, "h", h[ib], "nb", v.nzcount[ib], "rows", v.row_part[ib], "max err", e.max(), "bad rows", np.nonzero(e > 1e-3)[0][:8], "n bad", (e > 1e-3).sum())
print("bad block rows", bad, "of", v.block_rows, "heights hist", np.bincount(np.minimum(h, 70))[:70].nonzero())
