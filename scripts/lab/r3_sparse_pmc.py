#!/usr/bin/env python3
"""one part of configs[4] (fp16, N = 256), three products: run under `rocprofv3 --pmc FETCH_SIZE -- python3 scripts/lab/r3_sparse_pmc.py [part]` to see what
the sparse-row kernels fetch from HBM next to the bytes they gather (one 512-byte row of B per nonzero)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, sparta_amd as sa
part = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scale, dens, P, N = 23, 1e-4, 8, 256
E = sa.gen.rmat_raw_edges_for_density(scale, dens)
r0, r1 = sa.gen.rmat_cuts(scale, E, P)[part]
m = sa.gen.rmat_rows(scale, E, r0, r1, device=0)
g = np.arange(m.rows) // 64
B = sa.gen.dense_rhs_rows(0, 1 << scale, N, dtype=torch.float16, device=0)
C = torch.zeros(m.rows * N, dtype=torch.float32, device="cuda")
d = sa.DeviceVBS.from_csr(m, g, 64, 64, False, device=0, dtype=sa.F16)
for _ in range(3):
    d.spmm(B, C, N)
torch.cuda.synchronize()
sp = d.sparse_info()
print("part %d rows %d nnz %d sparse nnz %d rows %d: gather bytes %.3f GB, C bytes %.3f GB" % (part, m.rows, m.nztot(), sp["nnz"], sp["rows"],
      sp["nnz"] * (N * 2 + 8) / 1e9, sp["rows"] * N * 4 / 1e9), flush=True)
d.close()
