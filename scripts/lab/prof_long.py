import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
N = 128
m = sa.gen.uniform_random(65536, 4096, int(65536 * 4096 * 0.01), seed=1)
vb = sa.VBR().fill_from_CSR_inplace_fixed(m, 64, 64)
d = vb.to_device(0)
B = torch.from_numpy(sa.gen.dense_rhs(vb.cols, N, seed=3)).cuda()
C = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
for _ in range(3):
    d.spmm(B, C, N)
torch.cuda.synchronize()
print(d.info())
