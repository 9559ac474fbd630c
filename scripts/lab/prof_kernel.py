"""Developer script: run one cant-like configuration for a few steps (target of rocprofv3 --pmc passes)."""
import sys, os, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
ap = argparse.ArgumentParser()
ap.add_argument("--fixed", type=int, default=0)
ap.add_argument("--tau", type=float, default=0.2)
ap.add_argument("--ncols", type=int, default=128)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
m = sa.gen.cant_like()
g = (np.arange(m.rows) // a.fixed) if a.fixed else sa.BlockingEngine(tau=a.tau, col_block_size=64).GetGrouping(m)
vb = sa.VBR().fill_from_CSR_inplace(m, g, 64)
d = vb.to_device(0)
N = a.ncols
B = torch.from_numpy(sa.gen.dense_rhs(vb.cols, N, seed=3)).cuda()
C = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
for _ in range(a.steps):
    d.spmm(B, C, N)
torch.cuda.synchronize()
print("done", d.info())
