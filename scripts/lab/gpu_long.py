"""Developer experiment: long tiles (many blocks per block-row) to separate per-tile overhead from steady-state rate."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
N = 128
for rows, cols, dens in ((65536, 8192, 0.01), (65536, 2048, 0.02), (262144, 1024, 0.02)):
    m = sa.gen.uniform_random(rows, cols, int(rows * cols * dens), seed=1)
    vb = sa.VBR().fill_from_CSR_inplace_fixed(m, 64, 64)
    d = vb.to_device(0)
    B = torch.from_numpy(sa.gen.dense_rhs(vb.cols, N, seed=3)).cuda()
    C = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
    for _ in range(3):
        d.spmm(B, C, N)
    torch.cuda.synchronize()
    ts = [d.spmm(B, C, N, timed=True) for _ in range(10)]
    t = float(np.median(ts))
    d.set_class_timing(True); d.spmm(B, C, N); kt = d.class_times(); mhz = d.clock_mhz(); d.set_class_timing(False)
    print(kt, mhz)
    print('%dx%d blocks/tile %.1f tiles %d: %.1f us exec %.1f TF' % (rows, cols, len(vb.jab) / vb.block_rows, vb.block_rows, t * 1e3, 2 * vb.nztot * N / t / 1e9))
