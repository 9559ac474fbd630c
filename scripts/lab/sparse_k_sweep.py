"""Developer script: where is the break-even between MFMA tiles and the sparse-row path?  Times the product for several
SPARTA_SPARSE_K (nonzeros per MFMA step below which a block-row goes to the sparse-row kernels) on matrices of different fill."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cases = []
for nnz_per_row in (2, 8, 32, 128):
    m = sa.gen.uniform_random(32768, 4096, 32768 * nnz_per_row, seed=nnz_per_row)       # 128 block columns of 32: fill = nnz_per_row / 4096
    cases.append(("uniform 32768x4096, %3d nnz/row, fixed 32x32" % nnz_per_row, m, np.arange(m.rows) // 32, 32))
m = sa.gen.rmat(16, 1 << 20, seed=3, symmetrize=True, pattern_only=False)
cases.append(("rmat16 16 e/row, minhash tau .4 w64", m, sa.BlockingEngine(blocking_algo=7, tau=0.4, col_block_size=64).GetGrouping(m), 64))
cases.append(("rmat16 16 e/row, fixed 32x32", m, np.arange(m.rows) // 32, 32))
m = sa.gen.cant_like()
cases.append(("cant-like keeper 32x32", m, sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=32, row_block_size=32, force_fixed_size=True).GetGrouping(m), 32))
for name, m, g, w in cases:
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = torch.rand(v.cols * N, device="cuda") - 0.5
    C = torch.zeros(v.rows * N, device="cuda")
    line = "%-44s fill %.4f nnz/step %7.1f |" % (name, m.nztot() / v.nztot, m.nztot() / (len(v.jab) * (w // 32)))
    for K in ("0", "3", "10", "30", "100", "1e9"):
        os.environ["SPARTA_SPARSE_K"] = K
        d = v.to_device(0)
        for _ in range(5):
            d.spmm(B, C, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            d.spmm(B, C, N)
        e1.record(); torch.cuda.synchronize()
        line += " K=%s: %7.1f us (%d sp rows)" % (K, e0.elapsed_time(e1) / 50 * 1e3, d.info()["sparse_rows"])
        d.close()
    print(line, flush=True)
