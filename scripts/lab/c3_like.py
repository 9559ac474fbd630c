"""Developer script: BASELINE.json configs[2] by size -- a power-law graph of 2^21 rows and ~124 M nonzeros (ogbn-products is
2.4 M x 2.4 M, 124 M nnz; not available offline: R-MAT stand-in), B = 256 feature columns in fp16, one MI355X.
Reorder by blocking_algo 7, handle by sparta_vbs_create_from_csr (the dense VBS image of this matrix would be ~25 GB).
    python scripts/c3_like.py [scale=21] [edges=62000000] [n_cols=256] [dtype=f16]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 21
edges = int(sys.argv[2]) if len(sys.argv) > 2 else 62000000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dtype = sys.argv[4] if len(sys.argv) > 4 else "f16"
w = 64
t0 = time.time()
m = sa.gen.rmat(scale, edges, seed=3, symmetrize=True, pattern_only=True)
print("R-MAT scale %d: %d rows, %d nnz, generated in %.1f s" % (scale, m.rows, m.nztot(), time.time() - t0), flush=True)
t0 = time.time()
eng = sa.BlockingEngine(blocking_algo="minhash", tau=0.4, col_block_size=w)
g = eng.GetGrouping(m)
print("reorder (algorithm 7, tau 0.4): %.1f s, %d clusters, %d comparisons" % (time.time() - t0, len(np.unique(g)), eng.comparison_counter), flush=True)
t0 = time.time()
d = sa.DeviceVBS.from_csr(m, g, w, device=0, dtype={"f32": sa.F32, "f16": sa.F16, "bf16": sa.BF16}[dtype])
print("build + upload (from CSR): %.1f s; %s; sparse part %s" % (time.time() - t0, d.info(), d.sparse_info()), flush=True)
tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
ldb = (m.cols + 7) // 8 * 8
B = (torch.rand(ldb * N, device="cuda") - 0.5).to(tdt)
C = torch.zeros(d.rows * N, dtype=torch.float32, device="cuda")
for _ in range(3):
    d.spmm(B, C, N, ldb=ldb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    d.spmm(B, C, N, ldb=ldb)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
esz = 4 if dtype == "f32" else 2
print("%s N=%d: %.3f ms per product, useful %.1f GFLOP/s, %.1f GB/s of nnz * N * %d bytes" % (dtype, N, ms, 2.0 * m.nztot() * N / ms / 1e6, m.nztot() * N * esz / ms / 1e6, esz), flush=True)
# spot check against a direct evaluation of a few rows (float64 on the host, rounded inputs)
Bh = B.view(N, ldb)[:, :m.cols].float().cpu().numpy()
perm = sa.get_permutation(g)
Cc = C.view(N, d.rows).cpu().numpy()
rng = np.random.Generator(np.random.PCG64(1))
worst = 0.0
for r in rng.integers(0, m.rows, 200):
    i = perm[r]
    cols_i = m.colidx[m.rowptr[i]:m.rowptr[i + 1]]
    want = Bh[:, cols_i].astype(np.float64).sum(axis=1)
    scale_ = np.abs(Bh[:, cols_i]).astype(np.float64).sum(axis=1) + 1e-30
    worst = max(worst, float((np.abs(Cc[:, r] - want) / scale_).max()))
print("spot check of 200 rows: max |err| / sum|a||b| = %.2e" % worst)
assert worst < 1e-5
