#!/usr/bin/env python3
"""lab: what bounds the sparse-row gather -- the same number of nonzeros per row over column spans from L2-resident to HBM-sized.
   python scripts/lab/r4_sparse_gather.py [f16|bf16|f32] [N]
Every row has NNZ_ROW nonzeros at random columns inside [0, span); all block-rows are forced onto the sparse-row kernels
(SPARTA_SPARSE_K / _K_BLOCK huge); B is prepared once (no transposes in the timed region).  Prints ms, Gnnz/s, gather TB/s."""
import os, sys, time
import numpy as np
os.environ["SPARTA_SPARSE_K"] = "1e9"
os.environ["SPARTA_SPARSE_K_BLOCK"] = "1e9"
os.environ.setdefault("SPARTA_SPARSE_MIN_STEPS", "0")
os.environ.setdefault("SPARTA_LAUNCH_NNZ", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, sparta_amd as sa

dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sdt, tdt = {"f16": (sa.F16, torch.float16), "bf16": (sa.BF16, torch.bfloat16), "f32": (sa.F32, torch.float32)}[dt]
esz = 4 if dt == "f32" else 2
rows, nnz_row = 32768, 1024
rng = np.random.Generator(np.random.PCG64(5))
for span in (4096, 32768, 262144, 2097152, 8388608):
    cols = span
    # nnz_row distinct sorted columns per row: a random start + a random stride pattern is enough for a cache experiment
    c = np.sort(rng.integers(0, span, size=(rows, nnz_row), dtype=np.int64), axis=1)
    c += np.arange(nnz_row)[None, :] * 0                                     # (duplicates are merged by the CSR builder below)
    r = np.repeat(np.arange(rows, dtype=np.int64), nnz_row)
    m = sa.gen._csr_from_coo(rows, cols, r, c.reshape(-1), rng.uniform(-1, 1, rows * nnz_row).astype(np.float32))
    nnz = int(m.nztot())
    g = np.arange(rows) // 64
    d = sa.DeviceVBS.from_csr(m, g, 64, 64, False, device=0, dtype=sdt)
    info = d.info()
    B = (torch.rand(cols * N, device="cuda") - 0.5).to(tdt)
    C = torch.zeros(rows * N, dtype=torch.float32, device="cuda")
    Bp = d.prepare_b(B, N)
    for _ in range(3): d.spmm_prepared(Bp, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): d.spmm_prepared(Bp, C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # sampled check against float64 on the CSR
    worst = 0.0
    perm = sa.get_permutation(g)
    for rr in rng.integers(0, rows, 6):
        i = perm[rr]; cj = m.colidx[m.rowptr[i]:m.rowptr[i + 1]].astype(np.int64)
        a = torch.from_numpy(m.vals[m.rowptr[i]:m.rowptr[i + 1]]).to(tdt).double().numpy()
        bb = B.view(N, cols)[:, torch.from_numpy(cj).cuda()].double().cpu().numpy()
        got = C.view(N, rows)[:, int(rr)].double().cpu().numpy()
        worst = max(worst, float((np.abs(got - bb @ a) / (np.abs(bb) @ np.abs(a) + 1e-30)).max()))
    print("%s N=%d span %8d (B %7.1f MB): nnz %d sparse_rows %d | %.3f ms  %.1f Gnnz/s  gather %.2f TB/s | check %.1e" %
          (dt, N, span, span * N * esz / 1e6, nnz, info["sparse_rows"], ms, nnz / ms / 1e6, nnz * N * esz / ms / 1e9, worst), flush=True)
    d.close()
    del B, C, Bp
    torch.cuda.empty_cache()
