"""Developer script: BASELINE.json configs[3] in miniature -- R-MAT power-law matrix, reorder OFF (fixed 64-row blocks) vs
reorder ON (blocking_algo 7, the LSH-bucketed clustering; the exact scans cannot handle this many rows), VBS build, SpMM.
    python scripts/rmat_reorder_sweep.py [scale=20] [edges_per_row=10] [n_cols=256] [dtype=f32|f16|bf16]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
epr = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
w = 64
t0 = time.time()
m = sa.gen.rmat(scale, epr << scale, seed=3, symmetrize=True, pattern_only=False)
print("R-MAT scale %d: %d rows, %d nnz (%.4f %%), generated in %.1f s" % (scale, m.rows, m.nztot(), 100.0 * m.nztot() / m.rows ** 2, time.time() - t0), flush=True)
gpu = torch.cuda.is_available()
tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
ldb = (m.cols + 7) // 8 * 8
if gpu:
    B = (torch.rand(ldb * N, device="cuda") - 0.5).to(tdt)
for name, eng in (("reorder off (fixed 64)", sa.BlockingEngine(blocking_algo="fixed_size", row_block_size=64, col_block_size=w)),
                  ("minhash tau 0.4", sa.BlockingEngine(blocking_algo="minhash", tau=0.4, col_block_size=w)),
                  ("minhash tau 0.6", sa.BlockingEngine(blocking_algo="minhash", tau=0.6, col_block_size=w)),
                  ("minhash tau 0.6 <=64 rows", sa.BlockingEngine(blocking_algo="minhash", tau=0.6, col_block_size=w, minhash_max_rows=64)),
                  ("minhash tau 0.8", sa.BlockingEngine(blocking_algo="minhash", tau=0.8, col_block_size=w))):
    t0 = time.time(); g = eng.GetGrouping(m); t_r = time.time() - t0
    if name.startswith("reorder off"):                       # the fixed grid's stored area, before building it
        rows_of = np.repeat(np.arange(m.rows, dtype=np.int64), np.diff(m.rowptr))
        nblk = len(np.unique((rows_of // 64) * ((m.cols + w - 1) // w) + m.colidx // w))
        del rows_of
        if nblk * 64.0 * w > 4e9:
            print("%-26s %d nonzero blocks -> area %.3e (%.0f GB in fp32), fill %.5f: not built" % (name, nblk, nblk * 64.0 * w, nblk * 64.0 * w * 4 / 1e9,
                                                                                               m.nztot() / (nblk * 64.0 * w)), flush=True)
            continue
    t0 = time.time(); vb = sa.VBR().fill_from_CSR_inplace(m, g, w); t_b = time.time() - t0
    line = "%-26s reorder %6.1f s build %5.1f s | block-rows %7d blocks %9d area %.3e fill %.4f" % (
        name, t_r, t_b, vb.block_rows, len(vb.jab), vb.nztot, m.nztot() / vb.nztot)
    if gpu and vb.nztot * 4 < 60e9:
        d = vb.to_device(0, dtype={"f32": sa.F32, "f16": sa.F16, "bf16": sa.BF16}[dtype])
        C = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
        for _ in range(3):
            d.spmm(B, C, N, ldb=ldb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            d.spmm(B, C, N, ldb=ldb)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += " | %s N=%d: %8.3f ms  useful %8.1f GFLOP/s  executed %6.1f TFLOP/s  path %d" % (
            dtype, N, ms, 2.0 * m.nztot() * N / ms / 1e6, 2.0 * vb.nztot * N / ms / 1e9, d.info()["last_path"])
        d.close(); del C
    print(line, flush=True)
    del vb
