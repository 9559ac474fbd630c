"""Developer fuzz: random shapes / blockings / plans through every product path, checked against the on-device exact-order
kernel (bit-identical to the reference's VBR::multiply, tests/test_spmm_gpu.py) with the MFMA tolerance.
usage: python scripts/fuzz_spmm.py [n_cases] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparta_amd as sa

os.environ["SPARTA_SPARSE_MIN_STEPS"] = "0"          # small matrices: let the sparse-row path take the nearly empty block-rows whatever they are worth
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
bad = 0
for case in range(n_cases):
    rows = int(rng.integers(1, 3000))
    cols = int(rng.integers(1, 3000))
    nnz = int(rng.integers(0, max(1, min(rows * cols // 3, 60000))))
    w = int(rng.choice([32, 64, 96, 128, 48, 17]))
    n = int(rng.choice([128, 256, 384, 64, 100]))
    kind = int(rng.integers(0, 4))
    m = sa.gen.uniform_random(rows, cols, nnz, seed=int(rng.integers(1 << 30)))
    if kind == 0:
        g = np.arange(rows) // int(rng.integers(1, 140))
    elif kind == 1:
        g = sa.BlockingEngine(tau=float(rng.uniform(0.1, 0.9)), col_block_size=w).GetGrouping(m)
    elif kind == 2:
        g = sa.BlockingEngine(tau=0.5, col_block_size=w, row_block_size=int(rng.integers(2, 100)), blocking_algo=5).GetGrouping(m)
    else:
        g = rng.integers(0, max(1, rows // int(rng.integers(1, 50))), rows)          # arbitrary grouping, ragged heights
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = torch.from_numpy(sa.gen.dense_rhs(v.cols, n, seed=case)).cuda()
    d = v.to_device(0)
    Ce = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
    d.spmm(B, Ce, n, algo=sa.SPMM_EXACT)
    # scale of the sums for the tolerance: |A| x |B|
    va = sa.VBR.from_arrays(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, np.abs(v.mab))
    da = va.to_device(0)
    Cs = torch.zeros_like(Ce)
    da.spmm(B.abs(), Cs, n, algo=sa.SPMM_EXACT)
    tol = 1e-5 * Cs + 1e-30
    # sparse-row path: off / the library's choice / forced for every block-row that has a block
    for path, spk in [(p_, k_) for p_ in ("auto", "stream", "class", "generic") for k_ in ("0", None)] + [("auto", "1e9")]:
        if spk is None:
            os.environ.pop("SPARTA_SPARSE_K", None)
        else:
            os.environ["SPARTA_SPARSE_K"] = spk
        for align in (("0", "1") if path == "stream" else ("",)):
            os.environ["SPARTA_PATH"] = path
            if align:
                os.environ["SPARTA_STREAM_ALIGN"] = align
            else:
                os.environ.pop("SPARTA_STREAM_ALIGN", None)
            d2 = v.to_device(0)
            for acc in (False, True):
                C = torch.full_like(Ce, 2.0)
                d2.spmm(B, C, n, accumulate=acc)
                torch.cuda.synchronize()
                want = Ce + (2.0 if acc else 0.0)
                err = (C - want).abs()
                if bool((err > tol + (1e-6 if acc else 0.0)).any()):
                    bad += 1
                    print("MISMATCH case", case, dict(rows=rows, cols=cols, nnz=nnz, w=w, n=n, kind=kind, path=path, align=align, acc=acc, sparse_k=spk),
                          "max err", float(err.max()), "last_path", d2.info()["last_path"])
    # handles made straight from the CSR with the per-block split (tiles + sparse rows that add to them on the same block-rows): tolerance
    for kb in ("1", "4", None):
        os.environ.pop("SPARTA_SPARSE_K", None)
        if kb is None:
            os.environ.pop("SPARTA_SPARSE_K_BLOCK", None)
        else:
            os.environ["SPARTA_SPARSE_K_BLOCK"] = kb
        os.environ["SPARTA_PATH"] = "auto"
        os.environ.pop("SPARTA_STREAM_ALIGN", None)
        d3 = sa.DeviceVBS.from_csr(m, g, w, device=0)
        for acc in (False, True):
            C = torch.full_like(Ce, 2.0)
            d3.spmm(B, C, n, accumulate=acc)
            torch.cuda.synchronize()
            err = (C - (Ce + (2.0 if acc else 0.0))).abs()
            if bool((err > tol + (1e-6 if acc else 0.0)).any()):
                bad += 1
                print("MISMATCH (per-block split) case", case, dict(rows=rows, cols=cols, nnz=nnz, w=w, n=n, kind=kind, kblock=kb, acc=acc), "max err", float(err.max()), d3.sparse_info())
    os.environ["SPARTA_SPARSE_K_BLOCK"] = "1e30"         # whole-block-row decisions only from here on: the two builders must agree
    # handles made straight from the CSR (nearly empty block-rows never expanded) must reproduce the two-step handle bit for bit
    for spk in (None, "1e9"):
        if spk is None:
            os.environ.pop("SPARTA_SPARSE_K", None)
        else:
            os.environ["SPARTA_SPARSE_K"] = spk
        os.environ["SPARTA_PATH"] = "stream" if (w % 32 == 0 and n % 128 == 0) else "generic"    # bit-for-bit: the same MFMA path on both handles
        os.environ.pop("SPARTA_STREAM_ALIGN", None)
        d1, d2 = v.to_device(0), sa.DeviceVBS.from_csr(m, g, w, device=0)
        for lay in (sa.COL_MAJOR, sa.ROW_MAJOR):
            Bl = B if lay == sa.COL_MAJOR else B.view(n, v.cols).t().contiguous().view(-1)
            C1, C2 = torch.full_like(Ce, 1.5), torch.full_like(Ce, 1.5)
            d1.spmm(Bl, C1, n, accumulate=True, b_layout=lay, c_layout=lay)
            d2.spmm(Bl, C2, n, accumulate=True, b_layout=lay, c_layout=lay)
            torch.cuda.synchronize()
            if not torch.equal(C1, C2) or d1.sparse_info() != d2.sparse_info():
                bad += 1
                print("MISMATCH from_csr case", case, dict(rows=rows, cols=cols, nnz=nnz, w=w, n=n, kind=kind, sparse_k=spk, lay=lay), d1.sparse_info(), d2.sparse_info())
    os.environ.pop("SPARTA_SPARSE_K", None)
    if w % 32 == 0 and n % 128 == 0:
        for dt, tdt in ((sa.F16, torch.float16), (sa.BF16, torch.bfloat16)):
            os.environ.pop("SPARTA_PATH", None)
            Br = B.view(n, v.cols).to(tdt)
            ldb = (v.cols + 7) // 8 * 8
            Bp = torch.zeros(ldb * n, dtype=tdt, device="cuda")
            Bp.view(n, ldb)[:, :v.cols] = Br
            mab_r = torch.from_numpy(v.mab).to(tdt).float().numpy()
            vr = sa.VBR.from_arrays(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r)
            dr = vr.to_device(0)
            Cr = torch.zeros_like(Ce)
            dr.spmm(Br.float().contiguous().view(-1), Cr, n, algo=sa.SPMM_EXACT)
            dh = v.to_device(0, dtype=dt)
            Ch = torch.full_like(Ce, -1.0)
            dh.spmm(Bp, Ch, n, ldb=ldb)
            torch.cuda.synchronize()
            err = (Ch - Cr).abs()
            if bool((err > tol).any()):
                bad += 1
                print("MISMATCH 16-bit case", case, dict(rows=rows, cols=cols, nnz=nnz, w=w, n=n, kind=kind, dtype=dt), "max err", float(err.max()))
print("fuzz done:", n_cases, "cases,", bad, "mismatches")
