"""The column-compacted ("union-pattern") tiles on the GPU (sparta_amd/csrc/k_union.hip): clusters whose rows share columns, kept as dense (rows x |union|) tiles + column
lists -- the VBS the reference builds at small block widths (/root/reference/src/general/vbr.cpp:177-228) -- and multiplied on the matrix cores against the gathered rows
of B.  Through the C-ABI (sparta_vbs_create_from_csr / sparta_vbs_spmm / sparta_vbs_prepare_b) against
  * the oracle's restatement of the reference's VBR::multiply (vbr.cpp:323-372) at -b 1, 2, 4, 8, EVERY element, bit for bit on small-integer data (every order of
    additions gives the same bits there), and within 1e-5 * sum|a||b| on real data (the tolerance of the MFMA kernels' tests: the matrix cores add in another order);
  * float64 over shapes x operand widths x layouts x leading dimensions x accumulate x pointer spaces."""
import numpy as np
import pytest

import sparta_amd as sa
from oracle import oracle
from test_union_host import clustered, true_grouping, reference_product

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


@pytest.fixture(autouse=True)
def small_matrices_keep_their_decisions(monkeypatch):
    monkeypatch.setenv("SPARTA_SPARSE_MIN_STEPS", "0")
    monkeypatch.setenv("SPARTA_LAUNCH_NNZ", "0")
    monkeypatch.setenv("SPARTA_COLRES", "0")          # the sparse remainder through the row gather (test_colres_gpu.py covers the resident-column kernel behind tiles)


def _dense(m):
    import scipy.sparse as sp
    v = np.ones(len(m.colidx), np.float32) if m.vals is None else m.vals
    return sp.csr_matrix((v, m.colidx, m.rowptr), shape=(m.rows, m.cols))


def _want(m, g, B, n, C0=None, rows_pad=None):
    """float64 product in the handle's row order, column-major flat, and its absolute bound"""
    A = _dense(m)
    perm = np.asarray(sa.get_permutation(g), np.int64)
    rows_pad = m.rows if rows_pad is None else rows_pad
    Bm = B.reshape(n, m.cols).T.astype(np.float64)
    Ap = A[perm].astype(np.float64)
    C, bound = np.zeros((rows_pad, n)), np.zeros((rows_pad, n))
    C[:m.rows], bound[:m.rows] = np.asarray(Ap @ Bm), np.asarray(abs(Ap) @ np.abs(Bm))
    if C0 is not None:
        C0m = C0.reshape(n, rows_pad).T.astype(np.float64)
        C, bound = C + C0m, bound + np.abs(C0m)
    return C.T.reshape(-1), bound.T.reshape(-1)


def _product(torch, d, B, n, ldb=None, ldc=None, C0=None, **kw):
    rows, cols = d.rows, d.cols
    ldb, ldc = ldb or cols, ldc or rows
    Bp = np.full((n, ldb), 3.0e38, np.float32)                              # padding of the leading dimensions: never read / never written
    Bp[:, :cols] = B.reshape(n, cols)
    Cp = np.full((n, ldc), -5.0, np.float32)
    if C0 is not None:
        Cp[:, :rows] = C0.reshape(n, rows)
    Bt, Ct = torch.from_numpy(Bp.reshape(-1)).cuda(), torch.from_numpy(Cp.reshape(-1)).cuda()
    d.spmm(Bt, Ct, n, accumulate=C0 is not None, ldb=ldb, ldc=ldc, **kw)
    torch.cuda.synchronize()
    got = Ct.cpu().numpy().reshape(n, ldc)
    assert np.all(got[:, rows:] == -5.0), "the product wrote into the padding of C"
    return np.ascontiguousarray(got[:, :rows]).reshape(-1)


@pytest.mark.parametrize("w", [1, 2, 4, 8])
@pytest.mark.parametrize("rows_per", [5, 48, 70])
def test_small_block_widths_every_element_against_the_reference_product(w, rows_per):
    torch = _torch()
    m, order = clustered(40, rows_per, 5000, 100, 3, seed=3 * rows_per + w, integer=True)
    g = true_grouping(order, rows_per)
    d = sa.DeviceVBS.from_csr(m, g, w, device=0)
    ui = d.union_info()
    assert ui["tiles32"] + ui["tiles64"] > 0 and ui["nnz"] > 0.7 * m.nztot(), ui
    n = 128
    B = np.random.default_rng(w).integers(-3, 4, m.cols * n).astype(np.float32)
    got = _product(torch, d, B, n)
    v = oracle.OracleVBR(m.rows, m.cols, m.rowptr, m.colidx, m.vals, g, w)
    ref = oracle.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    assert np.array_equal(got, ref)                                          # bit for bit: the reference's VBR::multiply on the VBS it builds at this block width
    d.close()


@pytest.mark.parametrize("n", [128, 256, 72, 5, 384])
@pytest.mark.parametrize("layouts", [(sa.COL_MAJOR, sa.COL_MAJOR), (sa.ROW_MAJOR, sa.COL_MAJOR), (sa.COL_MAJOR, sa.ROW_MAJOR), (sa.ROW_MAJOR, sa.ROW_MAJOR)])
def test_against_float64_over_widths_and_layouts(n, layouts):
    torch = _torch()
    bl, cl = layouts
    m, order = clustered(60, 48, 9000, 150, 5, seed=n)
    g = true_grouping(order, 48)
    d = sa.DeviceVBS.from_csr(m, g, 32, device=0)
    assert d.union_info()["tiles64"] == 60
    rng = np.random.default_rng(n + 1)
    B = rng.uniform(-1, 1, m.cols * n).astype(np.float32)                    # column-major image; the row-major operand is its transpose
    want, bound = _want(m, g, B, n)
    Bt = torch.from_numpy(B if bl == sa.COL_MAJOR else np.ascontiguousarray(B.reshape(n, m.cols).T).reshape(-1)).cuda()
    for acc in (False, True):
        C0 = rng.uniform(-1, 1, m.rows * n).astype(np.float32)
        Ct = torch.from_numpy(C0 if cl == sa.COL_MAJOR else np.ascontiguousarray(C0.reshape(n, m.rows).T).reshape(-1)).cuda()
        d.spmm(Bt, Ct, n, accumulate=acc, b_layout=bl, c_layout=cl)
        torch.cuda.synchronize()
        got = Ct.cpu().numpy()
        got = got if cl == sa.COL_MAJOR else np.ascontiguousarray(got.reshape(m.rows, n).T).reshape(-1)
        w_, b_ = (want + C0, bound + np.abs(C0)) if acc else (want, bound)
        assert np.max(np.abs(got - w_) / (b_ + 1e-30)) < TOL, (n, layouts, acc)
    d.close()


@pytest.mark.parametrize("rows_per,n_groups", [(3, 400), (17, 90), (32, 50), (33, 40), (64, 30), (100, 20), (200, 9)])
def test_cluster_heights_and_padded_leading_dimensions(rows_per, n_groups):
    torch = _torch()
    m, order = clustered(n_groups, rows_per, 7000, 120, 4, seed=rows_per)
    g = true_grouping(order, rows_per)
    d = sa.DeviceVBS.from_csr(m, g, 64, device=0)
    ui = d.union_info()
    assert ui["nnz"] > 0.5 * m.nztot(), ui
    n = 200
    rng = np.random.default_rng(rows_per)
    B = rng.uniform(-1, 1, m.cols * n).astype(np.float32)
    want, bound = _want(m, g, B, n)
    got = _product(torch, d, B, n, ldb=m.cols + 13, ldc=m.rows + 7)
    assert np.max(np.abs(got - want) / (bound + 1e-30)) < TOL
    C0 = rng.uniform(-1, 1, m.rows * n).astype(np.float32)
    want2, bound2 = _want(m, g, B, n, C0)
    got = _product(torch, d, B, n, ldc=m.rows + 64, C0=C0)
    assert np.max(np.abs(got - want2) / (bound2 + 1e-30)) < TOL
    d.close()


def _clusters_of_heights(heights, cols, shared, own, seed, integer=True):
    """one cluster per entry of `heights` (rows in cluster order, not scattered): rows of a cluster share `shared` columns (each present with probability 0.8) + `own` of their own"""
    rng = np.random.default_rng(seed)
    rr, cc, g, r0 = [], [], [], 0
    for h in heights:
        base = rng.choice(cols, shared, replace=False)
        for k in range(h):
            c = np.union1d(base[rng.random(shared) < 0.8], rng.choice(cols, own, replace=False))
            rr.append(np.full(len(c), r0 + k)); cc.append(c)
        g += [r0] * h
        r0 += h
    r, c = np.concatenate(rr), np.concatenate(cc)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=r0))]).astype(np.int64)
    v = rng.integers(1, 4, len(c)).astype(np.float32) if integer else rng.uniform(-1, 1, len(c)).astype(np.float32)
    return sa.CSR(r0, cols, rowptr, c.astype(np.int32), v), np.asarray(g, np.int64)


@pytest.mark.parametrize("dtype", [sa.F32, sa.BF16], ids=["f32", "bf16"])
def test_every_tile_height_in_one_handle(dtype):
    """tiles of 1..4 MFMA row tiles (fp32: 16-row granularity) in ONE launch: every workgroup walks its tiles type by type, tallest first (k_union.hip: four bodies behind
    barriers; vbs_union.cpp: one set of workers over all types).  Every element against the oracle's VBR::multiply at block width 1, bit for bit (small integers)."""
    torch = _torch()
    heights = [5, 16, 17, 31, 33, 40, 48, 49, 60, 64, 70, 3, 25, 57] * 6                # (70: a part of 64 + a part of 6)
    m, g = _clusters_of_heights(heights, 5000, 90, 2, seed=13)
    w_handle = 1 if dtype == sa.F32 else 32
    d = sa.DeviceVBS.from_csr(m, g, w_handle, device=0, dtype=dtype)
    ui = d.union_info()
    assert ui["tiles32"] >= 6 * 6 and ui["tiles64"] >= 7 * 6 and ui["nnz"] > 0.9 * m.nztot(), ui
    if dtype == sa.F32:
        # the kernel pays for whole 16-row tiles: 5 -> 16, 17 -> 32, 33 / 40 / 48 -> 48, 49 / 60 / 64 -> 64 rows per step
        assert ui["row_tile"] == 16 and ui["exec_area"] < 1.45 * ui["area"], ui
    n = 136
    B = np.random.default_rng(4).integers(-3, 4, m.cols * n).astype(np.float32)
    v = oracle.OracleVBR(m.rows, m.cols, m.rowptr, m.colidx, m.vals, g, 1)
    ref = oracle.vbr_multiply(v.rows, v.cols, 1, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    if dtype == sa.F32:
        assert np.array_equal(_product(torch, d, B, n), ref)
        C0 = np.random.default_rng(1).integers(-3, 4, m.rows * n).astype(np.float32)
        assert np.array_equal(_product(torch, d, B, n, C0=C0), ref + C0)
    else:
        Bt, ldb = _b16(torch, B, dtype, n, m.cols)
        Ct = torch.full((m.rows * n,), 7.0, device="cuda")
        d.spmm(Bt, Ct, n, ldb=ldb)
        torch.cuda.synchronize()
        assert np.array_equal(Ct.cpu().numpy(), ref)
    d.close()


def test_mixed_image_tiles_union_tiles_and_sparse_rows_in_one_product():
    torch = _torch()
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    m1, order = clustered(60, 40, 4096, 120, 4, seed=11, scatter=False)
    A = _dense(m1).tolil()
    A[1000:1040, :] = 0
    A[1000:1040, 512:1536] = rng.uniform(-1, 1, (40, 1024)).astype(np.float32)          # a dense cluster: stays w-wide MFMA tiles (a list of 1024 columns costs more than 32 blocks)
    A[2000:2040, :] = 0                                                               # an empty cluster
    A = A.tocsr(); A.eliminate_zeros(); A.sort_indices()
    m = sa.CSR(A.shape[0], A.shape[1], A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float32))
    g = np.arange(m.rows) // 40 * 40
    d = sa.DeviceVBS.from_csr(m, g, 32, device=0)
    ui, si, info = d.union_info(), d.sparse_info(), d.info()
    assert ui["tiles64"] > 0 and ui["tail_nnz"] > 0 and info["nztot"] > 0, (ui, si, info)
    for n in (128, 130):
        B = rng.uniform(-1, 1, m.cols * n).astype(np.float32)
        want, bound = _want(m, g, B, n)
        got = _product(torch, d, B, n)
        assert np.max(np.abs(got - want) / (bound + 1e-30)) < TOL
        # host pointers (the reference back-ends' contract), C += A B
        C0 = rng.uniform(-1, 1, m.rows * n).astype(np.float32)
        Ch = C0.copy()
        d.spmm_host(B, n, Ch, accumulate=True)
        want2, bound2 = _want(m, g, B, n, C0)
        assert np.max(np.abs(Ch - want2) / (bound2 + 1e-30)) < TOL
    d.close()


@pytest.mark.parametrize("longest_tail", [False, True], ids=["overflow-to-sparse-rows", "tails-of-31"])
@pytest.mark.parametrize("n", [128, 40])
def test_tails_and_their_overflow_into_sparse_rows(monkeypatch, n, longest_tail):
    torch = _torch()
    m, order = clustered(40, 48, 6000, 100, 20, seed=21, integer=True)      # 20 columns of their own per row: 16 in the tile's tail, the rest sparse rows that add ...
    g = true_grouping(order, 48)
    if not longest_tail:
        monkeypatch.setenv("SPARTA_UNION_STRAGGLERS", "0")                  # ... unless that rest is all the sparse-row kernels would run for: then tails of up to 31 (the library's default here)
    d = sa.DeviceVBS.from_csr(m, g, 1, device=0)
    ui, si = d.union_info(), d.sparse_info()
    assert ui["tail_nnz"] > 0 and (si["nnz"] == 0) == longest_tail and ui["nnz"] + si["nnz"] == m.nztot(), (ui, si)
    B = np.random.default_rng(n).integers(-3, 4, m.cols * n).astype(np.float32)
    v = oracle.OracleVBR(m.rows, m.cols, m.rowptr, m.colidx, m.vals, g, 1)
    ref = oracle.vbr_multiply(v.rows, v.cols, 1, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    assert np.array_equal(_product(torch, d, B, n), ref)
    C0 = np.random.default_rng(1).integers(-3, 4, m.rows * n).astype(np.float32)
    assert np.array_equal(_product(torch, d, B, n, C0=C0), ref + C0)         # accumulate
    d.close()


def test_gathered_b_and_a_row_major_b_that_is_not_16_byte_aligned():
    """the tiles read a ROW-major B in 16-byte pieces: an all-gather-shaped B is transposed into the handle's copy like a column-major one; a row-major B whose rows do not
    start on 16-byte boundaries (odd leading dimension) is copied once per product into an aligned image"""
    torch = _torch()
    w, world = 32, 2
    m, order = clustered(40, 48, 2 * 2048, 100, 4, seed=31)
    g = true_grouping(order, 48)
    d = sa.DeviceVBS.from_csr(m, g, w, device=0)
    assert d.union_info()["tiles64"] == 40
    n = 136
    rng = np.random.default_rng(8)
    B = rng.uniform(-1, 1, m.cols * n).astype(np.float32)
    want, bound = _want(m, g, B, n)
    # gathered: `world` column-major slabs of shard_rows x n back to back
    shard_rows = m.cols // world
    Bg = np.concatenate([np.ascontiguousarray(B.reshape(n, m.cols)[:, r * shard_rows:(r + 1) * shard_rows]).reshape(-1) for r in range(world)])
    Ct = torch.full((m.rows * n,), 7.0, device="cuda")
    d.spmm_gathered(torch.from_numpy(Bg).cuda(), shard_rows, Ct, n)
    torch.cuda.synchronize()
    assert np.max(np.abs(Ct.cpu().numpy() - want) / (bound + 1e-30)) < TOL
    # row-major with ldb = n + 1 (rows 4 bytes off every time)
    ldb = n + 1
    Brm = np.full((m.cols, ldb), 3.0e38, np.float32)
    Brm[:, :n] = B.reshape(n, m.cols).T
    Ct = torch.full((m.rows * n,), 7.0, device="cuda")
    d.spmm(torch.from_numpy(Brm.reshape(-1)).cuda(), Ct, n, b_layout=sa.ROW_MAJOR, ldb=ldb)
    torch.cuda.synchronize()
    assert np.max(np.abs(Ct.cpu().numpy() - want) / (bound + 1e-30)) < TOL
    d.close()


def test_prepared_b_gives_the_same_bits_and_force_fixed_padding():
    torch = _torch()
    m, order = clustered(50, 24, 3000, 80, 3, seed=9)
    g = true_grouping(order, 24)
    d = sa.DeviceVBS.from_csr(m, g, 16, 32, True, device=0)                  # -F 1: rows padded to multiples of 32
    assert d.rows == (m.rows + 31) // 32 * 32 and d.union_info()["nnz"] > 0
    n = 136
    rng = np.random.default_rng(5)
    B = np.zeros((n, d.cols), np.float32)
    B[:, :m.cols] = rng.uniform(-1, 1, (n, m.cols)).astype(np.float32)
    Bt = torch.from_numpy(B.reshape(-1)).cuda()
    C1, C2 = torch.zeros(d.rows * n, device="cuda"), torch.full((d.rows * n,), 9.0, device="cuda")
    d.spmm(Bt, C1, n)
    Bp = d.prepare_b(Bt, n)
    d.spmm_prepared(Bp, C2)
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)
    want, bound = _want(m, g, np.ascontiguousarray(B[:, :m.cols]).reshape(-1), n, rows_pad=d.rows)
    assert np.max(np.abs(C1.cpu().numpy() - want) / (bound + 1e-30)) < TOL
    Bp.close(); d.close()


def test_clustering_found_by_the_library_on_the_suite_family():
    """the benchmark set's `clustered` family in small: rows of a group scattered, blocking_algo 7 finds them, the product is the tiles' (bench_suite.py)"""
    torch = _torch()
    import bench_suite
    m = bench_suite._clustered(sa, 150, 48, 20000, 300, 6, 5)
    g = sa.BlockingEngine(col_block_size=32, blocking_algo=7, tau=0.6).GetGrouping(m)
    d = sa.DeviceVBS.from_csr(m, g, 32, device=0)
    ui = d.union_info()
    assert ui["nnz"] > 0.8 * m.nztot(), ui
    n = 128
    B = np.random.default_rng(1).uniform(-1, 1, m.cols * n).astype(np.float32)
    want, bound = _want(m, g, B, n)
    got = _product(torch, d, B, n)
    assert np.max(np.abs(got - want) / (bound + 1e-30)) < TOL
    # switched off: the same product from the sparse rows, same tolerance
    import os
    os.environ["SPARTA_UNION"] = "0"
    try:
        d0 = sa.DeviceVBS.from_csr(m, g, 32, device=0)
    finally:
        os.environ.pop("SPARTA_UNION")
    assert d0.union_info()["nnz"] == 0
    got0 = _product(torch, d0, B, n)
    assert np.max(np.abs(got0 - want) / (bound + 1e-30)) < TOL
    d.close(); d0.close()


# ---- 16-bit handles: the same tiles through `v_mfma_f32_32x32x16_{f16,bf16}` (panel of B read with the LDS transpose loads) -----------------------------------------------------

def _round16(x, dtype):
    """fp32 -> fp16 / bf16 -> fp32, round to nearest even"""
    x = np.ascontiguousarray(x, np.float32)
    if dtype == sa.F16:
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def _b16(torch, B, dtype, n, cols, ldb=None):
    """column-major 16-bit device image of B (ld = ldb)"""
    ldb = ldb or (cols + 7) // 8 * 8
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :cols] = torch.from_numpy(B.reshape(n, cols)).cuda().to(tdt)
    return Bt, ldb


@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("rows_per,n", [(48, 128), (20, 256), (70, 72), (5, 130)])
def test_16bit_tiles_every_element_against_the_reference_product(dtype, rows_per, n):
    torch = _torch()
    m, order = clustered(40, rows_per, 5000, 100, 3, seed=rows_per + n, integer=True)      # small integers: exact in both 16-bit types, every order of additions the same bits
    g = true_grouping(order, rows_per)
    d = sa.DeviceVBS.from_csr(m, g, 32, device=0, dtype=dtype)                           # (16-bit handles take block widths that are multiples of 32; the tiles hold columns, not blocks:
    ui = d.union_info()                                                                  #  the product is the one of the reference's VBS at -b 1 below)
    assert ui["tiles32"] + ui["tiles64"] > 0 and ui["nnz"] > 0.7 * m.nztot(), ui
    B = np.random.default_rng(n).integers(-3, 4, m.cols * n).astype(np.float32)
    Bt, ldb = _b16(torch, B, dtype, n, m.cols)
    v = oracle.OracleVBR(m.rows, m.cols, m.rowptr, m.colidx, m.vals, g, 1)
    ref = oracle.vbr_multiply(v.rows, v.cols, 1, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
        Ct = torch.full((m.rows * n,), 5.0, dtype=torch.float32, device="cuda")
        d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl)
        torch.cuda.synchronize()
        got = Ct.cpu().numpy()
        got = got if cl == sa.COL_MAJOR else np.ascontiguousarray(got.reshape(m.rows, n).T).reshape(-1)
        assert np.array_equal(got, ref), (dtype, rows_per, n, cl)
    d.close()


@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
def test_16bit_tiles_on_real_data_with_tails_accumulate_and_prepared_b(monkeypatch, dtype):
    """reference = float64 on the ROUNDED inputs (products of two 16-bit values are exact in fp32; the sums differ by their order only)"""
    torch = _torch()
    m, order = clustered(50, 48, 8000, 150, 20, seed=77)                      # 20 columns of their own per row: tails of 16 + sparse rows that add
    g = true_grouping(order, 48)
    monkeypatch.setenv("SPARTA_UNION_STRAGGLERS", "0")                        # (by default such a small rest rides in tails of 31: the fp32 test above runs both)
    d = sa.DeviceVBS.from_csr(m, g, 32, device=0, dtype=dtype)
    ui, si = d.union_info(), d.sparse_info()
    assert ui["tiles64"] == 50 and ui["tail_nnz"] > 0 and si["nnz"] > 0, (ui, si)
    n = 200
    rng = np.random.default_rng(9)
    B = rng.uniform(-1, 1, m.cols * n).astype(np.float32)
    mr = sa.CSR(m.rows, m.cols, m.rowptr, m.colidx, _round16(m.vals, dtype))
    want, bound = _want(mr, g, _round16(B, dtype), n)
    Bt, ldb = _b16(torch, B, dtype, n, m.cols)
    C0 = rng.uniform(-1, 1, m.rows * n).astype(np.float32)
    for acc in (False, True):
        Ct = torch.from_numpy(C0.copy()).cuda()
        d.spmm(Bt, Ct, n, ldb=ldb, accumulate=acc)
        torch.cuda.synchronize()
        w_, b_ = (want + C0, bound + np.abs(C0)) if acc else (want, bound)
        assert np.max(np.abs(Ct.cpu().numpy() - w_) / (b_ + 1e-30)) < TOL, (dtype, acc)
    C1, C2 = torch.zeros(m.rows * n, device="cuda"), torch.full((m.rows * n,), 3.0, device="cuda")
    d.spmm(Bt, C1, n, ldb=ldb)
    Bp = d.prepare_b(Bt, n, ldb=ldb)
    d.spmm_prepared(Bp, C2)
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)
    Bp.close(); d.close()
