"""The resident-column product (sparta_amd/csrc/k_colres.hip): small all-sparse fp32 matrices times a column-major B -- the reference's real matrices at its operand
widths (/root/reference/src/scripts/run_multiplication_experiments_fixed_cluster.sh:6-7).  Through the C-ABI (sparta_vbs_create_from_csr / sparta_vbs_spmm) against
float64 within 1e-5 * sum|a||b| (the tolerance of the MFMA kernels' tests), bit-identical across the columns-per-workgroup variants and across runs, and against
the row gather (k_sparse.hip) it replaces on these shapes."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import sparta_amd as sa

pytestmark = pytest.mark.gpu
TOL = 1e-5
HERE = os.path.dirname(os.path.abspath(__file__))


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _matrix(rows, cols, density, hubs, seed, empty_every=0):
    rng = np.random.default_rng(seed)
    A = sp.random(rows, cols, density, format="lil", random_state=seed, dtype=np.float32)
    for h in range(hubs):
        k = int(rng.integers(cols // 3, cols))
        A[(h * 37) % rows, rng.choice(cols, k, replace=False)] = rng.uniform(-1, 1, k).astype(np.float32)
    if empty_every:
        for i in range(0, rows, empty_every):
            A[i, :] = 0
    A = A.tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    return A


def _handle(A, grouping=None, w=64):
    m = sa.CSR.from_scipy(A)
    g = np.arange(A.shape[0], dtype=np.int64) // 4 if grouping is None else grouping
    d = sa.DeviceVBS.from_csr(m, g, w, device=0)
    return d, np.asarray(sa.get_permutation(g), np.int64)


def _want(A, perm, B, n, C0=None):
    """float64 product in the handle's row order (row r of C = row perm[r] of A . B), column-major flat, and its absolute bound"""
    Bm = B.reshape(n, A.shape[1]).T.astype(np.float64)
    Ap = A[perm].astype(np.float64)
    C = Ap @ Bm
    bound = abs(Ap) @ np.abs(Bm)
    if C0 is not None:
        C0m = C0.reshape(n, A.shape[0]).T.astype(np.float64)
        C, bound = C + C0m, bound + np.abs(C0m)
    return np.asarray(C).T.reshape(-1), np.asarray(bound).T.reshape(-1)


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


SHAPES = [(1000, 900, 0.01, 0, 0), (5000, 5003, 0.002, 3, 7), (300, 39000, 0.001, 2, 0), (64, 64, 0.05, 0, 0), (63, 5, 0.6, 0, 2), (12001, 11950, 0.0006, 1, 0),
          (20500, 20500, 0.0003, 1, 0)]


@pytest.mark.parametrize("rows,cols,density,hubs,empty", SHAPES)
def test_resident_column_product_against_float64(monkeypatch, rows, cols, density, hubs, empty):
    torch = _torch()
    A = _matrix(rows, cols, density, hubs, rows + cols, empty)
    d, perm = _handle(A)
    info = d.colres_info()
    assert info["slices"] > 0 and info["nnz"] == A.nnz, "this handle should have a resident-column image"
    fits = 40960 // max(info["plane"], (cols + 1) // 2 * 2)
    for n in (1, 3, 4, 5, 130):
        B = sa.gen.dense_rhs(cols, n, seed=n)
        want, bound = _want(A, perm, B, n)
        got = _product(torch, d, B, n)
        assert 1 <= d.colres_info()["nc"] <= min(4, fits, n), "the product did not take the resident-column kernel"
        assert np.all(np.abs(got - want) <= TOL * bound + 1e-30), (n, "resident-column product")
        first = got
        for nc in (1, 2, 3, 4):                                             # a column's arithmetic does not depend on how many columns share the workgroup
            monkeypatch.setenv("SPARTA_COLRES_NC", str(nc))
            got = _product(torch, d, B, n)
            assert d.colres_info()["nc"] == min(nc, fits, n)
            assert np.array_equal(got, first), (n, nc, "columns per workgroup changed the bits")
        monkeypatch.delenv("SPARTA_COLRES_NC")
    # leading dimensions with padding (scalar stores: ldc % 4 != 0), accumulate
    n = 37
    B, C0 = sa.gen.dense_rhs(cols, n, seed=50), sa.gen.dense_rhs(rows, n, seed=51)
    want, bound = _want(A, perm, B, n, C0)
    for ldb, ldc in ((cols + 3, rows + 1), (cols, (rows + 3) // 4 * 4 + 4)):
        got = _product(torch, d, B, n, ldb=ldb, ldc=ldc, C0=C0)
        assert d.colres_info()["nc"] > 0
        assert np.all(np.abs(got - want) <= TOL * bound + 1e-30), (ldb, ldc, "accumulate, padded leading dimensions")
    # host pointers (the reference back-ends' contract: C += A B)
    Ch = C0.copy()
    d.spmm_host(B, n, Ch, accumulate=True)
    assert np.all(np.abs(Ch - want) <= TOL * bound + 1e-30), "host pointers"
    # a row-major C takes the row gather: same product
    Bt = torch.from_numpy(B).cuda()
    Ct = torch.zeros(rows * n, dtype=torch.float32, device="cuda")
    d.spmm(Bt, Ct, n, c_layout=sa.ROW_MAJOR)
    torch.cuda.synchronize()
    assert d.colres_info()["nc"] == 0
    want0, bound0 = _want(A, perm, B, n)
    got = np.ascontiguousarray(Ct.cpu().numpy().reshape(rows, n).T).reshape(-1)
    assert np.all(np.abs(got - want0) <= TOL * bound0 + 1e-30)
    d.close()


def test_resident_column_product_under_a_clustering_and_against_the_row_gather(monkeypatch):
    """a real permutation of the rows (blocking_algo 7), rows of C through crow; the same handle built with SPARTA_COLRES=0 multiplies by the row gather:
    both within the tolerance of float64, and the resident-column bits the same on every run"""
    torch = _torch()
    A = _matrix(9000, 9000, 0.0015, 0, 77).tolil()
    rng = np.random.default_rng(78)
    for h in range(4):                                                        # long rows, but thin in every 64-wide block (a well-filled block would be a tile)
        A[h * 37, rng.choice(9000, 500, replace=False)] = rng.uniform(-1, 1, 500).astype(np.float32)
    A = A.tocsr()
    A.sort_indices()
    m = sa.CSR.from_scipy(A)
    g = sa.BlockingEngine(blocking_algo=7, tau=0.4, col_block_size=64).GetGrouping(m)
    d, perm = _handle(A, g)
    assert not np.array_equal(perm, np.arange(9000)) and d.colres_info()["slices"] > 0 and d.colres_info()["long_rows"] >= 4
    n = 256
    B = sa.gen.dense_rhs(9000, n, seed=5)
    want, bound = _want(A, perm, B, n)
    got = _product(torch, d, B, n)
    assert d.colres_info()["nc"] >= 1
    assert np.all(np.abs(got - want) <= TOL * bound + 1e-30)
    for _ in range(3):
        assert np.array_equal(_product(torch, d, B, n), got), "not bit-reproducible"
    monkeypatch.setenv("SPARTA_COLRES", "0")
    d0, _ = _handle(A, g)
    assert d0.colres_info()["slices"] == 0
    got0 = _product(torch, d0, B, n)
    assert d0.colres_info()["nc"] == 0
    assert np.all(np.abs(got0 - want) <= TOL * bound + 1e-30)
    d.close(); d0.close()


def test_unit_image_of_a_pattern_matrix(monkeypatch):
    """every value 1.0f (the reference's -P 1): the image holds 16-bit columns only; same bits as the image with the value array"""
    torch = _torch()
    A = _matrix(6000, 5000, 0.002, 3, 31, empty_every=11)
    A.data[:] = 1.0
    d, perm = _handle(A)
    assert d.colres_info()["unit"] == 1
    n = 96
    B = sa.gen.dense_rhs(5000, n, seed=9)
    want, bound = _want(A, perm, B, n)
    got = _product(torch, d, B, n, ldb=5000 + 4, ldc=6000)                 # (aligned columns: the 16-byte loads of B)
    assert np.all(np.abs(got - want) <= TOL * bound + 1e-30)
    monkeypatch.setenv("SPARTA_COLRES_UNIT", "0")
    d2, _ = _handle(A)
    assert d2.colres_info()["unit"] == 0 and d2.colres_info()["slices"] > 0
    assert np.array_equal(_product(torch, d2, B, n), got)
    d.close(); d2.close()


@pytest.mark.parametrize("rng_cols,plane_cells", [(2048, 0), (0, 2048), (1500, 2500)])
def test_parts_of_rows_and_k_ranges_give_the_bits_of_the_whole_image(monkeypatch, rng_cols, plane_cells):
    """columns of B / of C that do not fit LDS whole: K ranges walked one after the other with the sums in registers, parts of the rows of C on grid y.  Forced on a small
    matrix (SPARTA_COLRES_RANGE / SPARTA_COLRES_PLANE): the same additions in the same order as the uncut image -- the same bits -- and float64 within the tolerance"""
    torch = _torch()
    A = _matrix(6100, 5900, 0.003, 3, 51, empty_every=13)
    d0, perm = _handle(A)
    assert (d0.colres_info()["parts"], d0.colres_info()["ranges"]) == (1, 1)
    monkeypatch.setenv("SPARTA_COLRES_CUTS", "1")
    if rng_cols:
        monkeypatch.setenv("SPARTA_COLRES_RANGE", str(rng_cols))
    if plane_cells:
        monkeypatch.setenv("SPARTA_COLRES_PLANE", str(plane_cells))
    d, _ = _handle(A)
    info = d.colres_info()
    assert info["ranges"] == (-(-5900 // rng_cols) if rng_cols else 1) and info["parts"] >= (2 if plane_cells else 1)
    for n, ldb, ldc in ((7, 5900, 6100), (64, 5904, 6104), (130, 5901, 6101)):
        B = sa.gen.dense_rhs(5900, n, seed=n)
        C0 = sa.gen.dense_rhs(6100, n, seed=n + 1)
        want, bound = _want(A, perm, B, n, C0)
        got = _product(torch, d, B, n, ldb=ldb, ldc=ldc, C0=C0)
        assert d.colres_info()["nc"] >= 1
        assert np.all(np.abs(got - want) <= TOL * bound + 1e-30)
        assert np.array_equal(got, _product(torch, d0, B, n, ldb=ldb, ldc=ldc, C0=C0)), "the cut image adds in another order"
    d.close(); d0.close()


@pytest.mark.parametrize("name", ["social_location.el", "ia-wikiquote-user-edits-nodup.el"])
def test_the_two_larger_real_matrices_in_parts_and_ranges(monkeypatch, name):
    """58 k x 58 k and 21.6 k x 94 k: two parts x two ranges / one part x three ranges, one launch; the reference's fixed-grid arm against float64 (on request only,
    SPARTA_COLRES_CUTS=1: measured slower than the row gather on exactly these two)"""
    torch = _torch()
    monkeypatch.setenv("SPARTA_COLRES_CUTS", "1")
    m0 = sa.CSR.read_from_edgelist(os.path.join(HERE, "golden", "ref_data", "minitest", name), pattern_only=True)
    r = np.repeat(np.arange(m0.rows), np.diff(m0.rowptr))
    key = np.unique(r.astype(np.int64) * m0.cols + m0.colidx)
    rp = np.concatenate([[0], np.cumsum(np.bincount(key // m0.cols, minlength=m0.rows))])
    m = sa.CSR(m0.rows, m0.cols, rp, (key % m0.cols).astype(np.int32), None)
    g = np.arange(m.rows, dtype=np.int64) // 64
    d = sa.DeviceVBS.from_csr(m, g, 64, 64, True, device=0)
    info = d.colres_info()
    assert info["slices"] > 0 and info["parts"] * info["ranges"] > 1 and info["unit"] == 1
    A = sp.csr_matrix((np.ones(len(key), np.float32), m.colidx, m.rowptr), shape=(m.rows, d.cols))
    perm = np.asarray(sa.get_permutation(g), np.int64)
    n = 256
    B = sa.gen.dense_rhs(d.cols, n, seed=8)
    want, bound = _want(A, perm, B, n)
    got = _product(torch, d, B, n).reshape(n, d.rows)
    assert d.colres_info()["nc"] >= 1
    assert np.all(got[:, m.rows:] == 0.0)
    got = np.ascontiguousarray(got[:, :m.rows]).reshape(-1)
    assert np.all(np.abs(got - want) <= TOL * bound + 1e-30)
    d.close()


def test_few_column_sets_take_the_four_part_image(monkeypatch):
    """N <= 128: the product is fewer workgroups than CUs and one workgroup's stream of A is its time -- the handle keeps a second image in four parts of the rows of C
    (4 x the workgroups, a quarter of the stream each).  Same slots, same order of additions: the bits of the whole image; wide products keep the whole image"""
    torch = _torch()
    A = _matrix(9000, 8000, 0.002, 3, 71, empty_every=17)
    d, perm = _handle(A)
    assert d.colres_info()["small_parts"] == 4
    for n in (1, 2, 64, 100, 128):
        B = sa.gen.dense_rhs(8000, n, seed=n)
        C0 = sa.gen.dense_rhs(9000, n, seed=n + 1)
        want, bound = _want(A, perm, B, n, C0)
        got = _product(torch, d, B, n, C0=C0)
        assert d.colres_info()["used_small"] == 1 and d.colres_info()["nc"] >= 1, n
        assert np.all(np.abs(got - want) <= TOL * bound + 1e-30)
        monkeypatch.setenv("SPARTA_COLRES_SMALL", "0")
        whole = _product(torch, d, B, n, C0=C0)
        assert d.colres_info()["used_small"] == 0
        monkeypatch.delenv("SPARTA_COLRES_SMALL")
        assert np.array_equal(got, whole), (n, "the four-part image adds in another order")
    B = sa.gen.dense_rhs(8000, 1024, seed=5)
    _product(torch, d, B, 1024)
    assert d.colres_info()["used_small"] == 0
    d.close()
    d1, _ = _handle(_matrix(1000, 900, 0.01, 0, 1))                           # a matrix this small keeps one image
    assert d1.colres_info()["small_parts"] == 0
    d1.close()


def test_prepared_b_is_read_where_it_lies():
    """sparta_vbs_prepare_b makes no row-major copy for a handle the resident-column kernel carries; the prepared product is the plain one"""
    import ctypes as C
    from sparta_amd._lib import lib, check
    torch = _torch()
    A = _matrix(4000, 4100, 0.003, 1, 9)
    d, perm = _handle(A)
    n = 64
    B = sa.gen.dense_rhs(4100, n, seed=6)
    plain = _product(torch, d, B, n)
    Bt = torch.from_numpy(B).cuda()
    Ct = torch.zeros(4000 * n, dtype=torch.float32, device="cuda")
    free0 = torch.cuda.mem_get_info()[0]
    bp = C.c_void_p(None)
    check(lib.sparta_vbs_prepare_b(d.h, C.c_void_p(Bt.data_ptr()), 4100, 0, 0, n, None, C.byref(bp)))
    assert torch.cuda.mem_get_info()[0] == free0, "prepare_b allocated a copy the products never read"
    check(lib.sparta_vbs_spmm_prepared(d.h, bp, C.c_void_p(Ct.data_ptr()), 4000, sa.COL_MAJOR, 0, None, None))
    torch.cuda.synchronize()
    assert d.colres_info()["nc"] >= 1
    assert np.array_equal(Ct.cpu().numpy(), plain)
    check(lib.sparta_b_destroy(bp))
    d.close()


def test_handles_with_tiles_take_the_kernel_for_their_sparse_rows(monkeypatch):
    """a handle with MFMA tiles: the tile launches come first, the resident-column kernel then stores the rows of the all-sparse block-rows, ADDS the sparse part of the mixed
    block-rows and leaves the rows of pure tile block-rows alone.  Against float64, against the same handle on the row gather, with and without accumulate; a handle with only
    a handful of sparse rows gets no image"""
    torch = _torch()
    monkeypatch.setenv("SPARTA_LAUNCH_NNZ", "0")                              # (the library sends a SMALL matrix to the sparse-row kernels altogether: this test wants its tiles ...
    monkeypatch.setenv("SPARTA_SPARSE_MIN_STEPS", "0")                        # ... and leaves a handful of nearly empty block-rows with the tiles: this test wants them as sparse rows)
    rng = np.random.default_rng(1)
    dense = np.zeros((1024, 4096), np.float32)
    dense[:256, :640] = rng.uniform(-1, 1, (256, 640))                        # full blocks: block-rows of tiles only
    dense[256:320, 1024:1088] = rng.uniform(-1, 1, (64, 64))                  # a full block inside block-rows that are otherwise scattered: mixed
    dense[256:, :] += (rng.random((768, 4096)) < 0.002) * rng.uniform(-1, 1, (768, 4096)).astype(np.float32)
    A = sp.csr_matrix(dense)
    g = np.arange(1024, dtype=np.int64) // 32
    d, perm = _handle(A, g, w=32)
    info = d.info()
    assert info["tiles16"] + info["tiles32"] + info["tiles64"] + info["stream_steps"] > 0 and 0 < d.sparse_info()["rows"] < 1024, (info, d.sparse_info())
    assert d.colres_info()["slices"] > 0
    monkeypatch.setenv("SPARTA_COLRES", "0")
    d0, _ = _handle(A, g, w=32)
    monkeypatch.delenv("SPARTA_COLRES")
    assert d0.colres_info()["slices"] == 0
    for n, acc in ((128, False), (128, True), (200, False)):
        B = sa.gen.dense_rhs(4096, n, seed=n)
        C0 = sa.gen.dense_rhs(1024, n, seed=n + 1) if acc else None
        want, bound = _want(A, perm, B, n, C0)
        got = _product(torch, d, B, n, C0=C0)
        assert d.colres_info()["nc"] >= 1
        assert np.all(np.abs(got - want) <= TOL * bound + 1e-30), (n, acc)
        got0 = _product(torch, d0, B, n, C0=C0)
        assert d0.colres_info()["nc"] == 0
        assert np.all(np.abs(got0 - want) <= TOL * bound + 1e-30)
    d.close(); d0.close()
    dense[300:, :] = 0                                                         # 44 sparse rows of 1024: the row gather keeps them
    d1, _ = _handle(sp.csr_matrix(dense), g, w=32)
    assert d1.colres_info()["slices"] == 0
    d1.close()


@pytest.mark.parametrize("name", ["bcsstk18_r.el", "wiki-Vote_r.el", "ca-HepPh_r.el"])
def test_reference_real_matrices_at_the_reference_widths(name):
    """the reference's real inputs at B_COLs = 1024 under its fixed-grid arm (-a 2 -F 1, w = 64): carried by the resident-column kernel, equal to float64"""
    torch = _torch()
    m0 = sa.CSR.read_from_edgelist(os.path.join(HERE, "golden", "ref_data", "minitest", name), pattern_only=True)        # (the reference's -P 1)
    # rows in ascending column order, duplicates dropped (two of the files list a row's entries out of order; sparta_vbs_create_from_csr wants them ascending)
    r = np.repeat(np.arange(m0.rows), np.diff(m0.rowptr))
    key = np.unique(r.astype(np.int64) * m0.cols + m0.colidx)
    rp = np.concatenate([[0], np.cumsum(np.bincount(key // m0.cols, minlength=m0.rows))])
    m = sa.CSR(m0.rows, m0.cols, rp, (key % m0.cols).astype(np.int32), None)
    A = m.to_scipy().astype(np.float32)
    g = np.arange(m.rows, dtype=np.int64) // 64
    d = sa.DeviceVBS.from_csr(m, g, 64, 64, True, device=0)
    assert d.colres_info()["slices"] > 0
    assert m.rows <= d.rows < m.rows + 64 and m.cols <= d.cols < m.cols + 64     # (-F 1 pads rows and columns to whole blocks: zeros)
    A = sp.csr_matrix((A.data, A.indices, A.indptr), shape=(m.rows, d.cols))
    perm = np.asarray(sa.get_permutation(g), np.int64)          # (the reference's unstable sort by group: NOT the identity inside a group)
    n = 1024
    B = sa.gen.dense_rhs(d.cols, n, seed=8)
    want, bound = _want(A, perm, B, n)
    got = _product(torch, d, B, n).reshape(n, d.rows)
    assert d.colres_info()["nc"] >= 1
    assert np.all(got[:, m.rows:] == 0.0), "the rows -F 1 added are not zero"
    got = np.ascontiguousarray(got[:, :m.rows]).reshape(-1)
    bad = np.nonzero(~(np.abs(got - want) <= TOL * bound + 1e-30))[0]
    assert len(bad) == 0, ("%d elements off, first at row %d column %d: got %r want %r" % (len(bad), bad[0] % m.rows, bad[0] // m.rows, got[bad[0]], want[bad[0]]),
                           "rows", sorted(set((bad % m.rows).tolist()))[:10], "columns", sorted(set((bad // m.rows).tolist()))[:10])
    d.close()
