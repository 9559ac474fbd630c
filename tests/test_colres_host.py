"""The resident-column image (k_colres.hip; sparta_amd/csrc/vbs_capi.cpp: build_colres) on the HOST: sparta_colres_host_check builds the image of a CSR matrix and
walks it exactly as the kernel does for one column of B -- slots in slice order, a slot's entries in order, the chunks of a long row added in chunk order.
Checked against float64 and, where no row is cut, bit-for-bit against the oracle's restatement of the reference's CSR::multiply
(/root/reference/src/general/csr.cpp:49-65: the same ascending-column order of additions).  No GPU involved."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from sparta_amd._lib import lib, check

_i64p, _i32p, _f32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_float)


def _walk(A, x, crow=None):
    rows, cols = A.shape
    rp, ci, va = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float32)
    y, info = np.full(rows, 7.0, np.float32), np.zeros(12, np.int64)
    cr = None if crow is None else np.ascontiguousarray(crow, np.int64)
    check(lib.sparta_colres_host_check(rows, cols, rp.ctypes.data_as(_i64p), ci.ctypes.data_as(_i32p), va.ctypes.data_as(_f32p),
                                       None if cr is None else cr.ctypes.data_as(_i64p), x.ctypes.data_as(_f32p), y.ctypes.data_as(_f32p), info.ctypes.data_as(_i64p)))
    keys = ["slices", "entries", "long_rows", "plane", "lmax", "nc", "nnz", "unit", "parts", "ranges"]
    return y, {k: int(info[i]) for i, k in enumerate(keys)}


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


@pytest.mark.parametrize("rows,cols,density,hubs,empty", [(1000, 900, 0.01, 0, 0), (5000, 5000, 0.002, 3, 7), (300, 40000, 0.001, 2, 0), (64, 64, 0.5, 0, 0),
                                                           (63, 5, 0.6, 0, 2), (12000, 12000, 0.0006, 1, 0), (1, 1, 1.0, 0, 0)])
def test_walk_of_the_image_equals_the_product(rows, cols, density, hubs, empty):
    A = _matrix(rows, cols, density, hubs, rows + cols, empty)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(cols).astype(np.float32)
    crow = rng.permutation(rows)
    y, info = _walk(A, x, crow)
    if A.nnz == 0:
        assert info["slices"] == 0
        return
    assert info["slices"] > 0 and info["nnz"] == A.nnz
    ref = A.astype(np.float64) @ x.astype(np.float64)
    bound = np.abs(A).astype(np.float64) @ np.abs(x).astype(np.float64)
    assert np.all(np.abs(y[crow] - ref) <= 1e-5 * bound + 1e-30)
    # layout facts the kernel relies on
    assert info["plane"] % 4 == 0 and info["plane"] >= rows
    assert info["entries"] % 256 == 0 and info["entries"] >= A.nnz and info["unit"] == 0
    longest = int(np.diff(A.indptr).max())
    assert (info["long_rows"] > 0) == (longest > info["lmax"])
    if A.nnz > 5000:
        assert info["entries"] <= 1.9 * A.nnz, "padding of the sorted slices (whole batches of 4 steps) out of proportion"


def test_rows_that_are_not_cut_are_added_in_the_order_of_csr_multiply():
    """no long row: a slot IS a row, its entries ascending in column -- the additions of the reference's CSR::multiply (csr.cpp:49-65), fused: bit-identical
    to a float32 FMA chain in that order"""
    A = _matrix(3000, 2500, 0.004, 0, 11)
    assert int(np.diff(A.indptr).max()) <= 32
    x = np.random.default_rng(4).standard_normal(2500).astype(np.float32)
    y, info = _walk(A, x)
    assert info["long_rows"] == 0
    want = np.zeros(3000, np.float32)
    for i in range(3000):
        acc = np.float64(0.0)
        for k in range(A.indptr[i], A.indptr[i + 1]):                      # fma(a, b, acc) in float32 = round(a * b + acc) with the product exact in float64
            acc = np.float64(np.float32(np.float64(A.data[k]) * np.float64(x[A.indices[k]]) + acc))
        want[i] = acc
    assert np.array_equal(y, want)


def test_a_long_row_is_cut_and_its_chunks_added_in_order(monkeypatch):
    monkeypatch.setenv("SPARTA_COLRES_LMAX", "8")
    A = _matrix(200, 300, 0.05, 2, 5)
    x = np.random.default_rng(6).standard_normal(300).astype(np.float32)
    y, info = _walk(A, x)
    assert info["lmax"] == 8 and info["long_rows"] > 0
    want = np.zeros(200, np.float32)
    for i in range(200):
        parts = []
        for o in range(A.indptr[i], A.indptr[i + 1], 8):
            acc = np.float64(0.0)
            for k in range(o, min(o + 8, A.indptr[i + 1])):
                acc = np.float64(np.float32(np.float64(A.data[k]) * np.float64(x[A.indices[k]]) + acc))
            parts.append(np.float32(acc))
        s = np.float32(parts[0]) if parts else np.float32(0)
        for q in parts[1:]:
            s = np.float32(s + q)
        want[i] = s
    assert np.array_equal(y, want)


def test_no_image_for_matrices_the_kernel_cannot_hold(monkeypatch):
    x = np.ones(100, np.float32)
    y, info = _walk(_matrix(41000, 100, 0.01, 0, 1), x)                   # a column of C does not fit LDS whole
    assert info["slices"] == 0 and np.all(y == 7.0)
    y, info = _walk(_matrix(100, 41000, 0.01, 0, 1), np.ones(41000, np.float32))      # nor does a column of B
    assert info["slices"] == 0
    monkeypatch.setenv("SPARTA_COLRES_CUTS", "1")                         # on request: parts of the rows of C, K ranges of the columns of A (up to 4 x 4)
    y, info = _walk(_matrix(41000, 100, 0.01, 0, 1), x)
    assert info["slices"] > 0 and info["parts"] == 2 and info["ranges"] == 1
    y, info = _walk(_matrix(100, 41000, 0.01, 0, 1), np.ones(41000, np.float32))
    assert info["slices"] > 0 and info["parts"] == 1 and info["ranges"] == 2
    y, info = _walk(_matrix(170000, 100, 0.002, 0, 1), x)
    assert info["slices"] == 0
    y, info = _walk(_matrix(100, 170000, 0.002, 0, 1), np.ones(170000, np.float32))
    assert info["slices"] == 0
    monkeypatch.delenv("SPARTA_COLRES_CUTS")
    monkeypatch.setenv("SPARTA_COLRES", "0")
    y, info = _walk(_matrix(100, 100, 0.1, 0, 1), x)
    assert info["slices"] == 0
    monkeypatch.delenv("SPARTA_COLRES")
    y, info = _walk(_matrix(100, 100, 0.1, 0, 1), x, np.zeros(100, np.int64))      # two sparse rows for one row of C: no image
    assert info["slices"] == 0 and np.all(y == 7.0)
    few = np.full(100, -1, np.int64)                                        # a handful of sparse rows among rows of tiles: not worth an image
    few[:20] = np.arange(20)
    A = _matrix(100, 100, 0.1, 0, 1).tolil()
    A[20:, :] = 0
    y, info = _walk(A.tocsr(), x, few)
    assert info["slices"] == 0


def test_sparse_rows_among_tile_rows_and_mixed_rows():
    """a handle with MFMA tiles: some rows of C are not this kernel's (crow -1: left alone), the sparse part of a mixed block-row ADDS (crow + 2^31) to what the tiles stored"""
    rng = np.random.default_rng(12)
    A = _matrix(3000, 2600, 0.004, 2, 61).tolil()
    tile_rows = rng.choice(3000, 900, replace=False)
    A[tile_rows, :] = 0                                                      # those rows of the CSR are not sparse rows: their rows of C belong to tiles
    A = A.tocsr()
    perm = rng.permutation(3000).astype(np.int64)
    crow = perm.copy()
    add_rows = np.setdiff1d(np.arange(3000), tile_rows)[::5]
    crow[add_rows] += 1 << 31
    crow[tile_rows] = -1
    x = rng.standard_normal(2600).astype(np.float32)
    y, info = _walk(A, x, crow)                                              # y arrives filled with 7.0
    assert info["slices"] > 0
    ref = A.astype(np.float64) @ x.astype(np.float64)
    bound = abs(A).astype(np.float64) @ np.abs(x).astype(np.float64)
    is_tile = np.zeros(3000, bool); is_tile[tile_rows] = True
    is_add = np.zeros(3000, bool); is_add[add_rows] = True
    assert np.all(y[perm[is_tile]] == 7.0), "a row of tiles was written"
    store = ~is_tile & ~is_add
    assert np.all(np.abs(y[perm[store]] - ref[store]) <= 1e-5 * bound[store] + 1e-30)
    assert np.all(np.abs(y[perm[is_add]] - (7.0 + ref[is_add])) <= 1e-5 * (bound[is_add] + 7.0) + 1e-30)


def test_pattern_matrices_get_a_unit_image(monkeypatch):
    """every value 1.0f (the reference's -P 1 runs): columns only, the sums are sums of elements of x in ascending column order; padding reads the zero cell"""
    A = _matrix(3000, 2500, 0.004, 2, 21, empty_every=9)
    A.data[:] = 1.0
    x = np.random.default_rng(5).standard_normal(2500).astype(np.float32)
    y, info = _walk(A, x)
    assert info["unit"] == 1 and info["slices"] > 0
    ref = A.astype(np.float64) @ x.astype(np.float64)
    bound = np.abs(A).astype(np.float64) @ np.abs(x).astype(np.float64)
    assert np.all(np.abs(y - ref) <= 1e-5 * bound + 1e-30)
    assert np.all(y[np.diff(A.indptr) == 0] == 0.0)
    monkeypatch.setenv("SPARTA_COLRES_UNIT", "0")
    y2, info2 = _walk(A, x)
    assert info2["unit"] == 0 and np.array_equal(y2, y)               # fma(1, b, acc) = acc + b
    # a matrix with infinities in x: a padding entry must not turn them into NaN in OTHER rows (it reads the zero cell, not a column of its row)
    x2 = x.copy()
    x2[7] = np.inf
    y3, _ = _walk(A, x2)
    touched = np.asarray((A[:, 7] != 0).todense()).reshape(-1)
    assert np.all(np.isfinite(y3[~touched])) and np.all(np.isinf(y3[touched]))


@pytest.mark.parametrize("rng_cols,plane_cells", [(512, 0), (0, 1024), (700, 900), (512, 1000)])
def test_parts_of_rows_and_k_ranges(monkeypatch, rng_cols, plane_cells):
    """a matrix whose columns of B / of C do not fit LDS whole: the columns of A in K ranges (a slot's sum carries over from range to range), the rows of C in parts;
    SPARTA_COLRES_RANGE / SPARTA_COLRES_PLANE force the cuts on a small matrix.  Same sums in the same order as the uncut image: the same bits."""
    A = _matrix(2100, 1900, 0.01, 3, 41, empty_every=13)
    x = np.random.default_rng(8).standard_normal(1900).astype(np.float32)
    crow = np.random.default_rng(9).permutation(2100)
    y0, info0 = _walk(A, x, crow)
    assert info0["parts"] == 1 and info0["ranges"] == 1
    monkeypatch.setenv("SPARTA_COLRES_CUTS", "1")
    if rng_cols:
        monkeypatch.setenv("SPARTA_COLRES_RANGE", str(rng_cols))
    if plane_cells:
        monkeypatch.setenv("SPARTA_COLRES_PLANE", str(plane_cells))
    y, info = _walk(A, x, crow)
    assert info["slices"] > 0
    assert info["ranges"] == (-(-1900 // rng_cols) if rng_cols else 1)
    assert info["parts"] >= (2 if plane_cells else 1)
    assert np.array_equal(y, y0)


def test_the_two_larger_real_matrices_get_an_image_in_parts_and_ranges(monkeypatch):
    """(on request only: measured slower than the row gather on exactly these two -- vbs_capi.cpp, build_colres)"""
    import os
    monkeypatch.setenv("SPARTA_COLRES_CUTS", "1")
    import sparta_amd as sa
    here = os.path.dirname(os.path.abspath(__file__))
    for name, parts, ranges in (("social_location.el", 2, 2), ("ia-wikiquote-user-edits-nodup.el", 1, 3)):
        m0 = sa.CSR.read_from_edgelist(os.path.join(here, "golden", "ref_data", "minitest", name), pattern_only=True)
        r = np.repeat(np.arange(m0.rows), np.diff(m0.rowptr))
        key = np.unique(r.astype(np.int64) * m0.cols + m0.colidx)
        A = sp.csr_matrix((np.ones(len(key), np.float32), (key % m0.cols).astype(np.int32), np.concatenate([[0], np.cumsum(np.bincount(key // m0.cols, minlength=m0.rows))])),
                          shape=(m0.rows, m0.cols))
        x = np.random.default_rng(3).standard_normal(m0.cols).astype(np.float32)
        y, info = _walk(A, x)
        assert (info["parts"], info["ranges"], info["unit"]) == (parts, ranges, 1), info
        ref = A.astype(np.float64) @ x.astype(np.float64)
        bound = abs(A).astype(np.float64) @ np.abs(x).astype(np.float64)
        assert np.all(np.abs(y - ref) <= 1e-5 * bound + 1e-30)


def test_the_four_part_image_for_few_column_sets(monkeypatch):
    """a handle keeps a second image in four parts of the rows of C for products of at most 64 column sets (4 x the workgroups, a quarter of the stream of A each):
    SPARTA_COLRES_FORCE_PARTS=4 walks that image on the host -- the same slots, chunks and order of additions per row: the bits of the whole image"""
    A = _matrix(9000, 8000, 0.002, 3, 71, empty_every=17)
    x = np.random.default_rng(2).standard_normal(8000).astype(np.float32)
    crow = np.random.default_rng(3).permutation(9000)
    y0, info0 = _walk(A, x, crow)
    monkeypatch.setenv("SPARTA_COLRES_FORCE_PARTS", "4")
    y, info = _walk(A, x, crow)
    assert info0["parts"] == 1 and info["parts"] == 4 and info["ranges"] == 1
    assert info["plane"] <= info0["plane"] and info["long_rows"] == info0["long_rows"]
    assert np.array_equal(y, y0)
