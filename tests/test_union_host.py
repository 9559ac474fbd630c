"""The column-compacted ("union-pattern") tiles (k_union.hip; sparta_amd/csrc/vbs_build.cpp mode 3, vbs_union.cpp) on the HOST: sparta_union_host_check builds the hybrid
image of a CSR matrix under a grouping exactly as sparta_vbs_create_from_csr does for an fp32 handle and multiplies it with one column of B, walking the DEVICE form of the
tiles (per-worker step records, list entries, MFMA-fragment-order slices).  Checked against the oracle's restatement of the reference's VBR::multiply
(/root/reference/src/general/vbr.cpp:323-372) on the VBS the reference builds at SMALL block widths (-b 1, 2, 4, 8: vbr.cpp:177-228 -- the union of the touched column
blocks per cluster, which is what the tiles hold), and against float64.  No GPU involved."""
import ctypes as C

import numpy as np
import pytest

import sparta_amd as sa
from sparta_amd._lib import lib, check
from oracle import oracle

_i64p, _i32p, _f32p, _f64p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)


@pytest.fixture(autouse=True)
def small_matrices_keep_their_decisions(monkeypatch):
    # the library keeps matrices this small on ONE kind of launch (vbs_build.cpp: sparse_min_steps, SPARTA_LAUNCH_NNZ); the subject here is the builder behind those rules
    monkeypatch.setenv("SPARTA_SPARSE_MIN_STEPS", "0")
    monkeypatch.setenv("SPARTA_LAUNCH_NNZ", "0")


def clustered(n_groups, rows_per, cols, shared, own, seed, fill=0.8, scatter=True, integer=False):
    """rows of a group share `shared` columns (each present with probability `fill`) and hold `own` columns of their own"""
    rng = np.random.default_rng(seed)
    n = n_groups * rows_per
    order = rng.permutation(n) if scatter else np.arange(n)
    rr, cc = [], []
    for gi in range(n_groups):
        base = rng.choice(cols, shared, replace=False)
        for k in range(rows_per):
            c = np.union1d(base[rng.random(shared) < fill], rng.choice(cols, own, replace=False))
            rr.append(np.full(len(c), order[gi * rows_per + k]))
            cc.append(c)
    r, c = np.concatenate(rr), np.concatenate(cc)
    o = np.lexsort((c, r))
    r, c = r[o], c[o]
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int64)
    v = rng.integers(1, 4, len(c)).astype(np.float32) if integer else rng.uniform(-1, 1, len(c)).astype(np.float32)
    return sa.CSR(n, cols, rowptr, c.astype(np.int32), v), order


def true_grouping(order, rows_per):
    """the grouping a perfect clustering would return: group id = smallest row of the group (the reference's convention: blocking.cpp:207)"""
    n = len(order)
    g = np.empty(n, np.int64)
    for gi in range(n // rows_per):
        rows = order[gi * rows_per:(gi + 1) * rows_per]
        g[rows] = rows.min()
    return g


def walk(m, g, w, x, rbs=0, ff=False, workers=7):
    rows_pad = m.rows if not ff else ((m.rows - 1) // rbs + 1) * rbs
    y, info = np.zeros(rows_pad, np.float64), np.zeros(14, np.int64)
    vals = None if m.vals is None else m.vals.ctypes.data_as(_f32p)
    gg = np.ascontiguousarray(g, np.int64)
    check(lib.sparta_union_host_check(m.rows, m.cols, m.rowptr.ctypes.data_as(_i64p), m.colidx.ctypes.data_as(_i32p), vals, gg.ctypes.data_as(_i64p), w, rbs, int(ff),
                                      workers, x.ctypes.data_as(_f32p), y.ctypes.data_as(_f64p), info.ctypes.data_as(_i64p)))
    keys = ["tiles32", "tiles64", "steps32", "steps64", "area", "list_entries", "nnz", "sparse_nnz", "tile_area", "rows", "workers32", "workers64", "tail_nnz", "tile_rows"]
    return y, {k: int(info[i]) for i, k in enumerate(keys)}


def reference_product(m, g, w, x, rbs=0, ff=False):
    """the reference's own path on the same grouping: VBR::fill_from_CSR_inplace + VBR::multiply, as restated by the oracle"""
    v = oracle.OracleVBR(m.rows, m.cols, m.rowptr, m.colidx, m.vals, g, w, rbs, ff)
    xp = np.zeros(v.cols, np.float32)
    xp[:m.cols] = x
    return oracle.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, xp, 1)


@pytest.mark.parametrize("w", [1, 2, 4, 8])
@pytest.mark.parametrize("rows_per", [5, 20, 48, 70])
def test_clusters_at_small_block_widths_match_the_reference_product(w, rows_per):
    m, order = clustered(12, rows_per, 3000, 90, 3, seed=rows_per + w, integer=True)
    g = true_grouping(order, rows_per)
    x = np.random.default_rng(1).integers(-3, 4, m.cols).astype(np.float32)
    y, info = walk(m, g, w, x)
    assert info["tiles32"] + info["tiles64"] > 0 and info["nnz"] > 0.7 * m.nztot(), info           # the clusters ARE tiles
    ref = np.asarray(reference_product(m, g, w, x)).reshape(-1)
    assert np.array_equal(y[:m.rows].astype(np.float32), ref[:m.rows])                             # small integers: every order of additions gives the same bits
    assert info["nnz"] + info["sparse_nnz"] == m.nztot() and info["tile_area"] == 0


def test_every_nonzero_is_somewhere_exactly_once_on_a_mixed_matrix():
    # clusters of several heights + uniform noise rows + a dense band: w-wide tiles, column-compacted tiles and sparse rows in one image
    rng = np.random.default_rng(3)
    m1, order = clustered(30, 40, 4096, 120, 4, seed=11, scatter=False)
    import scipy.sparse as sp
    A = sp.csr_matrix((m1.vals, m1.colidx, m1.rowptr), shape=(m1.rows, m1.cols)).tolil()
    A[1000:1040, :] = 0
    A[1000:1040, 512:1536] = rng.uniform(-1, 1, (40, 1024)).astype(np.float32)                       # a dense cluster: stays w-wide tiles (a list of 1024 columns costs more than 32 blocks)
    A = A.tocsr(); A.sort_indices()
    m = sa.CSR(A.shape[0], A.shape[1], A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float32))
    g = np.arange(m.rows) // 40 * 40
    x = rng.uniform(-1, 1, m.cols).astype(np.float32)
    for workers in (1, 3, 64):
        y, info = walk(m, g, 32, x, workers=workers)
        perm = np.asarray(sa.get_permutation(g))                                                 # y is in reordered order: row r of y = original row perm[r]
        want = (A.astype(np.float64) @ x.astype(np.float64))[perm]
        scale = (abs(A).astype(np.float64) @ np.abs(x).astype(np.float64))[perm] + 1e-30
        assert np.max(np.abs(y - want) / scale) < 1e-12, info
        assert info["tiles64"] > 0 and info["tile_area"] > 0 and info["tail_nnz"] > 0 and info["nnz"] + info["sparse_nnz"] <= m.nztot()


def test_tails_hold_a_rows_own_columns_and_overflow_into_sparse_rows(monkeypatch):
    # 20 columns of their own per row: 16 ride in the tile's tail, the others are sparse-row entries that add ...
    m, order = clustered(10, 48, 4000, 100, 20, seed=21, integer=True)
    g = true_grouping(order, 48)
    x = np.random.default_rng(3).integers(-3, 4, m.cols).astype(np.float32)
    ref = np.asarray(reference_product(m, g, 1, x)).reshape(-1)
    monkeypatch.setenv("SPARTA_UNION_STRAGGLERS", "0")
    y, info = walk(m, g, 1, x)
    assert info["tail_nnz"] >= 16 * m.rows * 0.9 and info["sparse_nnz"] > 0 and info["nnz"] + info["sparse_nnz"] == m.nztot(), info
    assert np.array_equal(y.astype(np.float32), ref)
    capped = info
    # ... unless that overflow is all the sparse-row kernels would be launched for: then the rows get the longest tail a step record can name (31) and nothing is left
    monkeypatch.delenv("SPARTA_UNION_STRAGGLERS")
    y, info = walk(m, g, 1, x)
    assert info["sparse_nnz"] == 0 and info["nnz"] == m.nztot() and info["tail_nnz"] == capped["tail_nnz"] + capped["sparse_nnz"], info
    assert np.array_equal(y.astype(np.float32), ref)


def test_stragglers_become_tiles_when_little_else_is_left_for_the_sparse_rows(monkeypatch):
    # 30 clusters + 5 rows outside every cluster (groups of their own): with tiles in the handle anyway, those rows become one-row tiles with every column in the list
    m, order = clustered(30, 48, 6000, 100, 3, seed=41, integer=True)
    g = true_grouping(order, 48)
    loners = order[:5 * 48:48]                                     # one row of each of the first five clusters, torn out of its cluster: a group of its own
    g[loners] = m.rows + np.arange(5)                              # (ids no other row has; ids only order the groups)
    x = np.random.default_rng(5).integers(-3, 4, m.cols).astype(np.float32)
    ref = np.asarray(reference_product(m, g, 1, x)).reshape(-1)
    y, info = walk(m, g, 1, x)
    assert info["sparse_nnz"] == 0 and info["nnz"] == m.nztot() and info["tiles32"] >= 5, info
    assert np.array_equal(y.astype(np.float32), ref)
    monkeypatch.setenv("SPARTA_UNION_STRAGGLERS", "0")             # switched off: they are sparse rows again
    y, info = walk(m, g, 1, x)
    assert info["sparse_nnz"] > 0 and np.array_equal(y.astype(np.float32), ref), info


def test_tiles_cut_into_row_tile_pieces_give_the_same_product(monkeypatch):
    # whole tiles quantise the makespan: where a CU (here: a worker) would carry a tile more than the others, a few tiles are cut into pieces of one MFMA row tile (16 rows, same list)
    # and dealt again (vbs_union.cpp).  SPARTA_UNION_SPLIT=2 cuts whenever a worker is above the mean; =0 never.  Same product either way, bit for bit.
    m, order = clustered(23, 48, 3000, 90, 3, seed=77, integer=True)
    g = true_grouping(order, 48)
    x = np.random.default_rng(5).integers(-3, 4, m.cols).astype(np.float32)
    ref = np.asarray(reference_product(m, g, 1, x)).reshape(-1)
    monkeypatch.setenv("SPARTA_UNION_SPLIT", "0")
    y0, whole = walk(m, g, 1, x, workers=5)
    monkeypatch.setenv("SPARTA_UNION_SPLIT", "2")
    y2, cut = walk(m, g, 1, x, workers=5)
    assert whole["tiles64"] == cut["tiles64"] == 23 and cut["steps64"] > whole["steps64"] and cut["list_entries"] > whole["list_entries"], (whole, cut)
    assert cut["tile_rows"] == whole["tile_rows"] == m.rows and cut["area"] == whole["area"]
    assert np.array_equal(y0.astype(np.float32), ref) and np.array_equal(y2.astype(np.float32), ref)
    monkeypatch.delenv("SPARTA_UNION_SPLIT")                              # the default: only where the makespan falls by more than 1.5 %
    y1, info = walk(m, g, 1, x, workers=5)
    assert np.array_equal(y1.astype(np.float32), ref) and whole["steps64"] <= info["steps64"] <= cut["steps64"]


def test_tiles_follow_the_parts_rules():
    # 70-row clusters: one part of 64 rows (a 33..64-row tile) + one of 6 (a <= 32-row tile)
    m, order = clustered(8, 70, 2000, 60, 2, seed=5, scatter=False)
    g = np.arange(m.rows) // 70 * 70
    x = np.ones(m.cols, np.float32)
    y, info = walk(m, g, 32, x)
    assert info["tiles64"] == 8 and info["tiles32"] == 8 and info["tile_rows"] == m.rows, info
    m, order = clustered(12, 5, 2000, 60, 2, seed=6, scatter=False)
    g = np.arange(m.rows) // 5 * 5
    y, info = walk(m, g, 32, np.ones(m.cols, np.float32))
    assert info["tiles64"] == 0 and info["tiles32"] == 12, info                                    # every cluster its own tile (clusters share no column: packing them would save nothing)
    assert info["steps32"] >= info["list_entries"] // 32


def test_force_fixed_padding_and_empty_rows():
    m, order = clustered(10, 24, 1500, 50, 2, seed=9)
    g = true_grouping(order, 24)
    x = np.random.default_rng(2).uniform(-1, 1, m.cols).astype(np.float32)
    y, info = walk(m, g, 16, x, rbs=32, ff=True)
    ref = np.asarray(reference_product(m, g, 16, x, 32, True)).reshape(-1)
    assert len(y) == len(ref)
    assert np.allclose(y, ref, rtol=0, atol=1e-4)


def test_switched_off_means_no_tiles(monkeypatch):
    m, order = clustered(6, 40, 1000, 60, 2, seed=4)
    g = true_grouping(order, 40)
    x = np.ones(m.cols, np.float32)
    monkeypatch.setenv("SPARTA_UNION", "0")
    y, info = walk(m, g, 32, x)
    assert info["tiles32"] + info["tiles64"] == 0 and info["nnz"] == 0
