"""GPU (-m gpu): parity of the hand-written HIP SpMM (through the C-ABI) with the oracle and the golden vectors.

Bars: SPARTA_SPMM_EXACT -> bit-identical to the reference's VBR::multiply (golden C);
      SPARTA_SPMM_MFMA  -> |C - C_oracle| <= 1e-5 * sum_k |a||b| per element (fp32; the MFMA sums k in another order
                           and fuses multiply-add) -- the tolerance BASELINE.md section 3 states."""
import numpy as np
import pytest

import sparta_amd as sa
from oracle import oracle as O
import _util as U

pytestmark = pytest.mark.gpu


# the tests that ALSO run with the library's own decisions -- no override of the small-matrix rule (SPARTA_LAUNCH_NNZ), of the few-block-rows rule
# (SPARTA_SPARSE_MIN_STEPS) or of the resident-column kernel (SPARTA_COLRES) -- and say, from the handle's own records, which kernels carried each product
_DEFAULTS_TOO = {"test_ragged_shapes_vs_oracle", "test_accumulate_overwrite_layouts_and_pointer_spaces", "test_special_block_rows",
                 "test_create_from_csr_gives_the_same_product", "test_sparse_row_path_mixed_matrix"}


def pytest_generate_tests(metafunc):
    if "_sparse_row_mode" in metafunc.fixturenames:
        modes = ["mfma-only", "with-sparse-rows"] + (["library-defaults"] if metafunc.function.__name__ in _DEFAULTS_TOO else [])
        metafunc.parametrize("_sparse_row_mode", modes, indirect=True)


@pytest.fixture(autouse=True)
def _sparse_row_mode(request, monkeypatch):
    """every test runs twice: with the sparse-row path switched off (all block-rows on the MFMA kernels) and with the sparse-row path on and the rules that keep
    SMALL matrices on one kind of launch switched off (nearly empty block-rows -- most of the small test matrices -- go to the sparse-row kernels); the tests of
    _DEFAULTS_TOO a third time with nothing overridden: the decisions a caller of the library gets"""
    mode = getattr(request, "param", "with-sparse-rows")
    if mode == "library-defaults":
        for k in ("SPARTA_SPARSE_K", "SPARTA_SPARSE_MIN_STEPS", "SPARTA_LAUNCH_NNZ", "SPARTA_COLRES", "SPARTA_UNION"):
            monkeypatch.delenv(k, raising=False)
        return mode
    if mode == "mfma-only":
        monkeypatch.setenv("SPARTA_SPARSE_K", "0")
    else:
        monkeypatch.delenv("SPARTA_SPARSE_K", raising=False)
        monkeypatch.setenv("SPARTA_SPARSE_MIN_STEPS", "0")        # the library leaves a handful of such block-rows with the tiles: the test matrices are all "a handful"
    # ... and sends a SMALL matrix to the sparse-row kernels altogether when its tiles are not worth their launches (vbs_build.cpp, SPARTA_LAUNCH_NNZ): the test
    # matrices are all small, and these tests are about the tiles (tests/test_real_matrices.py and test_small_matrices_* run with the library's own rule)
    monkeypatch.setenv("SPARTA_LAUNCH_NNZ", "0")
    # ... and multiplies a small all-sparse fp32 matrix by the resident-column kernel (k_colres.hip): these tests are about the tiles and the row gather
    # (tests/test_colres_gpu.py, test_real_matrices.py run that kernel)
    monkeypatch.setenv("SPARTA_COLRES", "0")
    return mode


CARRIED = {}          # library-defaults mode: test id -> which kernels carried its products (printed by the last test of the file)


def _carrier(d, mode, what, reference_layouts=True, fp32=True):
    """which kernels carried the LAST product of handle d, from the handle's own records; asserts that the records agree with the mode"""
    info, sp, cr, ui = d.info(), d.sparse_info(), d.colres_info(), d.union_info()
    tiles = info["tiles16"] + info["tiles32"] + info["tiles64"]
    rec = {"tiles": tiles, "path": {0: "none", 1: "stream", 2: "per-class", 3: "generic"}[info["last_path"]], "sparse_rows": sp["rows"], "sparse_nnz": sp["nnz"],
           "resident_columns": cr["nc"], "union_tiles": ui["tiles32"] + ui["tiles64"]}
    if mode == "mfma-only":
        assert sp["rows"] == 0 and cr["slices"] == 0 and ui["area"] == 0, rec
    elif mode == "with-sparse-rows":
        assert cr["slices"] == 0, rec                                      # SPARTA_COLRES=0: the sparse rows went through the row gather
    else:
        # the resident-column kernel carries the sparse rows of a small fp32 handle exactly when it has an image and the call uses the reference's layouts
        if fp32 and cr["slices"] > 0 and reference_layouts:
            assert cr["nc"] > 0, rec
        else:
            assert cr["nc"] == 0, rec
        if tiles > 0:
            assert info["last_path"] in (1, 2, 3), rec
        CARRIED.setdefault(what, rec)
    return rec


TOL = 1e-5


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU test on a box without a GPU"
    return torch


def _check(C, Co, bound, what=""):
    err = np.abs(C - Co)
    bad = err > TOL * bound + 1e-30
    assert not bad.any(), "%s: %d elements outside tolerance, max err %.3e" % (what, int(bad.sum()), float(err.max()))


def _oracle_c(v, B, n, C_in=None):
    return O.vbr_multiply(v.rows, v.cols, v.block_col_size, v.row_part, v.nzcount, v.jab, v.mab, B, n, C_in)


def test_native_library_is_loaded_and_sees_the_gpu():
    assert sa.device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libsparta_amd.so" in maps


def test_appendix_b_kat_on_gpu():
    k = U.load("kat9.npz")
    v = sa.VBR.from_arrays(9, 9, 3, k["row_part"], k["nzcount"], k["jab"], k["mab"])
    B = np.arange(1, 19, dtype=np.float32)
    for algo in (sa.SPMM_MFMA, sa.SPMM_EXACT):
        C = np.zeros(18, np.float32)
        v.multiply(B, 2, C, algo=algo)            # VBR::multiply(B, B_cols, C): C += A*B, host buffers
        assert C.tolist() == [0, 0, 0, 0, 126, 22, 102, 14, 10, 0, 0, 0, 0, 306, 49, 219, 32, 55]   # small integers: exact either way


@pytest.mark.parametrize("key,name,cfg", U.case_list(), ids=[c[0] for c in U.case_list()])
def test_golden_cases(key, name, cfg):
    m = U.matrices()[name]
    f = U.case_fields(key)
    w, rbs, ff = cfg["w"], cfg.get("rbs", 1), cfg.get("ff", False)
    v = sa.VBR().fill_from_CSR_inplace(m, f["grouping"].astype(np.int64), w, rbs, ff)
    assert U.sha(v.mab) == str(f["mab_sha"])
    n = f["C"].size // v.rows
    B = sa.gen.dense_rhs(v.cols, n, seed=77)
    d = v.to_device(0)
    Cx = np.zeros(v.rows * n, np.float32)
    d.spmm_host(B, n, Cx, accumulate=True, algo=sa.SPMM_EXACT)
    assert np.array_equal(Cx, f["C"]), "exact-order kernel is not bit-identical to the reference's VBR::multiply"
    Cm = np.zeros(v.rows * n, np.float32)
    d.spmm_host(B, n, Cm, accumulate=True, algo=sa.SPMM_MFMA)
    _check(Cm, f["C"], U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n), key)


SHAPES = [
    # rows, cols, nnz, w, blocking, N   -- ragged everything: w not a multiple of 4, cols % w != 0, N % 32 != 0, N > 128
    (257, 391, 5000, 13, ("tau", 0.6), 37),
    (1000, 777, 30000, 48, ("tau", 0.7), 100),
    (1000, 777, 30000, 100, ("fixed", 200), 33),
    (640, 640, 20000, 64, ("fixed", 64), 128),
    (640, 640, 20000, 64, ("fixed", 128), 300),
    (500, 2000, 9000, 200, ("tau", 0.9), 64),
    (333, 65, 3000, 1, ("tau", 0.5), 5),
    (70, 70, 70 * 70 // 2, 16, ("fixed", 70), 129),
    (2048, 2048, 40000, 32, ("keeper", 16), 64),
]


@pytest.mark.parametrize("rows,cols,nnz,w,blk,n", SHAPES)
def test_ragged_shapes_vs_oracle(_sparse_row_mode, rows, cols, nnz, w, blk, n):
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols + w)
    if blk[0] == "tau":
        g = sa.BlockingEngine(tau=blk[1], col_block_size=w).GetGrouping(m)
    elif blk[0] == "keeper":
        g = sa.BlockingEngine(tau=0.5, col_block_size=w, row_block_size=blk[1], blocking_algo=5).GetGrouping(m)
    else:
        g = np.arange(rows) // blk[1]
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=3)
    Co = _oracle_c(v, B, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    d = v.to_device(0)
    for algo in (sa.SPMM_MFMA, sa.SPMM_EXACT):
        C = np.zeros(v.rows * n, np.float32)
        d.spmm_host(B, n, C, accumulate=True, algo=algo)
        if algo == sa.SPMM_EXACT:
            assert np.array_equal(C, Co)
        else:
            _check(C, Co, bound, "mfma")
            _carrier(d, _sparse_row_mode, "ragged %dx%d w%d %s n%d" % (rows, cols, w, blk[0], n))
    # independent check through the permutation: C_vbs[r] == C_csr[perm[r]]  
    perm = sa.get_permutation(g)
    Cc = O.csr_multiply(m.rows, m.rowptr, m.colidx, m.vals, B, m.cols, n).reshape(n, m.rows)
    C = np.zeros(v.rows * n, np.float32)
    d.spmm_host(B, n, C, accumulate=False)
    _check(C.reshape(n, v.rows), Cc[:, perm], bound.reshape(n, v.rows), "vs CSR through perm")


def test_accumulate_overwrite_layouts_and_pointer_spaces(_sparse_row_mode):
    torch = _torch()
    m = sa.gen.uniform_random(900, 700, 25000, seed=5)
    w, n = 40, 72
    g = sa.BlockingEngine(tau=0.6, col_block_size=w).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=8)
    C0 = sa.gen.dense_rhs(v.rows, n, seed=9)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n) + np.abs(C0)
    Co_acc = _oracle_c(v, B, n, C0)            # reference semantics: C += A*B
    Co = _oracle_c(v, B, n)
    d = v.to_device(0)
    C = C0.copy(); d.spmm_host(B, n, C, accumulate=True); _check(C, Co_acc, bound, "host accumulate")
    _carrier(d, _sparse_row_mode, "layouts: host pointers")
    C = C0.copy(); d.spmm_host(B, n, C, accumulate=True, algo=sa.SPMM_EXACT); assert np.array_equal(C, Co_acc)
    C = np.full(v.rows * n, 3.25, np.float32); d.spmm_host(B, n, C, accumulate=False); _check(C, Co, bound, "host overwrite")
    # device pointers, all four layout combinations, both kernels
    Bcm = torch.from_numpy(B).cuda()
    Brm = torch.from_numpy(np.ascontiguousarray(B.reshape(n, v.cols).T).reshape(-1)).cuda()
    for bl, Bt in ((sa.COL_MAJOR, Bcm), (sa.ROW_MAJOR, Brm)):
        for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
            for algo in (sa.SPMM_MFMA, sa.SPMM_EXACT):
                Ct = torch.full((v.rows * n,), -1.0, dtype=torch.float32, device="cuda")
                d.spmm(Bt, Ct, n, accumulate=False, algo=algo, b_layout=bl, c_layout=cl)
                torch.cuda.synchronize()
                got = Ct.cpu().numpy()
                if cl == sa.ROW_MAJOR:
                    got = np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
                if algo == sa.SPMM_EXACT:
                    assert np.array_equal(got, Co), (bl, cl)
                else:
                    _check(got, Co, bound, "layouts %d %d" % (bl, cl))
                    _carrier(d, _sparse_row_mode, "layouts: B %s C %s" % ("col" if bl == sa.COL_MAJOR else "row", "col" if cl == sa.COL_MAJOR else "row"),
                             reference_layouts=(bl == sa.COL_MAJOR and cl == sa.COL_MAJOR))
    # leading dimensions larger than the matrix
    ldb, ldc = v.cols + 5, v.rows + 3
    Bp = np.zeros(ldb * n, np.float32); Bp.reshape(n, ldb)[:, :v.cols] = B.reshape(n, v.cols)
    Bt = torch.from_numpy(Bp).cuda()
    Ct = torch.zeros(ldc * n, dtype=torch.float32, device="cuda")
    dt = d.spmm(Bt, Ct, n, accumulate=False, ldb=ldb, ldc=ldc, timed=True)
    assert dt > 0
    _check(Ct.cpu().numpy().reshape(n, ldc)[:, :v.rows].reshape(-1), Co, bound, "padded ld")
    # a non-default stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        Ct2 = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
        d.spmm(Bcm, Ct2, n, accumulate=False)
    s.synchronize()
    _check(Ct2.cpu().numpy(), Co, bound, "side stream")


def test_special_block_rows(_sparse_row_mode):
    """empty block-rows (zero blocks), 1-row clusters, one huge block-row, an all-empty matrix row range"""
    torch = _torch()
    rows, cols, w, n = 300, 256, 32, 48
    rng = np.random.Generator(np.random.PCG64(3))
    dense = (rng.random((rows, cols)) < 0.08).astype(np.float32) * rng.uniform(-1, 1, (rows, cols)).astype(np.float32)
    dense[10:60] = 0           # 50 empty rows -> cluster into one block-row with nzcount = 0
    dense[200] = 0
    import scipy.sparse as sp
    m = sa.CSR.from_scipy(sp.csr_matrix(dense))
    for eng in (sa.BlockingEngine(tau=0.4, col_block_size=w), sa.BlockingEngine(tau=1.0, col_block_size=w),
                sa.BlockingEngine(tau=0.0, col_block_size=w)):
        g = eng.GetGrouping(m)
        v = sa.VBR().fill_from_CSR_inplace(m, g, w)
        assert (v.nzcount == 0).any() or eng.tau >= 1.0      # tau = 1 merges everything into ONE huge block-row
        B = sa.gen.dense_rhs(v.cols, n, seed=2)
        Co = _oracle_c(v, B, n)
        bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
        d = v.to_device(0)
        C = np.full(v.rows * n, 9.0, np.float32)
        d.spmm_host(B, n, C, accumulate=False)      # overwrite must zero the rows of empty block-rows
        _check(C, Co, bound)
        _carrier(d, _sparse_row_mode, "special block-rows tau %.1f" % eng.tau)
        Cx = np.zeros(v.rows * n, np.float32)
        d.spmm_host(B, n, Cx, accumulate=True, algo=sa.SPMM_EXACT)
        assert np.array_equal(Cx, Co)


def test_non_finite_b_outside_the_matrix_does_not_leak():
    """cols % w != 0: the reference multiplies stored zeros with whatever lies past B's column end (vbr.cpp:351,362).
    We must not read it: put NaNs right after every column of B (ld > cols) and expect a clean result."""
    torch = _torch()
    m = sa.gen.uniform_random(200, 150, 3000, seed=12)
    w, n = 64, 40                    # 150 = 2*64 + 22: last block column is padded
    v = sa.VBR().fill_from_CSR_inplace_fixed(m, 32, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=1)
    ldb = v.cols + 64
    Bp = np.full(ldb * n, np.nan, np.float32)
    Bp.reshape(n, ldb)[:, :v.cols] = B.reshape(n, v.cols)
    Co = _oracle_c(v, B, n)
    d = v.to_device(0)
    for algo in (sa.SPMM_MFMA, sa.SPMM_EXACT):
        Ct = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
        d.spmm(torch.from_numpy(Bp).cuda(), Ct, n, ldb=ldb, algo=algo)
        torch.cuda.synchronize()
        got = Ct.cpu().numpy()
        assert np.isfinite(got).all()
        _check(got, Co, U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n))


@pytest.mark.parametrize("world,w,n,mesh", [(3, 16, 40, (3, 3, 5)), (2, 32, 128, (5, 5, 12)), (4, 32, 256, (4, 4, 9))])
def test_row_block_exchange_on_one_gpu(world, w, n, mesh):
    """the sparsity-aware exchange with every rank played by this process: sparta_pack_blocks fills each rank's send buffer,
    the all-to-all is emulated by device copies in its (source rank, block) delivery order, and the two products
    (own shard || received row-blocks, both through sparta_vbs_spmm_gathered on the row-block-tiled layout) must add up to
    the oracle's product of the slab with the whole B"""
    torch = _torch()
    slabs = [sa.gen.fem3d_slab(mesh[0], mesh[1], mesh[2], r, world, dof=3, pad_to=w, seed=4) for r in range(world)]
    n_pad = slabs[0][2]
    shards = [sa.gen.dense_rhs(n_pad, n, seed=50 + r) for r in range(world)]
    Bfull = sa.dist.gathered_to_colmajor(np.concatenate(shards), world, n_pad, n)
    tiles = [torch.from_numpy(sa.dist.to_block_tiles(s, n_pad, n, w)).cuda() for s in shards]
    vbs, all_need = [], []
    for m, _, _ in slabs:
        g = sa.BlockingEngine(tau=0.4, col_block_size=w).GetGrouping(m)
        vbs.append(sa.VBR().fill_from_CSR_inplace(m, g, w))
        all_need.append(sa.dist.needed_blocks(vbs[-1].jab, w, n_pad, world))
    exs = [sa.dist.RowBlockExchange(vbs[r], r, world, n_pad, n, device=0, all_need=all_need) for r in range(world)]
    tile = w * n
    for r in range(world):                                   # pack (HIP)
        exs[r]._pack(tiles[r])
    for p in range(world):                                   # the all-to-all: chunk q of p's receive buffer <- chunk p of q's send buffer
        o = 0
        for q in range(world):
            k = exs[p].out_splits[q]
            i0 = sum(exs[q].in_splits[:p])
            assert exs[q].in_splits[p] == k
            exs[p].recv_buf[o:o + k].copy_(exs[q].send_buf[i0:i0 + k])
            o += k
        assert o == exs[p].n_recv * tile
    for r in range(world):
        v = vbs[r]
        Co = _oracle_c(v, Bfull, n)
        bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bfull, n)
        Ct = torch.full((v.rows * n,), 3.0, dtype=torch.float32, device="cuda")
        exs[r]._product("own", tiles[r], Ct, False)
        assert exs[r].remote is not None and 0 < exs[r].needed_fraction < 1
        exs[r]._product("remote", exs[r].recv_buf, Ct, True)
        torch.cuda.synchronize()
        _check(Ct.cpu().numpy(), Co, bound, "row-block exchange rank %d" % r)
        # accumulate = True on the own part: C += A * B
        exs[r]._product("own", tiles[r], Ct, True)
        exs[r]._product("remote", exs[r].recv_buf, Ct, True)
        torch.cuda.synchronize()
        _check(Ct.cpu().numpy(), 2 * Co, 2 * bound, "row-block exchange rank %d, accumulate" % r)
        exs[r].close()


def test_pack_blocks_contract():
    torch = _torch()
    import ctypes as C
    from sparta_amd._lib import lib
    src = torch.arange(64 * 40, dtype=torch.float32, device="cuda")
    ids = torch.tensor([5, 0, 5, 39, 17], dtype=torch.int32, device="cuda")
    dst = torch.zeros(5 * 64, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.sparta_pack_blocks(C.c_void_p(src.data_ptr()), 256, C.c_void_p(ids.data_ptr()), 5, C.c_void_p(dst.data_ptr()), C.c_void_p(st)) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst.view(5, 64), src.view(40, 64)[ids.long()])
    big = torch.rand(37 * 8192, device="cuda")                      # 32 KB chunks: several grid.y slices
    idb = torch.tensor([36, 1, 0, 20], dtype=torch.int32, device="cuda")
    out = torch.zeros(4 * 8192, device="cuda")
    assert lib.sparta_pack_blocks(C.c_void_p(big.data_ptr()), 32768, C.c_void_p(idb.data_ptr()), 4, C.c_void_p(out.data_ptr()), C.c_void_p(st)) == 0
    torch.cuda.synchronize()
    assert torch.equal(out.view(4, 8192), big.view(37, 8192)[idb.long()])
    assert lib.sparta_pack_blocks(C.c_void_p(src.data_ptr()), 256, C.c_void_p(ids.data_ptr()), 0, C.c_void_p(dst.data_ptr()), C.c_void_p(st)) == 0
    assert lib.sparta_pack_blocks(C.c_void_p(src.data_ptr()), 100, C.c_void_p(ids.data_ptr()), 5, C.c_void_p(dst.data_ptr()), C.c_void_p(st)) == sa._lib.ERR_INVALID
    assert lib.sparta_pack_blocks(None, 256, C.c_void_p(ids.data_ptr()), 5, C.c_void_p(dst.data_ptr()), C.c_void_p(st)) == sa._lib.ERR_INVALID


def test_block_row_range_handles_and_gathered_b():
    """the multi-GPU pieces on one GPU: row-range handles (sparta_vbs_create_range) reproduce the rows of the full
    product, and the gathered-B entry point reads an all-gather-shaped B"""
    torch = _torch()
    world, w, n = 3, 16, 40
    slabs = [sa.gen.fem3d_slab(3, 3, 5, r, world, dof=3, pad_to=w, seed=4) for r in range(world)]
    n_pad = slabs[0][2]
    shards = [sa.gen.dense_rhs(n_pad, n, seed=50 + r) for r in range(world)]
    gathered = np.concatenate(shards)                                   # what all_gather_into_tensor leaves
    Bfull = sa.dist.gathered_to_colmajor(gathered, world, n_pad, n)     # one column-major (world*n_pad) x n matrix
    for r, (m, n_local, _) in enumerate(slabs):
        g = sa.BlockingEngine(tau=0.4, col_block_size=w).GetGrouping(m)
        v = sa.VBR().fill_from_CSR_inplace(m, g, w)
        Co = _oracle_c(v, Bfull, n)
        bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bfull, n)
        d = v.to_device(0)
        Ct = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
        d.spmm_gathered(torch.from_numpy(gathered).cuda(), n_pad, Ct, n)
        torch.cuda.synchronize()
        _check(Ct.cpu().numpy(), Co, bound, "gathered B rank %d" % r)
        d.spmm_gathered(torch.from_numpy(gathered).cuda(), n_pad, Ct, n, algo=sa.SPMM_EXACT)
        torch.cuda.synchronize()
        assert np.array_equal(Ct.cpu().numpy(), Co)
        # row-range handles
        parts = sa.dist.partition_block_rows(v.row_part, v.nzcount, w, 4)
        assert parts[0][0] == 0 and parts[-1][1] == v.block_rows and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        Cfull = Co.reshape(n, v.rows)
        for b0, b1 in parts:
            if b0 == b1:
                continue
            dr = v.to_device(0, block_row_range=(b0, b1))
            r0, r1 = int(v.row_part[b0]), int(v.row_part[b1])
            assert dr.rows == r1 - r0
            Cr = torch.zeros(dr.rows * n, dtype=torch.float32, device="cuda")
            dr.spmm(torch.from_numpy(Bfull).cuda(), Cr, n, algo=sa.SPMM_EXACT)
            torch.cuda.synchronize()
            assert np.array_equal(Cr.cpu().numpy().reshape(n, dr.rows), Cfull[:, r0:r1])


def test_full_size_properties_config2():
    """BASELINE.json configs[1] at full size (cant-like 62 451^2, N = 128): size-independent properties instead of a
    full CPU product -- (i) linearity in B, (ii) column checksum 1^T C == (1^T A) B, (iii) sampled block-rows against the
    oracle, (iv) idempotence of overwrite mode (two runs bit-identical)."""
    torch = _torch()
    m = sa.gen.cant_like()
    w, n = 64, 128
    g = sa.BlockingEngine(tau=0.2, col_block_size=w).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    d = v.to_device(0)
    B1 = torch.from_numpy(sa.gen.dense_rhs(v.cols, n, seed=1)).cuda()
    B2 = torch.from_numpy(sa.gen.dense_rhs(v.cols, n, seed=2)).cuda()

    def run(Bt):
        Ct = torch.empty(v.rows * n, dtype=torch.float32, device="cuda")
        d.spmm(Bt, Ct, n, accumulate=False)
        torch.cuda.synchronize()
        return Ct
    C1, C2, C12 = run(B1), run(B2), run(0.5 * B1 + B2)
    assert torch.equal(C1, run(B1))                                            # (iv)
    scale = float(C1.abs().max())
    assert float((C12 - (0.5 * C1 + C2)).abs().max()) <= 2e-5 * scale * 4      # (i)
    colsum = np.zeros(v.cols, np.float64)                                      # (ii) 1^T A over the stored nonzeros
    np.add.at(colsum, m.colidx, m.vals.astype(np.float64))
    want = colsum @ B1.cpu().numpy().astype(np.float64).reshape(n, v.cols).T
    got = C1.cpu().numpy().astype(np.float64).reshape(n, v.rows).sum(axis=1)
    assert np.allclose(got, want, rtol=0, atol=1e-3 * max(1.0, float(np.abs(want).max())))
    rng = np.random.Generator(np.random.PCG64(0))                              # (iii)
    B1h = B1.cpu().numpy()
    C1h = C1.cpu().numpy().reshape(n, v.rows)
    absA, absB = np.abs(v.mab), np.abs(B1h)
    for ib in rng.choice(v.block_rows, size=12, replace=False):
        r0, r1 = int(v.row_part[ib]), int(v.row_part[ib + 1])
        Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B1h, n, block_row_range=(int(ib), int(ib) + 1)).reshape(n, v.rows)
        bd = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, absA, absB, n, block_row_range=(int(ib), int(ib) + 1)).reshape(n, v.rows)
        _check(C1h[:, r0:r1], Co[:, r0:r1], bd[:, r0:r1], "block-row %d" % ib)


# ---- the two branch-free product paths, forced one at a time (SPARTA_PATH), on shapes that qualify for them ----------
PRODUCT_SHAPES = [
    # rows, cols, nnz, w, blocking, N          (w % 64 == 0 or w % 32 == 0, N % 128 == 0)
    (1500, 1500, 60000, 64, ("tau", 0.5), 128),        # clustered, ragged heights, cols % w != 0 (tail block column)
    (1500, 1500, 60000, 64, ("fixed", 64), 256),       # two 128-column slabs
    (1000, 2077, 50000, 64, ("tau", 0.7), 128),        # rectangular, tail
    (2000, 1024, 40000, 128, ("fixed", 100), 128),     # w = 128: two panel steps per block (stream: four)
    (900, 960, 30000, 32, ("tau", 0.6), 128),          # w = 32: stream path only
    (3000, 640, 20000, 64, ("keeper", 48), 128),
    (64, 6400, 30000, 64, ("fixed", 64), 128),         # ONE tile with 100 blocks: split across many stream workers
]


@pytest.mark.parametrize("path", ["stream", "class"])
@pytest.mark.parametrize("rows,cols,nnz,w,blk,n", PRODUCT_SHAPES)
def test_product_paths_vs_oracle(monkeypatch, path, rows, cols, nnz, w, blk, n):
    torch = _torch()
    monkeypatch.setenv("SPARTA_PATH", path)
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols + w + n)
    if blk[0] == "tau":
        g = sa.BlockingEngine(tau=blk[1], col_block_size=w).GetGrouping(m)
    elif blk[0] == "keeper":
        g = sa.BlockingEngine(tau=0.5, col_block_size=w, row_block_size=blk[1], blocking_algo=5).GetGrouping(m)
    else:
        g = np.arange(rows) // blk[1]
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=3)
    Co = _oracle_c(v, B, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    d = v.to_device(0)
    Bcm = torch.from_numpy(B).cuda()
    Brm = torch.from_numpy(np.ascontiguousarray(B.reshape(n, v.cols).T).reshape(-1)).cuda()
    want_path = {"stream": 1, "class": 2}[path]
    if path == "class" and w % 64 != 0:
        want_path = 3                      # the per-class branch-free kernels need w % 64 == 0: falls back to generic
    for bl, Bt in ((sa.COL_MAJOR, Bcm), (sa.ROW_MAJOR, Brm)):
        for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
            Ct = torch.full((v.rows * n,), 5.0, dtype=torch.float32, device="cuda")
            d.spmm(Bt, Ct, n, accumulate=False, b_layout=bl, c_layout=cl)
            torch.cuda.synchronize()
            assert d.info()["last_path"] == want_path or d.info()["sparse_rows"] == v.rows     # (every block-row on the sparse-row path: no MFMA launch at all)
            got = Ct.cpu().numpy()
            if cl == sa.ROW_MAJOR:
                got = np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
            _check(got, Co, bound, "%s layouts %d %d" % (path, bl, cl))
    # reference semantics: accumulate onto a non-zero C
    C0 = sa.gen.dense_rhs(v.rows, n, seed=9)
    Ct = torch.from_numpy(C0).cuda()
    d.spmm(Bcm, Ct, n, accumulate=True)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), C0 + Co, bound + np.abs(C0), path + " accumulate")
    # bit-reproducible (no atomics anywhere, split tiles are summed in a fixed order)
    C1 = torch.empty(v.rows * n, dtype=torch.float32, device="cuda"); d.spmm(Bcm, C1, n)
    C2 = torch.empty(v.rows * n, dtype=torch.float32, device="cuda"); d.spmm(Bcm, C2, n)
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)


@pytest.mark.parametrize("path", ["stream", "class", "auto"])
def test_gathered_b_product_paths(monkeypatch, path):
    torch = _torch()
    monkeypatch.setenv("SPARTA_PATH", path)
    world, w, n = 4, 64, 128
    slabs = [sa.gen.fem3d_slab(4, 4, 12, r, world, dof=3, pad_to=w, seed=4) for r in range(world)]
    n_pad = slabs[0][2]
    gathered = np.concatenate([sa.gen.dense_rhs(n_pad, n, seed=50 + r) for r in range(world)])
    Bfull = sa.dist.gathered_to_colmajor(gathered, world, n_pad, n)
    for r, (m, n_local, _) in enumerate(slabs):
        g = sa.BlockingEngine(tau=0.4, col_block_size=w).GetGrouping(m)
        v = sa.VBR().fill_from_CSR_inplace(m, g, w)
        Co = _oracle_c(v, Bfull, n)
        d = v.to_device(0)
        Ct = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
        d.spmm_gathered(torch.from_numpy(gathered).cuda(), n_pad, Ct, n)
        torch.cuda.synchronize()
        _check(Ct.cpu().numpy(), Co, U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bfull, n), "gathered %s rank %d" % (path, r))
        assert d.info()["last_path"] in (1, 2)


def test_autotune_picks_a_product_path_and_is_stable():
    torch = _torch()
    m = sa.gen.fem3d(6, 6, 40, 3, seed=3)
    v = sa.VBR().fill_from_CSR_inplace_fixed(m, 64, 64)
    d = v.to_device(0)
    n = 128
    B = torch.from_numpy(sa.gen.dense_rhs(v.cols, n, seed=1)).cuda()
    C = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
    C[:] = 1.0
    d.spmm(B, C, n, accumulate=True)            # first call measures both paths on a SCRATCH C: ours is accumulated once
    torch.cuda.synchronize()
    p1 = d.info()["last_path"]
    assert p1 in (1, 2)
    Co = _oracle_c(v, B.cpu().numpy(), n) + 1.0
    _check(C.cpu().numpy(), Co, U.abs_bound(v.rows, v.cols, 64, v.row_part, v.nzcount, v.jab, v.mab, B.cpu().numpy(), n) + 1.0, "autotune")
    d.spmm(B, C, n)
    assert d.info()["last_path"] == p1


@pytest.mark.parametrize("align", ["0", "1"])
@pytest.mark.parametrize("rows,cols,nnz,w,blk,n", [
    (1500, 1500, 60000, 64, ("tau", 0.5), 128),        # one 1500-row cluster: 64-row tiles + a 28-row tile (both tile types)
    (3000, 640, 20000, 32, ("keeper", 32), 128),       # many short 32-row tiles, w = 32: one step per block
    (64, 6400, 30000, 64, ("fixed", 64), 128),         # ONE long tile: aligned = one worker, split = many segments
    (2000, 1024, 40000, 128, ("fixed", 100), 256),     # 64 + 36-row tiles, two column slabs
])
def test_stream_plan_modes(monkeypatch, align, rows, cols, nnz, w, blk, n):
    """The stream plan either cuts worker ranges anywhere (split tiles + fix-up kernel) or on tile boundaries (no
    fix-up); SPARTA_STREAM_ALIGN forces one or the other.  Both must give the oracle's product, bit-reproducibly."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_PATH", "stream")
    monkeypatch.setenv("SPARTA_STREAM_ALIGN", align)
    monkeypatch.setenv("SPARTA_SPARSE_K", "0")              # the subject is the stream plan: keep every block-row on it
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + w)
    if blk[0] == "tau":
        g = sa.BlockingEngine(tau=blk[1], col_block_size=w).GetGrouping(m)
    elif blk[0] == "keeper":
        g = sa.BlockingEngine(tau=0.5, col_block_size=w, row_block_size=blk[1], blocking_algo=5).GetGrouping(m)
    else:
        g = np.arange(rows) // blk[1]
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=4)
    Co = _oracle_c(v, B, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    d = v.to_device(0)
    info = d.info()
    if align == "1":
        assert info["split_tiles"] == 0
    Bt = torch.from_numpy(B).cuda()
    C1 = torch.full((v.rows * n,), -3.0, dtype=torch.float32, device="cuda")
    d.spmm(Bt, C1, n)
    torch.cuda.synchronize()
    assert d.info()["last_path"] == 1
    _check(C1.cpu().numpy(), Co, bound, "stream align=" + align)
    C2 = torch.empty_like(C1)
    d.spmm(Bt, C2, n)
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)


def test_clock_probe_reports_a_plausible_shader_clock():
    torch = _torch()
    m = sa.gen.fem3d(6, 6, 40, 3, seed=3)
    v = sa.VBR().fill_from_CSR_inplace_fixed(m, 64, 64)
    d = v.to_device(0)
    n = 128
    B = torch.from_numpy(sa.gen.dense_rhs(v.cols, n, seed=1)).cuda()
    C = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
    d.set_class_timing(True)
    for _ in range(3):
        d.spmm(B, C, n)
    ms = d.class_times()
    mhz = d.clock_mhz()
    d.set_class_timing(False)
    assert sum(ms.values()) > 0
    assert any(500.0 < x < 3000.0 for x in mhz.values()), mhz


# ---- 16-bit storage (fp16 / bf16), fp32 accumulation ------------------------------------------------------------------------
def _round16(x, dtype):
    """fp32 -> fp16 / bf16 -> fp32, round to nearest even (numpy for fp16, bit arithmetic for bf16)"""
    x = np.ascontiguousarray(x, np.float32)
    if dtype == sa.F16:
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


H16_SHAPES = [
    (1500, 1500, 60000, 64, ("tau", 0.5), 128),        # KP 64, both tile types, tail block column (cols % w != 0)
    (900, 960, 30000, 32, ("tau", 0.6), 128),          # KP 32
    (2000, 1024, 40000, 128, ("fixed", 100), 256),     # w = 128: two 64-deep steps per block, two column slabs
    (3000, 640, 20000, 32, ("keeper", 32), 128),       # many 32-row tiles
    (64, 6400, 30000, 64, ("fixed", 64), 128),         # one long tile
]


@pytest.mark.parametrize("kernel", ["auto", "lds", "lds-depth4", "direct"])
@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("rows,cols,nnz,w,blk,n", H16_SHAPES)
def test_16bit_storage_vs_oracle_on_rounded_inputs(monkeypatch, kernel, dtype, rows, cols, nnz, w, blk, n):
    """A and B are rounded to the 16-bit type, products of two 16-bit values are exact in fp32, accumulation is fp32: the
    oracle (fp32 VBR::multiply restatement) on the ROUNDED inputs is the reference; tolerance as for the fp32 MFMA path.
    Every 16-bit kernel is held to it: the LDS-staged one (two and four register sets) and the direct-to-register one, each
    forced on every shape, next to the library's own choice."""
    torch = _torch()
    if kernel != "auto":
        monkeypatch.setenv("SPARTA_H16_PATH", kernel.split("-")[0])
    if kernel == "lds-depth4":
        monkeypatch.setenv("SPARTA_H16_DEPTH", "4")
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols + w)
    if blk[0] == "tau":
        g = sa.BlockingEngine(tau=blk[1], col_block_size=w).GetGrouping(m)
    elif blk[0] == "keeper":
        g = sa.BlockingEngine(tau=0.5, col_block_size=w, row_block_size=blk[1], blocking_algo=5).GetGrouping(m)
    else:
        g = np.arange(rows) // blk[1]
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=5)
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Co = O.vbr_multiply(v.rows, v.cols, v.block_col_size, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    d = v.to_device(0, dtype=dtype)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    ldb = (v.cols + 7) // 8 * 8                                       # padded, even leading dimension
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
        Ct = torch.full((v.rows * n,), 7.0, dtype=torch.float32, device="cuda")
        d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl)
        torch.cuda.synchronize()
        got = Ct.cpu().numpy()
        if cl == sa.ROW_MAJOR:
            got = np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
        _check(got, Co, bound, "16-bit c_layout %d" % cl)
    # accumulate, and the host-pointer contract (fp32 B in, converted on the device)
    C0 = sa.gen.dense_rhs(v.rows, n, seed=11)
    Ct = torch.from_numpy(C0).cuda()
    d.spmm(Bt, Ct, n, ldb=ldb, accumulate=True)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), C0 + Co, bound + np.abs(C0), "16-bit accumulate")
    Ch = np.zeros(v.rows * n, np.float32)
    d.spmm_host(B, n, Ch, accumulate=False)
    _check(Ch, Co, bound, "16-bit host pointers")
    C2 = torch.empty(v.rows * n, dtype=torch.float32, device="cuda")
    C3 = torch.empty(v.rows * n, dtype=torch.float32, device="cuda")
    d.spmm(Bt, C2, n, ldb=ldb); d.spmm(Bt, C3, n, ldb=ldb)
    torch.cuda.synchronize()
    assert torch.equal(C2, C3)


@pytest.mark.parametrize("align", ["aligned", "split"])
@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("rows,cols,nnz,blk", [(3000, 640, 20000, ("keeper", 32)), (900, 960, 30000, ("tau", 0.6)), (2500, 2047, 50000, ("fixed", 32))])
def test_16bit_one_tile_kernel_variants_of_32_wide_blocks_give_the_same_bits(monkeypatch, _sparse_row_mode, align, dtype, rows, cols, nnz, blk):
    """One-tile plans of 32-wide blocks have three forms of the no-barrier kernel: 64-column waves with two sub-workers per workgroup (a property
    of the PLAN: SPARTA_H16_WIDE=1), 32-column waves three steps ahead (the default), and seven steps ahead (SPARTA_H16_AHEAD=7, per launch).  They walk
    the same steps of a tile in the same order, so with whole-tile plans their products are the same bits; the default one is held to the oracle on the
    rounded inputs.  `split` forces plans whose tiles are cut between (sub-)workers - at different steps for 512 than for 256 of them, so there each form is
    held to the oracle: the partial images of the 64-column waves must land where the fix-up kernel reads.
    (2047 columns: the last block column hangs over B - the instantiations that read B_tail.)"""
    torch = _torch()
    monkeypatch.setenv("SPARTA_STREAM_ALIGN", "1" if align == "aligned" else "0")
    w, n = 32, 128
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols)
    if blk[0] == "tau":
        g = sa.BlockingEngine(tau=blk[1], col_block_size=w).GetGrouping(m)
    elif blk[0] == "keeper":
        g = sa.BlockingEngine(tau=0.5, col_block_size=w, row_block_size=blk[1], blocking_algo=5).GetGrouping(m)
    else:
        g = np.arange(rows) // blk[1]
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=5)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    ldb = (v.cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    outs = {}
    for name, env in [("wide", {"SPARTA_H16_WIDE": "1"}), ("narrow", {"SPARTA_H16_WIDE": "0"}), ("narrow7", {"SPARTA_H16_WIDE": "0", "SPARTA_H16_AHEAD": "7"})]:
        for k in ("SPARTA_H16_WIDE", "SPARTA_H16_AHEAD"):
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        d = v.to_device(0, dtype=dtype)                                 # (the plan is made here)
        res = []
        for cl, acc in ((sa.COL_MAJOR, False), (sa.ROW_MAJOR, False), (sa.COL_MAJOR, True)):
            Ct = torch.full((v.rows * n,), 0.5, dtype=torch.float32, device="cuda")
            d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl, accumulate=acc)
            torch.cuda.synchronize()
            res.append(Ct)
        outs[name] = res
        d.close()
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    for name in (("wide",) if align == "aligned" else ("wide", "narrow", "narrow7")):
        _check(outs[name][0].cpu().numpy(), Co, bound, name + ", column-major C")
        _check(np.ascontiguousarray(outs[name][1].cpu().numpy().reshape(v.rows, n).T).reshape(-1), Co, bound, name + ", row-major C")
        _check(outs[name][2].cpu().numpy(), Co + 0.5, bound + 0.5, name + ", accumulate")
    if align == "aligned":                                              # (split plans cut the tiles at different steps for 2 x as many sub-workers: other partial sums)
        for name in ("narrow", "narrow7"):
            for a, b in zip(outs["wide"], outs[name]):
                assert torch.equal(a, b), name


@pytest.mark.parametrize("plan", ["narrow", "wide"])
@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("rows,cols,nnz,blk,n", [(3000, 640, 20000, ("keeper", 32), 256), (1200, 2047, 50000, ("fixed", 32), 512), (900, 960, 30000, ("tau", 0.6), 256)])
def test_16bit_256_column_slabs_give_the_bits_of_the_128_column_ones(monkeypatch, _sparse_row_mode, plan, dtype, rows, cols, nnz, blk, n):
    """N % 256 == 0, one-tile plan of 32-wide blocks without split tiles: the four 64-column waves of a workgroup can take ONE tile over a 256-column slab
    (SPARTA_H16_SLAB256=1; A is then read once per 256 columns).  Same steps, same order per output element: the same bits as the 128-column launch, on the
    plan with two sub-worker ranges per workgroup as on the plain one; and the oracle's product on the rounded inputs."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_STREAM_ALIGN", "1")
    monkeypatch.setenv("SPARTA_H16_WIDE", "1" if plan == "wide" else "0")
    w = 32
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols + n)
    if blk[0] == "tau":
        g = sa.BlockingEngine(tau=blk[1], col_block_size=w).GetGrouping(m)
    elif blk[0] == "keeper":
        g = sa.BlockingEngine(tau=0.5, col_block_size=w, row_block_size=blk[1], blocking_algo=5).GetGrouping(m)
    else:
        g = np.arange(rows) // blk[1]
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=5)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    ldb = (v.cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    d = v.to_device(0, dtype=dtype)
    outs = {}
    for slab in ("0", "1"):
        monkeypatch.setenv("SPARTA_H16_SLAB256", "2" if slab == "1" else "0")      # (2: also on the plan with two sub-worker ranges)
        res = []
        for cl, acc in ((sa.COL_MAJOR, False), (sa.ROW_MAJOR, False), (sa.COL_MAJOR, True)):
            Ct = torch.full((v.rows * n,), 0.5, dtype=torch.float32, device="cuda")
            d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl, accumulate=acc)
            torch.cuda.synchronize()
            res.append(Ct)
        outs[slab] = res
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    _check(outs["1"][0].cpu().numpy(), Co, bound, "256-column slabs, column-major C")
    _check(np.ascontiguousarray(outs["1"][1].cpu().numpy().reshape(v.rows, n).T).reshape(-1), Co, bound, "256-column slabs, row-major C")
    _check(outs["1"][2].cpu().numpy(), Co + 0.5, bound + 0.5, "256-column slabs, accumulate")


@pytest.mark.parametrize("align", ["whole-tiles", "split"])
@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("rows,cols,nnz,n", [(640, 6400, 260000, 256), (200, 12800, 300000, 512), (1000, 2047, 150000, 256)])
def test_16bit_four_accumulator_slabs_give_the_bits_of_the_128_column_launch(monkeypatch, _sparse_row_mode, align, dtype, rows, cols, nnz, n):
    """64-row tiles of 64-wide blocks, N % 256 == 0 (the dense hub of a power-law matrix under the fixed 64 x 64 grid): four accumulators per wave over
    256-column slabs (k_h16.hip, QUAD; A read once per 256 columns).  Same steps in the same order per output element: the bits of the 128-column launch
    (SPARTA_H16_QUAD=0) -- on whole-tile plans and on split (stream-K) plans, whose partial images go through the same fix-up -- with a partial last block
    column (cols % 64 != 0), both layouts of C, accumulate, and a gathered B; and the oracle's product on the rounded inputs."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_STREAM_ALIGN", "1" if align == "whole-tiles" else "0")
    w = 64
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols + n)
    g = np.arange(rows) // 64
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=5)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    ldb = (v.cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    d = v.to_device(0, dtype=dtype)
    info = d.info()
    assert info["tiles64"] > 0
    if align == "split":
        assert info["split_tiles"] > 0, "this shape was meant to give a split plan"
    world = 2
    shard_rows = v.cols // world if v.cols % (world * w) == 0 else 0
    Bg = None
    if shard_rows:
        Bg = torch.zeros(world * shard_rows * n, dtype=tdt, device="cuda")
        for s_ in range(world):
            Bg[s_ * shard_rows * n:(s_ + 1) * shard_rows * n].view(n, shard_rows)[:] = Bt.view(n, ldb)[:, s_ * shard_rows:(s_ + 1) * shard_rows]
    outs = {}
    for quad in ("0", "1"):
        monkeypatch.setenv("SPARTA_H16_QUAD", quad)
        res = []
        for cl, acc in ((sa.COL_MAJOR, False), (sa.ROW_MAJOR, False), (sa.COL_MAJOR, True)):
            Ct = torch.full((v.rows * n,), 0.5, dtype=torch.float32, device="cuda")
            d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl, accumulate=acc)
            torch.cuda.synchronize()
            res.append(Ct)
        if Bg is not None:
            Ct = torch.full((v.rows * n,), 0.5, dtype=torch.float32, device="cuda")
            d.spmm_gathered(Bg, shard_rows, Ct, n)
            torch.cuda.synchronize()
            res.append(Ct)
        outs[quad] = res
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    _check(outs["1"][0].cpu().numpy(), Co, bound, "four accumulators, column-major C")
    _check(np.ascontiguousarray(outs["1"][1].cpu().numpy().reshape(v.rows, n).T).reshape(-1), Co, bound, "four accumulators, row-major C")
    _check(outs["1"][2].cpu().numpy(), Co + 0.5, bound + 0.5, "four accumulators, accumulate")
    if Bg is not None:
        _check(outs["1"][3].cpu().numpy(), Co, bound, "four accumulators, gathered B")


@pytest.mark.parametrize("n", [256, 200, 72])
@pytest.mark.parametrize("dtype", [sa.F32, sa.F16, sa.BF16], ids=["f32", "f16", "bf16"])
def test_a_constant_b_prepared_once_gives_the_bits_of_the_plain_product(dtype, n):
    """sparta_vbs_prepare_b / sparta_vbs_spmm_prepared: the row-major copy of B the sparse-row kernels read is made once instead of per product; everything
    else is the same call -- the same bits, column-major B and the gathered layout; a B prepared for one handle is refused by another shape.  n = 200 / 72: a 16-bit
    call is cut into whole 128-column slabs (which gather from the prepared copy, rows n elements apart) and a zero-padded tail slab (which must not)."""
    torch = _torch()
    w = 64
    m = sa.gen.rmat(13, 60000, seed=5, symmetrize=True, pattern_only=False)
    g = sa.BlockingEngine(blocking_algo=7, tau=0.4, col_block_size=w, row_block_size=32).GetGrouping(m)
    d = sa.DeviceVBS.from_csr(m, g, w, device=0, dtype=dtype)
    if d.sparse_info()["nnz"] == 0:
        pytest.skip("no sparse rows in this mode (SPARTA_SPARSE_K=0): nothing to prepare")
    tdt = {sa.F32: torch.float32, sa.F16: torch.float16, sa.BF16: torch.bfloat16}[dtype]
    B = (torch.rand(d.cols * n, generator=torch.Generator().manual_seed(3)) - 0.5).to(tdt).cuda()
    C1 = torch.full((d.rows * n,), 2.0, dtype=torch.float32, device="cuda")
    C2 = torch.full((d.rows * n,), 3.0, dtype=torch.float32, device="cuda")
    d.spmm(B, C1, n)
    Bp = d.prepare_b(B, n)
    for _ in range(2):                                                      # prepared once, used twice
        d.spmm_prepared(Bp, C2)
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)
    d.spmm(B, C1, n, accumulate=True)
    d.spmm_prepared(Bp, C2, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)
    # gathered layout (two shards)
    shard_rows = d.cols // 2
    Bg = torch.empty_like(B)
    for s_ in range(2):
        Bg[s_ * shard_rows * n:(s_ + 1) * shard_rows * n].view(n, shard_rows)[:] = B.view(n, d.cols)[:, s_ * shard_rows:(s_ + 1) * shard_rows]
    Bpg = d.prepare_b(Bg, n, shard_rows=shard_rows)
    C3 = torch.full((d.rows * n,), 4.0, dtype=torch.float32, device="cuda")
    C4 = torch.full((d.rows * n,), 5.0, dtype=torch.float32, device="cuda")
    d.spmm_gathered(Bg, shard_rows, C3, n)
    d.spmm_prepared(Bpg, C4)
    torch.cuda.synchronize()
    assert torch.equal(C3, C4)
    other = sa.DeviceVBS.from_csr(sa.gen.uniform_random(300, 512, 4000, seed=1), np.arange(300) // 32, w, device=0, dtype=dtype)
    with pytest.raises(ValueError):
        other.spmm_prepared(Bp, C2)
    Bp.close(); Bpg.close()
    with pytest.raises(ValueError):
        d.spmm_prepared(Bp, C2)


@pytest.mark.parametrize("rows_per_tile", [32, 19, 5])
def test_fp32_k_compaction_every_count_of_non_empty_columns(monkeypatch, rows_per_tile):
    """the fp32 no-barrier kernel multiplies only the non-empty columns of a step's 32 x 32 slice of A (vbs_plan.cpp: compacted fragment image, position table,
    MFMA pairs per step): blocks with exactly 1, 2, 3, ..., 32 non-empty columns in scattered positions, full and short tiles, a partial last block column --
    every element against the oracle, and bit-identical to the LDS-staged kernel's... tolerance class (same products, another order of the k sum)."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_PATH", "stream")                           # the no-barrier kernel, not whichever path the autotune likes on this small case
    w, n = 32, 128
    rng = np.random.Generator(np.random.PCG64(77 + rows_per_tile))
    n_tiles, n_bcols = 40, 33
    cols = n_bcols * w - 11                                               # cols % w != 0: the last block column is partial (B_tail steps)
    rr, cc = [], []
    for t in range(n_tiles):
        for jb in rng.choice(n_bcols, size=6, replace=False):
            nk = int(rng.integers(1, 33)) if jb != n_bcols - 1 else int(rng.integers(1, w - 11 + 1))
            width = w if jb != n_bcols - 1 else w - 11
            ks = rng.choice(width, size=nk, replace=False)              # exactly nk non-empty columns, anywhere in the block
            for k in ks:
                rws = rng.choice(rows_per_tile, size=int(rng.integers(1, rows_per_tile + 1)), replace=False)
                rr.append(t * rows_per_tile + rws)
                cc.append(np.full(len(rws), jb * w + k))
    r, c = np.concatenate(rr), np.concatenate(cc)
    m = sa.gen._csr_from_coo(n_tiles * rows_per_tile, cols, r, c, rng.uniform(-1, 1, len(r)).astype(np.float32))
    g = np.arange(m.rows) // rows_per_tile
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    nk_hist = set()
    off = 0
    for ib in range(v.block_rows):
        h = int(v.row_part[ib + 1] - v.row_part[ib])
        for _ in range(int(v.nzcount[ib])):
            nk_hist.add(int((v.mab[off:off + h * w].reshape(w, h) != 0).any(axis=1).sum()))
            off += h * w
    assert len(nk_hist) >= 20, sorted(nk_hist)                          # the case really covers the range of column counts
    B = sa.gen.dense_rhs(v.cols, n, seed=9)
    Co = _oracle_c(v, B, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    d = v.to_device(0)
    info = d.info()
    for acc in (False, True):
        C0 = sa.gen.dense_rhs(v.rows, n, seed=12) if acc else np.full(v.rows * n, 5.0, np.float32)
        Ct = torch.from_numpy(C0.copy()).cuda()
        d.spmm(torch.from_numpy(B).cuda(), Ct, n, accumulate=acc)
        torch.cuda.synchronize()
        _check(Ct.cpu().numpy(), Co + (C0 if acc else 0.0), bound + (np.abs(C0) if acc else 0.0), "k-compaction, %d rows per tile, accumulate %s" % (rows_per_tile, acc))
    if info["sparse_rows"] == 0:
        assert d.info()["last_path"] == 1, "the stream kernels were meant to run"


@pytest.mark.parametrize("rows,cols,nnz,w,blk", [(2048, 2048, 60000, 32, ("keeper", 16)), (1500, 1000 - 7, 40000, 32, ("fixed", 32)), (900, 640, 30000, 64, ("fixed", 19))])
def test_fp32_handle_keeps_one_image_of_a_and_rebuilds_the_reference_layout_on_demand(monkeypatch, _sparse_row_mode, rows, cols, nnz, w, blk):
    """a fp32 handle whose tiles all have <= 32 rows is created with A in two layouts (the reference's mab, and the fragment image the no-barrier kernel reads); once that
    kernel carries the products the reference-layout image is dropped (a_bytes = one image, vbs_capi.cpp: drop_legacy_image) and rebuilt on the device -- an exact copy --
    when a later call reads it: SPARTA_SPMM_EXACT must still be bit-identical to the reference's VBR::multiply, a row-major B and the per-class kernels within the bar."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_PATH", "stream")
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + w)
    g = sa.BlockingEngine(tau=0.5, col_block_size=w, row_block_size=blk[1], blocking_algo=5).GetGrouping(m) if blk[0] == "keeper" else np.arange(rows) // blk[1]
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    n = 128
    B = sa.gen.dense_rhs(v.cols, n, seed=5)
    Co = _oracle_c(v, B, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    d = v.to_device(0)
    two = d.info()["a_bytes"]
    legacy = (int(v.mab.size) + 128) * 4
    Bt = torch.from_numpy(B).cuda()
    Ct = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
    d.spmm(Bt, Ct, n)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), Co, bound, "first product")
    one = d.info()["a_bytes"]
    if d.info()["stream_steps"] > 0 and d.info()["sparse_rows"] == 0:            # (a handle with sparse rows keeps it: their blocks are not in the fragment image)
        assert one == two - legacy, "the reference-layout image of A was meant to be dropped after the first product (%d -> %d, image %d)" % (two, one, legacy)
    Ct.zero_()
    d.spmm(Bt, Ct, n)                                                     # ... and the next products run without it
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), Co, bound, "second product")
    Cx = np.zeros(v.rows * n, np.float32)
    d.spmm_host(B, n, Cx, accumulate=True, algo=sa.SPMM_EXACT)           # reads the reference layout: rebuilt from the fragment image, bit for bit
    assert np.array_equal(Cx, Co), "exact-order kernel after the rebuild is not bit-identical to the reference's VBR::multiply"
    assert d.info()["a_bytes"] == two
    Brm = torch.from_numpy(np.ascontiguousarray(B.reshape(n, v.cols).T)).cuda()      # row-major B: the LDS-staged kernels, reference-layout image
    Ct.zero_()
    d.spmm(Brm, Ct, n, b_layout=sa.ROW_MAJOR)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), Co, bound, "row-major B after the rebuild")
    Ct.zero_()
    d.spmm(Bt, Ct, n)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), Co, bound, "column-major B again")
    assert d.info()["a_bytes"] == two, "a rebuilt image stays (no free / rebuild ping-pong)"
    # a handle whose FIRST use reads the reference layout after a drop: row-major B straight after the first product
    d2 = v.to_device(0)
    Ct.zero_()
    d2.spmm(Bt, Ct, n)
    Ct.zero_()
    d2.spmm(Brm, Ct, n, b_layout=sa.ROW_MAJOR)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), Co, bound, "row-major B on a handle that had dropped the image")


def test_small_matrices_whose_tiles_are_not_worth_their_launches_go_to_the_sparse_rows(monkeypatch):
    """every launch of a product costs 5-10 us whatever it does: a small matrix whose tiles hold fewer nonzeros than their steps + three launches are worth is
    multiplied by the sparse-row kernels alone (one launch chain instead of two); a small DENSE matrix keeps its tiles.  Same product either way."""
    torch = _torch()
    monkeypatch.delenv("SPARTA_LAUNCH_NNZ", raising=False)
    monkeypatch.delenv("SPARTA_SPARSE_K", raising=False)
    monkeypatch.delenv("SPARTA_SPARSE_MIN_STEPS", raising=False)
    n, w = 128, 32
    thin = sa.gen.banded(6000, 40, density=0.08, seed=3)                    # ~39 k nonzeros in thin blocks
    dense = sa.gen.uniform_random(1024, 1024, 600000, seed=4)               # 57 % fill: tiles
    for m, want_tiles in ((thin, False), (dense, True)):
        g = np.arange(m.rows) // 32
        st = sa.DeviceVBS.plan_stats(m, g, w, 32, False)
        assert (st["tile_blocks"] > 0) == want_tiles, st
        d = sa.DeviceVBS.from_csr(m, g, w, 32, False, device=0)
        assert (d.info()["nblocks"] > 0) == want_tiles and d.info()["nblocks"] == st["tile_blocks"]
        v = sa.VBR().fill_from_CSR_inplace(m, g, w, 32, False)
        B = sa.gen.dense_rhs(v.cols, n, seed=5)
        Co = _oracle_c(v, B, n)
        bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
        C = torch.full((v.rows * n,), 9.0, dtype=torch.float32, device="cuda")
        d.spmm(torch.from_numpy(B).cuda(), C, n)
        torch.cuda.synchronize()
        _check(C.cpu().numpy(), Co, bound, "small matrix, tiles %s" % want_tiles)
        d.close()


@pytest.mark.parametrize("align", ["whole-tiles", "split"])
@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("w,h2,n", [(32, 32, 128), (32, 7, 128), (32, 1, 256), (64, 32, 256), (64, 19, 128)])
def test_16bit_pair_tiles_against_the_oracle_and_the_unpaired_plan(monkeypatch, _sparse_row_mode, align, dtype, w, h2, n):
    """pair tiles (vbs_plan.cpp): block-rows of 32 rows followed by block-rows of h2 rows are walked in pairs as 64-row tiles over the union of their block columns;
    a block only one of the two has = a zero half that is not fetched.  Blocks present in the upper row only, the lower only, both; a partial last block column;
    whole-tile and split plans; both layouts of C, accumulate, gathered B -- against the oracle on the rounded inputs and against the plan without pairs."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_STREAM_ALIGN", "1" if align == "whole-tiles" else "0")
    rng = np.random.Generator(np.random.PCG64(w + h2 + n))
    n_pairs, n_bcols = (24 if align == "whole-tiles" else 3), (40 if align == "whole-tiles" else 150)
    cols = n_bcols * w - 5
    rows = n_pairs * (32 + h2) + 13                                       # + a last block-row that finds no partner
    bounds = np.concatenate([[0], np.cumsum(np.tile([32, h2], n_pairs)), [rows]])
    g = np.repeat(np.arange(len(bounds) - 1), np.diff(bounds))
    rr, cc = [], []
    for ib in range(len(bounds) - 1):
        h = int(bounds[ib + 1] - bounds[ib])
        for jb in np.flatnonzero(rng.random(n_bcols) < 0.45):               # ~45 % of the block columns per block-row: every presence pattern occurs
            width = min(w, cols - jb * w)
            cnt = int(h * width * 0.4) + 1
            flat = rng.choice(h * width, size=min(cnt, h * width), replace=False)
            rr.append(bounds[ib] + flat // width)
            cc.append(jb * w + flat % width)
    r, c = np.concatenate(rr), np.concatenate(cc)
    m = sa.gen._csr_from_coo(rows, cols, r, c, rng.uniform(-1, 1, len(r)).astype(np.float32))
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=5)
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    ldb = (v.cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    outs = {}
    for pair in ("1", "0"):
        monkeypatch.setenv("SPARTA_H16_PAIR", pair)
        d = v.to_device(0, dtype=dtype)
        info = d.info()
        if _sparse_row_mode == "mfma-only":
            assert (info["stream_steps"] < sum(int(x) for x in v.nzcount) * (w // (64 if w % 64 == 0 else 32))) == (pair == "1"), "pairs share steps"
        if align == "split":
            assert info["split_tiles"] > 0
        res = []
        for cl, acc in ((sa.COL_MAJOR, False), (sa.ROW_MAJOR, False), (sa.COL_MAJOR, True)):
            Ct = torch.full((v.rows * n,), 0.5, dtype=torch.float32, device="cuda")
            d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl, accumulate=acc)
            torch.cuda.synchronize()
            res.append(Ct.cpu().numpy())
        outs[pair] = res
        d.close()
    for res in outs.values():
        _check(res[0], Co, bound, "pair tiles, column-major C")
        _check(np.ascontiguousarray(res[1].reshape(v.rows, n).T).reshape(-1), Co, bound, "pair tiles, row-major C")
        _check(res[2], Co + 0.5, bound + 0.5, "pair tiles, accumulate")


def test_16bit_handles_reject_what_they_cannot_do():
    torch = _torch()
    m = sa.gen.uniform_random(256, 256, 3000, seed=2)
    v48 = sa.VBR().fill_from_CSR_inplace_fixed(m, 32, 48)
    with pytest.raises(sa.SpartaError):
        v48.to_device(0, dtype=sa.F16)                                # w % 32 != 0
    v = sa.VBR().fill_from_CSR_inplace_fixed(m, 32, 32)
    d = v.to_device(0, dtype=sa.F16)
    B = torch.zeros(v.cols * 128, dtype=torch.float16, device="cuda")
    C = torch.zeros(v.rows * 128, dtype=torch.float32, device="cuda")
    with pytest.raises(sa.SpartaError):
        d.spmm(B, C, 128, b_layout=sa.ROW_MAJOR)
    with pytest.raises(sa.SpartaError):
        d.spmm(B, C, 128, algo=sa.SPMM_EXACT)
    with pytest.raises(ValueError):
        d.spmm(B.float(), C, 128)                                     # fp32 B on a 16-bit handle
    B96 = torch.zeros(v.cols * 96, dtype=torch.float16, device="cuda")
    d.spmm(B96, C, 96)                                                # n_cols % 128 != 0 is NOT one of them any more (round 3: tail slab)
    Bodd = torch.zeros((v.cols + 1) * 128, dtype=torch.float16, device="cuda")
    with pytest.raises(sa.SpartaError):
        d.spmm(Bodd, C, 128, ldb=v.cols + 1)                          # odd leading dimension


H16_ANY_N = [
    (1500, 1500, 60000, 64, ("tau", 0.5)),            # KP 64, both tile types, tail block column (cols % w != 0)
    (900, 960, 30000, 32, ("keeper", 32)),            # KP 32, one-tile plan
    (700, 2048, 9000, 64, ("csr", 0.5)),              # handle straight from the CSR: most block-rows on the sparse-row kernels
]


@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("n", [1, 37, 200, 300])
@pytest.mark.parametrize("rows,cols,nnz,w,blk", H16_ANY_N)
def test_16bit_handles_take_any_number_of_columns(dtype, n, rows, cols, nnz, w, blk):
    """the reference's -c (columns of B) is arbitrary (include/input.h:15-42): whole 128-column slabs + one zero-padded tail slab through the
    same kernels, both layouts of C, accumulate, host pointers, and the gathered layout of a multi-GPU B"""
    torch = _torch()
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols + w + 1)
    if blk[0] == "keeper":
        g = sa.BlockingEngine(tau=0.5, col_block_size=w, row_block_size=blk[1], blocking_algo=5).GetGrouping(m)
    else:
        g = sa.BlockingEngine(tau=blk[1], col_block_size=w).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    d = sa.DeviceVBS.from_csr(m, g, w, device=0, dtype=dtype) if blk[0] == "csr" else v.to_device(0, dtype=dtype)
    B = sa.gen.dense_rhs(v.cols, n, seed=5)
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    ldb = (v.cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n + 64, dtype=tdt, device="cuda")
    Bt[ldb * n:] = float("nan")                                       # what follows B in memory must never be read into the product
    Bt[:ldb * n].view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
        Ct = torch.full((v.rows * n + 32,), 7.0, dtype=torch.float32, device="cuda")
        d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl)
        torch.cuda.synchronize()
        got = Ct[:v.rows * n].cpu().numpy()
        assert float(Ct[v.rows * n:].min()) == 7.0 and float(Ct[v.rows * n:].max()) == 7.0, "wrote past the end of C"
        if cl == sa.ROW_MAJOR:
            got = np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
        _check(got, Co, bound, "16-bit n_cols %d c_layout %d" % (n, cl))
    C0 = sa.gen.dense_rhs(v.rows, n, seed=11)
    Ct = torch.from_numpy(C0).cuda()
    d.spmm(Bt, Ct, n, ldb=ldb, accumulate=True)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), C0 + Co, bound + np.abs(C0), "16-bit n_cols %d accumulate" % n)
    Ch = np.zeros(v.rows * n, np.float32)
    d.spmm_host(B, n, Ch, accumulate=False)
    _check(Ch, Co, bound, "16-bit n_cols %d host pointers" % n)
    # gathered B: `world` column-major slabs of shard_rows x n (shard_rows a multiple of w; the matrix is padded with zero columns up to it)
    world = 2
    shard_rows = -(-v.cols // (world * w)) * w
    if world * shard_rows == v.cols:
        Bg = torch.zeros(world * shard_rows * n, dtype=tdt, device="cuda")
        Bfull = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
        for s_ in range(world):
            Bg[s_ * shard_rows * n:(s_ + 1) * shard_rows * n].view(n, shard_rows)[:] = Bfull[:, s_ * shard_rows:(s_ + 1) * shard_rows]
        Ct = torch.full((v.rows * n,), 7.0, dtype=torch.float32, device="cuda")
        d.spmm_gathered(Bg, shard_rows, Ct, n)
        torch.cuda.synchronize()
        _check(Ct.cpu().numpy(), Co, bound, "16-bit n_cols %d gathered" % n)


def test_huge_leading_dimensions_take_the_64bit_kernels():
    """The stream kernels address a 128-column slab with 32-bit byte offsets: leading dimensions beyond 4.2 M elements must
    not reach them (a buffer load past 2 GB returns 0, a store is dropped -- silently)."""
    torch = _torch()
    m = sa.gen.uniform_random(700, 900, 9000, seed=31)
    v = sa.VBR().fill_from_CSR_inplace_fixed(m, 64, 64)
    n, ld = 128, 4_400_000
    B = sa.gen.dense_rhs(v.cols, n, seed=2)
    Co = _oracle_c(v, B, n)
    bound = U.abs_bound(v.rows, v.cols, 64, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    d = v.to_device(0)
    Bt = torch.zeros(ld * n, dtype=torch.float32, device="cuda")
    Bt.view(n, ld)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda()
    Ct = torch.full((ld * n,), 3.0, dtype=torch.float32, device="cuda")
    d.spmm(Bt, Ct, n, ldb=ld, ldc=ld)
    torch.cuda.synchronize()
    assert d.info()["last_path"] in (2, 3)
    got = Ct.view(n, ld)[:, :v.rows].contiguous().view(-1).cpu().numpy()
    _check(got, Co, bound, "huge ld")
    assert float(Ct.view(n, ld)[:, v.rows:v.rows + 8].min()) == 3.0          # nothing written outside the rows of C
    del Bt, Ct
    torch.cuda.empty_cache()


# ---- sparse-row path: nearly empty block-rows are multiplied as rows of (column, value) --------------------------------

@pytest.mark.parametrize("n", [1, 37, 64, 128, 256, 320])
def test_sparse_row_path_mixed_matrix(_sparse_row_mode, n):
    """a matrix with BOTH kinds of block-rows: dense clusters (MFMA tiles) and scattered singletons, a hub row (> 4096 nonzeros:
    the workgroup-per-row kernel), empty rows, a ragged last block column; every layout, accumulate, host pointers, gathered B"""
    torch = _torch()
    m, w = _mixed_matrix()
    g = sa.BlockingEngine(tau=0.5, col_block_size=w).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=11)
    _mixed_checks(torch, _sparse_row_mode, v, w, B, n)


def _mixed_matrix():
    rng = np.random.Generator(np.random.PCG64(5))
    rows, cols, w = 700, 9000 + 13, 32
    rr, cc = [], []
    for i in range(0, 256):                                   # dense part: 256 rows sharing 40 columns blocks -> tall, full blocks
        c = rng.choice(1280, 400, replace=False)
        rr.append(np.full(400, i)); cc.append(c)
    for i in range(256, 690):                                 # scattered part: 1-6 nonzeros anywhere
        k = int(rng.integers(1, 7))
        rr.append(np.full(k, i)); cc.append(rng.choice(cols, k, replace=False))
    rr.append(np.full(6000, 690)); cc.append(rng.choice(cols, 6000, replace=False))     # hub row
    r, c = np.concatenate(rr), np.concatenate(cc)                                        # rows 691..699 stay empty
    order = np.lexsort((c, r))
    r, c = r[order], c[order]
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=rows))])
    return sa.CSR(rows, cols, rowptr, c.astype(np.int32), rng.uniform(-1, 1, len(c)).astype(np.float32)), w


def _mixed_checks(torch, _sparse_row_mode, v, w, B, n):
    Co = _oracle_c(v, B, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    d = v.to_device(0)
    info = d.info()
    if _sparse_row_mode == "with-sparse-rows":
        assert 0 < info["sparse_rows"] < v.rows and info["tiles16"] + info["tiles32"] + info["tiles64"] > 0
    elif _sparse_row_mode == "mfma-only":
        assert info["sparse_rows"] == 0
    # (library-defaults: whatever the library decides for a matrix this small -- _carrier records it)
    Bcm = torch.from_numpy(B).cuda()
    Brm = torch.from_numpy(np.ascontiguousarray(B.reshape(n, v.cols).T).reshape(-1)).cuda()
    for bl, Bt in ((sa.COL_MAJOR, Bcm), (sa.ROW_MAJOR, Brm)):
        for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
            for acc in (False, True):
                C0 = sa.gen.dense_rhs(v.rows, n, seed=12)
                Ct = torch.from_numpy(C0 if cl == sa.COL_MAJOR else np.ascontiguousarray(C0.reshape(n, v.rows).T).reshape(-1)).cuda()
                d.spmm(Bt, Ct, n, accumulate=acc, b_layout=bl, c_layout=cl)
                torch.cuda.synchronize()
                got = Ct.cpu().numpy()
                if cl == sa.ROW_MAJOR:
                    got = np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
                want = _oracle_c(v, B, n, C0) if acc else Co
                _check(got, want, bound + (np.abs(C0) if acc else 0), "sparse rows n=%d layouts %d %d acc %d" % (n, bl, cl, acc))
                _carrier(d, _sparse_row_mode, "mixed n=%d B %s C %s" % (n, "col" if bl == sa.COL_MAJOR else "row", "col" if cl == sa.COL_MAJOR else "row"),
                         reference_layouts=(bl == sa.COL_MAJOR and cl == sa.COL_MAJOR))
    # host pointers (the reference's contract: C += A * B)
    Ch = sa.gen.dense_rhs(v.rows, n, seed=13)
    want = _oracle_c(v, B, n, Ch)
    d.spmm_host(B, n, Ch, accumulate=True)
    _check(Ch, want, bound + np.abs(want), "sparse rows, host pointers")
    # the exact-order kernel still covers every row
    Ce = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
    d.spmm(Bcm, Ce, n, algo=sa.SPMM_EXACT)
    torch.cuda.synchronize()
    assert np.array_equal(Ce.cpu().numpy(), Co)


def test_sparse_row_path_gathered_b(_sparse_row_mode):
    torch = _torch()
    world, w, n = 2, 16, 96
    m = sa.gen.uniform_random(500, 2 * 640, 4000, seed=8)          # scattered: everything goes to the sparse-row path
    n_pad = 640
    v = sa.VBR().fill_from_CSR_inplace(m, sa.BlockingEngine(tau=0.3, col_block_size=w).GetGrouping(m), w)
    gathered = np.concatenate([sa.gen.dense_rhs(n_pad, n, seed=60 + r) for r in range(world)])
    Bfull = sa.dist.gathered_to_colmajor(gathered, world, n_pad, n)
    d = v.to_device(0)
    Ct = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
    d.spmm_gathered(torch.from_numpy(gathered).cuda(), n_pad, Ct, n)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), _oracle_c(v, Bfull, n), U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bfull, n), "sparse rows, gathered B")
    if _sparse_row_mode == "with-sparse-rows":
        assert d.info()["sparse_rows"] > 0


@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("n", [128, 256, 512])                  # 512: two column chunks per row (blockIdx.y) with 4-element vectors
def test_sparse_row_path_16bit(_sparse_row_mode, dtype, n):
    """16-bit handles: the sparse rows hold the ROUNDED values of A and read a row-major 16-bit copy of B; reference = the oracle
    on the rounded inputs (products of two 16-bit values are exact in fp32), tolerance as everywhere"""
    torch = _torch()
    m, w = _mixed_matrix()
    v = sa.VBR().fill_from_CSR_inplace(m, sa.BlockingEngine(tau=0.5, col_block_size=w).GetGrouping(m), w)
    B = sa.gen.dense_rhs(v.cols, n, seed=21)
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    d = v.to_device(0, dtype=dtype)
    assert (d.info()["sparse_rows"] > 0) == (_sparse_row_mode == "with-sparse-rows")
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    ldb = (v.cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
        for acc in (False, True):
            C0 = sa.gen.dense_rhs(v.rows, n, seed=22)
            Ct = torch.from_numpy(C0 if cl == sa.COL_MAJOR else np.ascontiguousarray(C0.reshape(n, v.rows).T).reshape(-1)).cuda()
            d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl, accumulate=acc)
            torch.cuda.synchronize()
            got = Ct.cpu().numpy()
            if cl == sa.ROW_MAJOR:
                got = np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
            _check(got, Co + (C0 if acc else 0), bound + (np.abs(C0) if acc else 0), "16-bit sparse rows c_layout %d acc %d" % (cl, acc))
    Ch = np.zeros(v.rows * n, np.float32)
    d.spmm_host(B, n, Ch, accumulate=False)
    _check(Ch, Co, bound, "16-bit sparse rows, host pointers")


@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("win,min_steps,n", [(1024, 4, 256), (512, 1, 128), (4096, 16, 512)])
def test_16bit_window_plan_of_64_row_tiles(monkeypatch, _sparse_row_mode, dtype, win, min_steps, n):
    """SPARTA_TILE_WINDOW_COLS (an experiment, off by default: DESIGN.md section 10): the 64-row tiles of a 16-bit handle are cut into (tile, column window) pieces,
    the pieces dealt whole to the workers window by window, every piece of a cut tile a partial image that the fix-up adds in k order.  Same product as the oracle's on the
    rounded inputs (both layouts of C, accumulate) and as the default plan's up to the rounding of the regrouped sums; a partial last block column and a ragged last tile."""
    torch = _torch()
    w, rows, cols = 64, 1000, 8000 + 37
    m = sa.gen.uniform_random(rows, cols, 400000, seed=win + n)
    g = np.arange(rows) // 64
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=6)
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    ldb = (v.cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    outs = []
    for on in (True, False):
        if on:
            monkeypatch.setenv("SPARTA_TILE_WINDOW_COLS", str(win))
            monkeypatch.setenv("SPARTA_TILE_WINDOW_MIN", str(min_steps))
        else:
            monkeypatch.delenv("SPARTA_TILE_WINDOW_COLS")
        d = v.to_device(0, dtype=dtype)
        info = d.info()
        if _sparse_row_mode == "mfma-only":
            assert info["tiles64"] > 0
            if on:
                assert info["split_tiles"] >= info["tiles64"] - 1, info        # every tile (all of them span several windows) goes through the fix-up
        for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
            for acc in (False, True):
                C0 = sa.gen.dense_rhs(rows, n, seed=33)
                Ct = torch.from_numpy(C0 if cl == sa.COL_MAJOR else np.ascontiguousarray(C0.reshape(n, rows).T).reshape(-1)).cuda()
                d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl, accumulate=acc)
                torch.cuda.synchronize()
                got = Ct.cpu().numpy()
                if cl == sa.ROW_MAJOR:
                    got = np.ascontiguousarray(got.reshape(rows, n).T).reshape(-1)
                _check(got, Co + (C0 if acc else 0), bound + (np.abs(C0) if acc else 0), "window plan %s, c_layout %d acc %d" % (on, cl, acc))
                if cl == sa.COL_MAJOR and not acc:
                    outs.append(got)
        d.close()
    assert np.max(np.abs(outs[0] - outs[1])) <= 1e-5 * np.max(bound)


@pytest.mark.parametrize("dtype", [sa.F32, sa.BF16], ids=["f32", "bf16"])
@pytest.mark.parametrize("win,long_,minseg", [(512, 16, 4), (2048, 64, 16), (64, 8, 1)])
def test_sparse_rows_cut_at_column_windows_and_taken_window_by_window(monkeypatch, _sparse_row_mode, dtype, win, long_, minseg):
    """SPARTA_SP_WINDOW_COLS (automatic on large power-law parts): the long rows' segments end at column-window boundaries and the segment list is processed
    window by window; a row's partial rows are still added in its own segment order.  Same product as the oracle's (every layout, accumulate, host pointers)
    and as the handle without windows up to the rounding of the regrouped sums; rows of every length around the cut (SPARTA_SP_LONG) and the segment minimum."""
    if _sparse_row_mode != "with-sparse-rows":
        pytest.skip("the sparse-row path is what is tested")
    torch = _torch()
    monkeypatch.setenv("SPARTA_SP_WINDOW_COLS", str(win))
    monkeypatch.setenv("SPARTA_SP_LONG", str(long_))
    monkeypatch.setenv("SPARTA_SP_MINSEG", str(minseg))
    rng = np.random.Generator(np.random.PCG64(win + long_))
    rows, cols, w, n = 600, 9000 + 13, 32, 128
    rr, cc = [], []
    for i in range(rows - 10):                                # row lengths 0 .. ~300 around the cut, columns skewed towards the low ones (power law)
        k = int(rng.integers(0, 4 * long_ + 40)) if i % 7 else 0
        c = np.unique(np.minimum((cols * rng.random(k) ** 2.5).astype(np.int64), cols - 1))
        rr.append(np.full(len(c), i)); cc.append(c)
    rr.append(np.full(7000, rows - 10)); cc.append(rng.choice(cols, 7000, replace=False))      # a hub row through every window
    r, c = np.concatenate(rr), np.concatenate(cc)
    order = np.lexsort((c, r))
    r, c = r[order], c[order]
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=rows))])
    m = sa.CSR(rows, cols, rowptr, c.astype(np.int32), rng.uniform(-1, 1, len(c)).astype(np.float32))
    g = np.arange(rows) // 32
    B = sa.gen.dense_rhs(cols, n, seed=31)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    f32 = dtype == sa.F32
    mab_r, B_r = (v.mab, B) if f32 else (_round16(v.mab, dtype), _round16(B, dtype))
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    tdt = torch.float32 if f32 else torch.bfloat16
    ldb = (cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :cols] = torch.from_numpy(B.reshape(n, cols)).cuda().to(tdt)
    outs = []
    # eight streams of windows (one per XCD: sparse_segments_xcd_kernel under a row-major C, the fused small launch under a column-major one), one window-major list, no windows
    for on in ("streams", "one-list", False):
        if on == "one-list":
            monkeypatch.setenv("SPARTA_SP_XCD", "0")
        if not on:
            monkeypatch.setenv("SPARTA_SP_WINDOW_COLS", "0")
        d = sa.DeviceVBS.from_csr(m, g, w, 0, False, device=0, dtype=dtype)
        assert d.sparse_info()["hub_rows"] > (10 if on else 0)
        for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
            for acc in (False, True):
                C0 = sa.gen.dense_rhs(rows, n, seed=32)
                Ct = torch.from_numpy(C0 if cl == sa.COL_MAJOR else np.ascontiguousarray(C0.reshape(n, rows).T).reshape(-1)).cuda()
                d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl, accumulate=acc)
                torch.cuda.synchronize()
                got = Ct.cpu().numpy()
                if cl == sa.ROW_MAJOR:
                    got = np.ascontiguousarray(got.reshape(rows, n).T).reshape(-1)
                _check(got, Co + (C0 if acc else 0), bound + (np.abs(C0) if acc else 0), "sparse rows, windows %s, c_layout %d acc %d" % (on, cl, acc))
                if cl == sa.COL_MAJOR and not acc:
                    outs.append(got)
        d.close()
    assert np.array_equal(outs[0], outs[1]), "the order in which the segments are TAKEN must not change a bit"
    assert np.max(np.abs(outs[0] - outs[2])) <= 1e-5 * np.max(bound)


@pytest.mark.parametrize("dtype", [sa.F32, sa.F16], ids=["f32", "f16"])
@pytest.mark.parametrize("case", ["mixed", "rmat", "rmat-fixed"])
def test_create_from_csr_gives_the_same_product(monkeypatch, _sparse_row_mode, case, dtype):
    """sparta_vbs_create_from_csr never expands the nearly empty block-rows; the product must be BIT-identical to the one of a
    handle made by sparta_vbs_build + sparta_vbs_create (same decisions, same kernels, same data), and equal to the oracle's"""
    torch = _torch()
    monkeypatch.setenv("SPARTA_PATH", "stream")              # two handles compared bit for bit: same MFMA path on both (no autotune)
    monkeypatch.setenv("SPARTA_SPARSE_K_BLOCK", "1e30")      # whole-block-row decisions only, as sparta_vbs_create takes them (the per-block split has its own test below)
    defaults = _sparse_row_mode == "library-defaults"
    if not defaults:
        monkeypatch.setenv("SPARTA_UNION", "0")              # ... and no column-compacted tiles (sparta_vbs_create has no CSR to make them from: tests/test_union_gpu.py)
    n = 128
    if case == "mixed":
        m, w = _mixed_matrix()
        eng, rbs, ff = sa.BlockingEngine(tau=0.5, col_block_size=w), 0, False
    else:
        m, w = sa.gen.rmat(13, 60000, seed=5, symmetrize=True, pattern_only=(case == "rmat")), 32
        ff = case == "rmat-fixed"
        rbs = 24 if ff else 0
        eng = sa.BlockingEngine(blocking_algo=7, tau=0.5, col_block_size=w, row_block_size=max(rbs, 1), force_fixed_size=ff)
    g = eng.GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w, rbs, ff)
    d1 = v.to_device(0, dtype=dtype)
    d2 = sa.DeviceVBS.from_csr(m, g, w, rbs, ff, device=0, dtype=dtype)
    i1, i2 = d1.info(), d2.info()
    # (with the library's own decisions the two builders may differ: the small-matrix rule and the column-compacted tiles belong to the builder that sees the CSR.
    #  Then the two products agree within the tolerance, each with the oracle; with the rules pinned they agree bit for bit)
    if not defaults:
        assert (i1["rows"], i1["cols"], i1["sparse_rows"], i1["tiles16"], i1["tiles32"], i1["tiles64"]) == \
               (i2["rows"], i2["cols"], i2["sparse_rows"], i2["tiles16"], i2["tiles32"], i2["tiles64"])
        assert d1.sparse_info() == d2.sparse_info()
    if _sparse_row_mode == "with-sparse-rows":
        assert i2["sparse_rows"] > 0 and i2["nztot"] < i1["nztot"]               # the dense image shrank
    tdt = {sa.F32: torch.float32, sa.F16: torch.float16}[dtype]
    ldb = (v.cols + 7) // 8 * 8
    B = sa.gen.dense_rhs(v.cols, n, seed=31)
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
        for acc in (False, True):
            C1 = torch.full((v.rows * n,), 0.25, dtype=torch.float32, device="cuda")
            C2 = C1.clone()
            d1.spmm(Bt, C1, n, ldb=ldb, c_layout=cl, accumulate=acc)
            d2.spmm(Bt, C2, n, ldb=ldb, c_layout=cl, accumulate=acc)
            torch.cuda.synchronize()
            if not defaults:
                assert torch.equal(C1, C2), (case, cl, acc)
            elif dtype == sa.F32:
                Cr = _oracle_c(v, B, n, np.full(v.rows * n, 0.25, np.float32) if acc else None)
                bnd = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n) + (0.25 if acc else 0.0)
                for Cx, dd, nm in ((C1, d1, "create"), (C2, d2, "from_csr")):
                    got = Cx.cpu().numpy()
                    got = got if cl == sa.COL_MAJOR else np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
                    _check(got, Cr, bnd, "defaults %s %s" % (nm, case))
                    _carrier(dd, _sparse_row_mode, "%s %s C %s" % (nm, case, "col" if cl == sa.COL_MAJOR else "row"), reference_layouts=(cl == sa.COL_MAJOR))
            else:                                                 # 16-bit: the two handles against each other (different decisions: different rounding points -- fp32 sums of the same rounded products)
                assert torch.allclose(C1, C2, rtol=1e-4, atol=1e-3), (case, cl, acc)
                _carrier(d2, _sparse_row_mode, "from_csr %s f16 C %s" % (case, "col" if cl == sa.COL_MAJOR else "row"), reference_layouts=(cl == sa.COL_MAJOR), fp32=False)
    if dtype == sa.F32:
        Co = _oracle_c(v, B, n)
        C2 = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
        d2.spmm(Bt, C2, n, ldb=ldb)
        torch.cuda.synchronize()
        _check(C2.cpu().numpy(), Co, U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n), "from_csr " + case)
        if i2["sparse_rows"] > 0 or d2.union_info()["area"] > 0:
            with pytest.raises(sa.SpartaError):
                d2.spmm(Bt, C2, n, ldb=ldb, algo=sa.SPMM_EXACT)
    # unsorted columns are refused (the sparse rows are taken as they are)
    if _sparse_row_mode == "with-sparse-rows":
        bad = sa.CSR(2, 8, [0, 2, 3], np.array([5, 1, 2], np.int32), None)
        with pytest.raises(sa.SpartaError):
            sa.DeviceVBS.from_csr(bad, np.array([0, 1]), 4)


@pytest.mark.parametrize("kernel", ["auto", "lds", "direct"])
@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
def test_16bit_gathered_b_and_row_block_exchange(monkeypatch, _sparse_row_mode, dtype, kernel):
    """the multi-GPU entry points with 16-bit storage: sparta_vbs_spmm_gathered on an all-gather-shaped 16-bit B, and the
    row-block exchange (pack kernel + the two products on the row-block-tiled layout) with every rank played on one GPU;
    with the library's choice of 16-bit kernel and with each of the two forced"""
    torch = _torch()
    if kernel != "auto":
        monkeypatch.setenv("SPARTA_H16_PATH", kernel)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    world, w, n = 3, 32, 128
    slabs = [sa.gen.fem3d_slab(4, 4, 9, r, world, dof=3, pad_to=w, seed=4) for r in range(world)]
    n_pad = slabs[0][2]
    shards = [sa.gen.dense_rhs(n_pad, n, seed=50 + r) for r in range(world)]
    gathered = np.concatenate(shards)
    Bfull = sa.dist.gathered_to_colmajor(gathered, world, n_pad, n)
    B_r = _round16(Bfull, dtype)
    tiles = [torch.from_numpy(sa.dist.to_block_tiles(s, n_pad, n, w)).cuda().to(tdt) for s in shards]
    vbs, all_need = [], []
    for m, _, _ in slabs:
        vbs.append(sa.VBR().fill_from_CSR_inplace(m, sa.BlockingEngine(tau=0.4, col_block_size=w).GetGrouping(m), w))
        all_need.append(sa.dist.needed_blocks(vbs[-1].jab, w, n_pad, world))
    exs = [sa.dist.RowBlockExchange(vbs[r], r, world, n_pad, n, device=0, all_need=all_need, dtype=dtype) for r in range(world)]
    tile = w * n
    for r in range(world):
        exs[r]._pack(tiles[r])
    for p in range(world):
        o = 0
        for q in range(world):
            k = exs[p].out_splits[q]
            i0 = sum(exs[q].in_splits[:p])
            exs[p].recv_buf[o:o + k].copy_(exs[q].send_buf[i0:i0 + k])
            o += k
    Bg = torch.from_numpy(gathered).cuda().to(tdt)
    for r in range(world):
        v = vbs[r]
        mab_r = _round16(v.mab, dtype)
        Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
        bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
        d = v.to_device(0, dtype=dtype)
        Ct = torch.full((v.rows * n,), 2.0, dtype=torch.float32, device="cuda")
        d.spmm_gathered(Bg, n_pad, Ct, n)
        torch.cuda.synchronize()
        _check(Ct.cpu().numpy(), Co, bound, "16-bit gathered B rank %d" % r)
        Ct.fill_(5.0)
        exs[r]._product("own", tiles[r], Ct, False)
        exs[r]._product("remote", exs[r].recv_buf, Ct, True)
        torch.cuda.synchronize()
        _check(Ct.cpu().numpy(), Co, bound, "16-bit row-block exchange rank %d" % r)
        exs[r].close()


@pytest.mark.parametrize("dtype", [sa.F32, sa.BF16], ids=["f32", "bf16"])
def test_few_sparse_rows_read_a_column_major_b_in_place(_sparse_row_mode, dtype):
    """a handful of sparse rows next to a large dense part: transposing all of B for them would cost more than the rows
    themselves, so a column-major (or gathered) B is gathered in place"""
    torch = _torch()
    rng = np.random.Generator(np.random.PCG64(9))
    rows, cols, w, n = 200, 6400, 32, 128
    rr, cc = [], []
    for i in range(192):                                      # dense part
        c = rng.choice(640, 300, replace=False)
        rr.append(np.full(300, i)); cc.append(c)
    for i in range(192, 197):                                 # five scattered rows (rows 197..199 empty)
        c = rng.choice(cols, 4, replace=False)
        rr.append(np.full(4, i)); cc.append(c)
    r, c = np.concatenate(rr), np.concatenate(cc)
    o = np.lexsort((c, r)); r, c = r[o], c[o]
    m = sa.CSR(rows, cols, np.concatenate([[0], np.cumsum(np.bincount(r, minlength=rows))]), c.astype(np.int32), rng.uniform(-1, 1, len(c)).astype(np.float32))
    g = np.concatenate([np.arange(192) // 32, np.arange(192, 200)])
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=3)
    tdt = {sa.F32: torch.float32, sa.BF16: torch.bfloat16}[dtype]
    mab_r, B_r = (v.mab, B) if dtype == sa.F32 else (_round16(v.mab, dtype), _round16(B, dtype))
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    d = v.to_device(0, dtype=dtype)
    if _sparse_row_mode == "with-sparse-rows":
        sp = d.sparse_info()
        assert 0 < sp["rows"] <= 8 and sp["nnz"] * 8 < v.cols
    Bt = torch.from_numpy(B).cuda().to(tdt)
    for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
        Ct = torch.full((v.rows * n,), 9.0, dtype=torch.float32, device="cuda")
        d.spmm(Bt, Ct, n, c_layout=cl)
        torch.cuda.synchronize()
        got = Ct.cpu().numpy()
        if cl == sa.ROW_MAJOR:
            got = np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
        _check(got, Co, bound, "few sparse rows, c_layout %d" % cl)
    # gathered B (two slabs of 3200 rows)
    Bg = torch.from_numpy(np.ascontiguousarray(B.reshape(n, 2, 3200).transpose(1, 0, 2)).reshape(-1)).cuda().to(tdt)
    Ct = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
    d.spmm_gathered(Bg, 3200, Ct, n)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), Co, bound, "few sparse rows, gathered B")


@pytest.mark.parametrize("M", [1, 37, 128, 256])
def test_dense_times_vbs_product(_sparse_row_mode, M):
    """C += B * A (dense x VBS) through the handle of A^T: against a float64 product of B with the dense form of the VBS"""
    rng = np.random.Generator(np.random.PCG64(M))
    m = sa.gen.uniform_random(700, 520, 9000, seed=12)
    w = 16
    g = sa.BlockingEngine(tau=0.5, col_block_size=w).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    perm = sa.get_permutation(g)
    A = np.zeros((m.rows, m.cols), np.float64)
    for i in range(m.rows):
        A[i, m.colidx[m.rowptr[i]:m.rowptr[i + 1]]] = m.vals[m.rowptr[i]:m.rowptr[i + 1]]
    A = A[perm]                                              # rows in the VBS's order
    B = rng.uniform(-1, 1, (M, v.rows)).astype(np.float32)
    C0 = rng.uniform(-1, 1, (M, v.cols)).astype(np.float32)
    want = C0.astype(np.float64) + B.astype(np.float64) @ A
    scale = np.abs(C0).astype(np.float64) + np.abs(B).astype(np.float64) @ np.abs(A)
    Cc = np.ascontiguousarray(C0.T).reshape(-1)              # column-major M x cols
    dt = v.multiply_BA(np.ascontiguousarray(B.T).reshape(-1), M, Cc)
    got = Cc.reshape(v.cols, M).T
    assert dt >= 0 and np.all(np.abs(got - want) <= 1e-5 * scale + 1e-30)
    # a second call accumulates again
    v.multiply_BA(np.ascontiguousarray(B.T).reshape(-1), M, Cc)
    got2 = Cc.reshape(v.cols, M).T
    assert np.all(np.abs(got2 - (want + B.astype(np.float64) @ A)) <= 2e-5 * scale + 1e-30)


@pytest.mark.parametrize("dtype", [sa.F32, sa.F16], ids=["f32", "f16"])
@pytest.mark.parametrize("first_empty,n_empty", [(640, 4100), (0, 2048), (1001, 2500)])
def test_long_run_of_empty_rows_is_zero_filled_by_the_streamed_fill(monkeypatch, dtype, first_empty, n_empty):
    """A block-row without blocks that is thousands of rows tall (what the empty rows of a power-law matrix become after clustering)
    is zero-filled by `vbs_zero_rows_kernel` instead of 64-row fix-up tiles: C = 0 there under accumulate = 0 (also at a row offset
    that is not 16-byte aligned, and in both layouts of C), untouched under accumulate = 1; the rows around it are the oracle's."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_PATH", "stream")
    rows, cols, w, n = first_empty + n_empty + 700, 1024, 32, 128
    m0 = sa.gen.uniform_random(rows, cols, 40000, seed=77)
    keep = np.ones(rows, bool); keep[first_empty:first_empty + n_empty] = False
    cnt = np.diff(m0.rowptr) * keep
    sel = np.repeat(keep, np.diff(m0.rowptr))
    m = sa.CSR(rows, cols, np.concatenate([[0], np.cumsum(cnt)]), m0.colidx[sel], m0.vals[sel])
    g = np.empty(rows, np.int64)
    g[:first_empty] = np.arange(first_empty) // 32
    g[first_empty:first_empty + n_empty] = 10 ** 6                   # the empty rows: ONE group, starting at row `first_empty` of the permuted C
    g[first_empty + n_empty:] = 10 ** 6 + 1 + np.arange(rows - first_empty - n_empty) // 32
    g = np.unique(g, return_inverse=True)[1]
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    h = np.diff(v.row_part)
    assert h.max() == n_empty and v.row_part[int(np.argmax(h))] == first_empty
    B = sa.gen.dense_rhs(v.cols, n, seed=5)
    if dtype == sa.F16:
        mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    else:
        mab_r, B_r = v.mab, B
    Co = O.vbr_multiply(v.rows, v.cols, v.block_col_size, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    d = v.to_device(0, dtype=dtype)
    if dtype == sa.F16:
        ldb = (v.cols + 7) // 8 * 8
        Bt = torch.zeros(ldb * n, dtype=torch.float16, device="cuda")
        Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(torch.float16)
    else:
        ldb, Bt = v.cols, torch.from_numpy(B).cuda()
    for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
        Ct = torch.full((v.rows * n,), 7.0, dtype=torch.float32, device="cuda")
        d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl)
        torch.cuda.synchronize()
        got = Ct.cpu().numpy()
        if cl == sa.ROW_MAJOR:
            got = np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
        _check(got, Co, bound, "long empty block-row, c_layout %d" % cl)
        ib = int(np.argmax(h))
        assert v.nzcount[ib] == 0
        assert not got.reshape(n, v.rows)[:, v.row_part[ib]:v.row_part[ib + 1]].any()
    C0 = sa.gen.dense_rhs(v.rows, n, seed=11)
    Ct = torch.from_numpy(C0).cuda()
    d.spmm(Bt, Ct, n, ldb=ldb, accumulate=True)
    torch.cuda.synchronize()
    _check(Ct.cpu().numpy(), C0 + Co, bound + np.abs(C0), "long empty block-row, accumulate")


@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
def test_full_size_16bit_flagship_kernels_agree_and_match_the_oracle(monkeypatch, dtype):
    """The configuration `bench.py --dtype f16|bf16` times (cant-like 62 451^2, Keeper tau 0.6, 32 x 32 blocks, N = 128) at full size:
    (i) the two 16-bit kernels (LDS-staged, direct-to-register) and the library's own choice give the same bits, (ii) two runs are
    bit-identical, (iii) sampled block-rows match the oracle on the ROUNDED inputs within the fp32 bound, (iv) the column checksum
    1^T C == (1^T A_rounded) B_rounded."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_PATH", "stream")
    m = sa.gen.cant_like(seed=2)
    w, n = 32, 128
    g = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=32, force_fixed_size=True, sim_measure=1).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w, 32, True)
    d = v.to_device(0, dtype=dtype)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    B = sa.gen.dense_rhs(v.cols, n, seed=1)
    ldb = (v.cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)

    def run(kernel):
        if kernel is None:
            monkeypatch.delenv("SPARTA_H16_PATH", raising=False)
        else:
            monkeypatch.setenv("SPARTA_H16_PATH", kernel)
        Ct = torch.full((v.rows * n,), 3.0, dtype=torch.float32, device="cuda")
        d.spmm(Bt, Ct, n, ldb=ldb, accumulate=False)
        torch.cuda.synchronize()
        return Ct
    C_auto, C_lds, C_dir = run(None), run("lds"), run("direct")
    assert torch.equal(C_auto, C_lds) and torch.equal(C_auto, C_dir)           # (i)
    assert torch.equal(C_auto, run(None))                                      # (ii)
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Ch = C_auto.cpu().numpy().reshape(n, v.rows)
    absA, absB = np.abs(mab_r), np.abs(B_r)
    rng = np.random.Generator(np.random.PCG64(1))
    for ib in rng.choice(v.block_rows, size=12, replace=False):               # (iii)
        r0, r1 = int(v.row_part[ib]), int(v.row_part[ib + 1])
        Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, block_row_range=(int(ib), int(ib) + 1)).reshape(n, v.rows)
        bd = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, absA, absB, n, block_row_range=(int(ib), int(ib) + 1)).reshape(n, v.rows)
        _check(Ch[:, r0:r1], Co[:, r0:r1], bd[:, r0:r1], "16-bit flagship block-row %d" % ib)
    # (iv): 1^T A over the stored (rounded) blocks, column-block by column-block
    colsum = np.zeros((v.cols + w - 1) // w * w, np.float64)
    starts = np.concatenate([[0], np.cumsum(v.nzcount)]).astype(np.int64)
    hs = np.diff(v.row_part).astype(np.int64)
    pos = 0
    for ib in range(v.block_rows):
        hgt = int(hs[ib])
        for b in range(int(v.nzcount[ib])):
            jb = int(v.jab[starts[ib] + b])
            blk = mab_r[pos:pos + hgt * w].astype(np.float64).reshape(w, hgt)   # column-major h x w block: [k][row]
            colsum[jb * w:(jb + 1) * w] += blk.sum(axis=1)
            pos += hgt * w
    want = colsum[:v.cols] @ B_r.astype(np.float64).reshape(n, v.cols).T
    got = Ch.astype(np.float64).sum(axis=1)
    assert np.allclose(got, want, rtol=0, atol=1e-3 * max(1.0, float(np.abs(want).max())))


def test_a_handful_of_nearly_empty_block_rows_stays_with_the_tiles(monkeypatch, _sparse_row_mode):
    """the sparse-row kernels are extra launches: nearly empty block-rows leave the MFMA tiles only when together they are worth
    SPARTA_SPARSE_MIN_STEPS steps (default 4096); the product is the oracle's either way (both builders: VBS arrays and from_csr)"""
    if _sparse_row_mode == "mfma-only":
        pytest.skip("one mode is enough")
    torch = _torch()
    m, w = _mixed_matrix()
    g = sa.BlockingEngine(tau=0.5, col_block_size=w).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    n = 128
    B = sa.gen.dense_rhs(v.cols, n, seed=21)
    Co = _oracle_c(v, B, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    Bt = torch.from_numpy(B).cuda()
    for min_steps, want_sparse in (("0", True), (None, False)):
        if min_steps is None:
            monkeypatch.delenv("SPARTA_SPARSE_MIN_STEPS", raising=False)
        else:
            monkeypatch.setenv("SPARTA_SPARSE_MIN_STEPS", min_steps)
        for d in (v.to_device(0), sa.DeviceVBS.from_csr(m, g, w, device=0)):
            assert (d.info()["sparse_rows"] > 0) == want_sparse
            Ct = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
            d.spmm(Bt, Ct, n)
            torch.cuda.synchronize()
            _check(Ct.cpu().numpy(), Co, bound, "min steps %s" % min_steps)
            d.close()


@pytest.mark.parametrize("dtype", [sa.F32, sa.BF16], ids=["f32", "bf16"])
@pytest.mark.parametrize("kblock", ["2", "8", None])
def test_create_from_csr_splits_a_block_row_into_tiles_and_sparse_rows(monkeypatch, _sparse_row_mode, dtype, kblock):
    """sparta_vbs_create_from_csr decides per BLOCK: the well-filled blocks of a block-row stay MFMA tiles, the nonzeros of its other
    blocks become sparse rows that ADD to what the tiles stored (dense power-law matrices: hub columns fill their blocks, the tail does
    not).  Product against the oracle in both layouts of C, with and without accumulate, for several thresholds."""
    if _sparse_row_mode == "mfma-only":
        pytest.skip("needs the sparse-row path")
    torch = _torch()
    if kblock is None:
        monkeypatch.delenv("SPARTA_SPARSE_K_BLOCK", raising=False)
    else:
        monkeypatch.setenv("SPARTA_SPARSE_K_BLOCK", kblock)
    monkeypatch.setenv("SPARTA_UNION", "0")                   # (the subject is the per-block split; the column-compacted tiles would take these block-rows: tests/test_union_gpu.py)
    rng = np.random.Generator(np.random.PCG64(77))
    rows, cols, w, n = 640, 4096 + 17, 32, 128
    rr, cc = [], []
    for i in range(rows):                                     # every row: a dense stripe in the first 96 columns (3 well-filled blocks) + a thin random tail
        c = np.unique(np.concatenate([rng.choice(96, 40, replace=False), 96 + rng.choice(cols - 96, int(rng.integers(3, 30)), replace=False)]))
        rr.append(np.full(len(c), i)); cc.append(c)
    r, c = np.concatenate(rr), np.concatenate(cc)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=rows))])
    m = sa.CSR(rows, cols, rowptr, c.astype(np.int32), rng.uniform(-1, 1, len(c)).astype(np.float32))
    g = sa.BlockingEngine(tau=0.5, col_block_size=w, blocking_algo=7).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    d = sa.DeviceVBS.from_csr(m, g, w, device=0, dtype=dtype)
    info, sp = d.info(), d.sparse_info()
    if kblock is not None:                                    # at these thresholds the matrix must actually be split: tiles AND sparse rows on the same block-rows
        assert info["tiles16"] + info["tiles32"] + info["tiles64"] > 0 and 0 < sp["nnz"] < m.nztot()
    tdt = {sa.F32: torch.float32, sa.BF16: torch.bfloat16}[dtype]
    ldb = (v.cols + 7) // 8 * 8
    B = sa.gen.dense_rhs(v.cols, n, seed=41)
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    if dtype == sa.F32:
        vo, Bo = v, B
    else:                                                     # the oracle on the rounded inputs (products of two 16-bit values are exact in fp32)
        rnd = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(tdt).float().numpy()
        Bo = rnd(B)
        vo = sa.VBR.from_arrays(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, rnd(v.mab))
    for cl in (sa.COL_MAJOR, sa.ROW_MAJOR):
        for acc in (False, True):
            C0 = sa.gen.dense_rhs(v.rows, n, seed=42)
            Ct = torch.from_numpy(C0 if cl == sa.COL_MAJOR else np.ascontiguousarray(C0.reshape(n, v.rows).T).reshape(-1)).cuda()
            d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl, accumulate=acc)
            torch.cuda.synchronize()
            got = Ct.cpu().numpy()
            if cl == sa.ROW_MAJOR:
                got = np.ascontiguousarray(got.reshape(v.rows, n).T).reshape(-1)
            want = _oracle_c(vo, Bo, n, C0) if acc else _oracle_c(vo, Bo, n)
            bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, vo.mab, Bo, n)
            _check(got, want, bound + (np.abs(C0) if acc else 0), "per-block split kblock=%s layout %d acc %d" % (kblock, cl, acc))
    d.close()



def test_product_inside_a_graph_capture(monkeypatch, _sparse_row_mode):
    """a shape that has run once is kernel launches only and can be captured into a graph (replay = the eager product, bit for bit);
    a shape that would still have to allocate scratch or time its paths refuses to run under capture instead of breaking it"""
    if _sparse_row_mode == "with-sparse-rows":
        pytest.skip("one mode is enough")
    torch = _torch()
    monkeypatch.delenv("SPARTA_PATH", raising=False)
    m = sa.gen.fem3d(6, 6, 12, 3, 1)
    w, n = 32, 128
    eng = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=32, force_fixed_size=True)
    v = sa.VBR().fill_from_CSR_inplace(m, eng.GetGrouping(m), w, 32, True)
    d = v.to_device(0)
    B = torch.from_numpy(sa.gen.dense_rhs(v.cols, n, seed=5)).cuda()
    C_eager = torch.zeros(v.rows * n, dtype=torch.float32, device="cuda")
    C_graph = torch.zeros_like(C_eager)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(gph, stream=s):            # first call of this shape: autotune / scratch -> refused, the capture survives
                with pytest.raises(sa.SpartaError):
                    d.spmm(B, C_graph, n)
        except RuntimeError:
            pass                                             # (an empty capture may itself be rejected by the runtime: not the subject)
        d.spmm(B, C_eager, n)                                # once outside a capture
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            d.spmm(B, C_graph, n)
    torch.cuda.synchronize()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(C_graph, C_eager)
    d.close()


@pytest.mark.parametrize("align", ["aligned", "split"])
@pytest.mark.parametrize("n", [128, 384])
def test_tiles_parked_in_the_lds_ring_store_the_same_c(monkeypatch, _sparse_row_mode, align, n):
    """fp32, column-major C, tiles of arbitrary height: the no-barrier kernel parks finished tiles in a wave-private LDS ring and stores aligned
    blocks of 32 rows (k_f32_direct.hip, CSTAGE).  Same sums, other stores: C must be bit-identical to the direct-store form of the same kernel
    (SPARTA_CSTAGE=0) for whole-tile plans, within the tolerance for split plans, with and without accumulate, around gaps (empty block-rows, sparse rows, split tiles),
    and equal to the oracle's."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_PATH", "stream")
    monkeypatch.setenv("SPARTA_STREAM_ALIGN", "1" if align == "aligned" else "0")
    rng = np.random.Generator(np.random.PCG64(123))
    rows, cols, w = 3000, 2048 + 5, 32
    rr, cc = [], []
    for i in range(rows):                                     # a band with holes: runs of similar rows (tiles of 5..30 rows), every 40th row empty, a few scattered singletons
        if i % 40 == 39:
            continue
        c0 = (i // 7) * 3 % (cols - 200)
        c = np.unique(np.concatenate([c0 + rng.choice(160, 60, replace=False), rng.choice(cols, 1)]))
        rr.append(np.full(len(c), i)); cc.append(c)
    r, c = np.concatenate(rr), np.concatenate(cc)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=rows))])
    m = sa.CSR(rows, cols, rowptr, c.astype(np.int32), rng.uniform(-1, 1, len(c)).astype(np.float32))
    g = sa.BlockingEngine(tau=0.6, col_block_size=w, blocking_algo=7, minhash_max_rows=32).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    heights = np.diff(v.row_part)
    assert len(np.unique(heights)) > 3 and np.any(np.cumsum(heights)[:-1] % 32 != 0)      # tiles of several heights, not all starting on multiples of 32
    B = sa.gen.dense_rhs(v.cols, n, seed=51)
    Bt = torch.from_numpy(B).cuda()
    Co = _oracle_c(v, B, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    got = {}
    for ring in ("1", "0"):
        monkeypatch.setenv("SPARTA_CSTAGE", ring)          # read at create time (plan: contiguous ranges) and per process at the first launch
        d = v.to_device(0)
        for acc in (False, True):
            C0 = sa.gen.dense_rhs(v.rows, n, seed=52)
            Ct = torch.from_numpy(C0).cuda()
            d.spmm(Bt, Ct, n, accumulate=acc)
            torch.cuda.synchronize()
            got[(ring, acc)] = Ct.cpu().numpy()
            want = _oracle_c(v, B, n, C0) if acc else Co
            _check(got[(ring, acc)], want, bound + (np.abs(C0) if acc else 0), "ring %s acc %d" % (ring, acc))
        d.close()
    if align == "aligned":                                    # whole tiles: the same sums in the same order, whatever the stores do
        for acc in (False, True):
            assert np.array_equal(got[("1", acc)], got[("0", acc)])


@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
def test_16bit_handles_take_leading_dimensions_of_2_to_the_23(monkeypatch, _sparse_row_mode, dtype):
    """configs[4] on one GPU has a column-major B with 2^23 rows: 128 columns of it are 2 GB, beyond a 32-bit byte offset.  The 16-bit
    no-barrier kernel folds its wave's column base into the scalar descriptor, so only 32 columns have to fit (ldb < 34 M, ldc < 17 M); the
    LDS-staged kernel keeps the limit of the 128-column slab and says so."""
    torch = _torch()
    m = sa.gen.uniform_random(700, 900, 30000, seed=33)
    v = sa.VBR().fill_from_CSR_inplace_fixed(m, 32, 64)
    n, ldb, ldc = 128, (1 << 23) + 8, 4_400_000
    tdt = {sa.F16: torch.float16, sa.BF16: torch.bfloat16}[dtype]
    B = sa.gen.dense_rhs(v.cols, n, seed=2)
    rnd = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(tdt).float().numpy()
    vo = sa.VBR.from_arrays(v.rows, v.cols, 64, v.row_part, v.nzcount, v.jab, rnd(v.mab))
    Bo = rnd(B)
    Co = _oracle_c(vo, Bo, n)
    bound = U.abs_bound(v.rows, v.cols, 64, v.row_part, v.nzcount, v.jab, vo.mab, Bo, n)
    d = v.to_device(0, dtype=dtype)
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    Ct = torch.full((ldc * n,), 3.0, dtype=torch.float32, device="cuda")
    d.spmm(Bt, Ct, n, ldb=ldb, ldc=ldc)
    torch.cuda.synchronize()
    got = Ct.view(n, ldc)[:, :v.rows].contiguous().view(-1).cpu().numpy()
    _check(got, Co, bound, "16-bit, ldb 2^23")
    assert float(Ct.view(n, ldc)[:, v.rows:v.rows + 8].min()) == 3.0          # nothing written outside the rows of C
    monkeypatch.setenv("SPARTA_H16_PATH", "lds")
    with pytest.raises(sa.SpartaError):
        d.spmm(Bt, Ct, n, ldb=ldb, ldc=ldc)
    del Bt, Ct
    torch.cuda.empty_cache()


@pytest.mark.parametrize("sched", ["cut-by-steps", "dealt", "consecutive-units"])
@pytest.mark.parametrize("G", ["2", "4"])
@pytest.mark.parametrize("dtype", [sa.F16, sa.BF16], ids=["f16", "bf16"])
@pytest.mark.parametrize("rows,cols,nnz,n,drop", [(640, 6400, 260000, 256, 0.0), (1000, 8191, 800000, 384, 0.3), (300, 12800, 400000, 512, 0.15), (832, 4096, 300000, 200, 0.5)])
def test_16bit_hub_group_tiles_against_the_oracle(monkeypatch, _sparse_row_mode, sched, G, dtype, rows, cols, nnz, n, drop):
    """The hub plan of 16-bit handles of 64-wide blocks (vbs_plan.cpp, k_hub16.hip): long tiles of 33..64 rows grouped by the Jaccard similarity of their block
    columns into group tiles of G, multiplied by the GEMM-shaped kernel over the UNION of the group's block columns -- a member without a block in a column is a
    slice that is not fetched.  Against the oracle's product on the rounded inputs: groups of 2 and of 4 (and the short groups a count not divisible by G leaves),
    blocks missing from some members (`drop`: whole blocks removed at random, so that the unions differ from their members), a partial last block-row (rows % 64),
    a partial last block column (cols % 64: B_tail), N % 256 == 128 (a half slab), N % 128 != 0 (tail slab), both layouts of C, accumulate, a gathered B, split
    segments (every plan here has them: the step list is ordered by K range) -- and the plan with the hub switched off gives the same product within the tolerance."""
    torch = _torch()
    monkeypatch.setenv("SPARTA_HUB_G", G)
    monkeypatch.setenv("SPARTA_HUB_MIN_TOTAL", "0")
    monkeypatch.setenv("SPARTA_HUB_MIN_STEPS", "16")
    # the three schedules of the step list (vbs_plan.cpp, HUB PLAN part 2): four K ranges cut by step count anywhere (split segments inside units), whole (K chunk, group)
    # units dealt round-robin to the workers (SPARTA_HUB_DEAL=1), consecutive whole units per worker (the default)
    if sched == "cut-by-steps":
        monkeypatch.setenv("SPARTA_HUB_RANGES", "4")
    monkeypatch.setenv("SPARTA_HUB_DEAL", "1" if sched == "dealt" else "0")
    if sched != "cut-by-steps" and _sparse_row_mode != "mfma-only":
        pytest.skip("the schedules differ in the hub plan only: one sparse-row mode is enough")
    monkeypatch.setenv("SPARTA_HUB_TAU", "0.25")        # (random halves of the block columns are a third alike)
    w = 64
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols + n)
    if drop > 0.0:                                       # remove whole 64 x 64 blocks: the members of a group then own different block columns
        rng = np.random.Generator(np.random.PCG64(rows + n))
        keep_blk = rng.random(((rows + 63) // 64, (cols + 63) // 64)) >= drop
        r_of = np.repeat(np.arange(m.rows), np.diff(m.rowptr))
        keep = keep_blk[r_of // 64, m.colidx // 64]
        rowptr = np.zeros(m.rows + 1, np.int64)
        np.add.at(rowptr, r_of[keep] + 1, 1)
        m = sa.CSR(m.rows, m.cols, np.cumsum(rowptr), m.colidx[keep].copy(), m.vals[keep].copy())
    g = np.arange(rows) // 64
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    B = sa.gen.dense_rhs(v.cols, n, seed=5)
    tdt = torch.float16 if dtype == sa.F16 else torch.bfloat16
    ldb = (v.cols + 7) // 8 * 8
    Bt = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    Bt.view(n, ldb)[:, :v.cols] = torch.from_numpy(B.reshape(n, v.cols)).cuda().to(tdt)
    world = 2
    shard_rows = v.cols // world if v.cols % (world * w) == 0 else 0
    Bg = None
    if shard_rows:
        Bg = torch.zeros(world * shard_rows * n, dtype=tdt, device="cuda")
        for s_ in range(world):
            Bg[s_ * shard_rows * n:(s_ + 1) * shard_rows * n].view(n, shard_rows)[:] = Bt.view(n, ldb)[:, s_ * shard_rows:(s_ + 1) * shard_rows]
    outs = {}
    for hub in ("1", "0"):
        monkeypatch.setenv("SPARTA_HUB", hub)
        d = v.to_device(0, dtype=dtype)
        hi = d.hub_info()
        if hub == "1":
            assert hi["steps"] > 0 and hi["tiles_per_group"] == int(G) and hi["groups"] >= 1 and hi["tiles"] >= 2 * hi["groups"], hi
            assert hi["union_area"] >= hi["stored_area"] > 0
            if drop == 0.0 and rows % 64 == 0 and _sparse_row_mode == "mfma-only":
                assert hi["tiles"] >= rows // 64 - 1, hi       # identical block columns: every tile finds a partner (one may be left over when G does not divide)
        else:
            assert hi["steps"] == 0
        res = []
        for cl, acc in ((sa.COL_MAJOR, False), (sa.ROW_MAJOR, False), (sa.COL_MAJOR, True)):
            Ct = torch.full((v.rows * n,), 0.5, dtype=torch.float32, device="cuda")
            d.spmm(Bt, Ct, n, ldb=ldb, c_layout=cl, accumulate=acc)
            torch.cuda.synchronize()
            res.append(Ct.cpu().numpy())
        if Bg is not None:
            Ct = torch.full((v.rows * n,), 0.5, dtype=torch.float32, device="cuda")
            d.spmm_gathered(Bg, shard_rows, Ct, n)
            torch.cuda.synchronize()
            res.append(Ct.cpu().numpy())
        # twice the same bits (no atomics, fixed order of the partial images)
        Ct = torch.full((v.rows * n,), 0.5, dtype=torch.float32, device="cuda")
        d.spmm(Bt, Ct, n, ldb=ldb)
        torch.cuda.synchronize()
        assert np.array_equal(Ct.cpu().numpy(), res[0])
        outs[hub] = res
        d.close()
    mab_r, B_r = _round16(v.mab, dtype), _round16(B, dtype)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n, None)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, n)
    for hub in ("1", "0"):
        o = outs[hub]
        _check(o[0], Co, bound, "hub %s, column-major C" % hub)
        _check(np.ascontiguousarray(o[1].reshape(v.rows, n).T).reshape(-1), Co, bound, "hub %s, row-major C" % hub)
        _check(o[2], Co + 0.5, bound + 0.5, "hub %s, accumulate" % hub)
        if Bg is not None:
            _check(o[3], Co, bound, "hub %s, gathered B" % hub)


def test_zz_library_defaults_report():
    """(last in the file) what carried the products of the library-defaults runs above: printed into the log (pytest -s / the captured output of a failure) and
    checked for coverage -- every kind of carrier must have been exercised by the library's OWN decisions at least once"""
    if not CARRIED:
        pytest.skip("the library-defaults runs were deselected")
    for k in sorted(CARRIED):
        print("library-defaults | %-44s | %s" % (k, CARRIED[k]))
    kinds = {"tiles": any(r["tiles"] > 0 for r in CARRIED.values()), "sparse rows": any(r["sparse_rows"] > 0 for r in CARRIED.values()),
             "resident columns": any(r["resident_columns"] > 0 for r in CARRIED.values())}
    assert all(kinds.values()), kinds
