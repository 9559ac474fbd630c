"""BASELINE.json configs at their stated sizes on one MI355X (-m gpu): one test per config / arm.

The oracle (the reference's CPU loops restated) cannot multiply 10^8-nonzero matrices in test time, so each test combines
  (i)   rows / block-rows of C sampled at random against a float64 evaluation of the same rows (values of A and B rounded to the
        storage type first: products of two 16-bit values are exact in fp32, so the fp32 tolerance 1e-5 * sum|a||b| applies unchanged),
  (ii)  the column checksum 1^T C == (1^T A) B  (every row of C takes part),
  (iii) bit-reproducibility of a second product into a dirty buffer,
  (iv)  where two groupings of the same matrix exist (reorder on / reorder off -- the experiment the reference runs,
        src/scripts/run_multiplication_experiments_fixed_cluster.sh:14-16): the two products agree row for row through the permutations.
configs[1] (the flagship blocking of bench.py) is small enough for the oracle's VBR::multiply on EVERY row.
Each test stays under about a minute on the GPU box (generation on the GPU, reorder + build on the host)."""
import numpy as np
import pytest

import sparta_amd as sa
from oracle import oracle as O
import _util as U

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _rounded(x, torch, tdt):
    """fp32 numpy -> the values the device multiplies (rounded to the storage type), as float64"""
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(tdt).float().numpy().astype(np.float64)


def _sampled_rows_check(torch, m, grouping, C, n, B, ldb, tdt, n_rows=96, seed=1):
    """rows of C (reordered order) against float64 sums over the row's nonzeros; returns the worst |err| / sum|a||b|"""
    perm = sa.get_permutation(grouping)
    rng = np.random.Generator(np.random.PCG64(seed))
    deg = np.diff(m.rowptr)[perm]
    # a mix: uniformly random rows + the heaviest rows (hubs take the segmented path)
    pick = np.unique(np.concatenate([rng.integers(0, m.rows, n_rows), np.argsort(deg)[-8:]]))
    Cv, Bv = C.view(n, -1), B.view(n, ldb)
    worst = 0.0
    for r in pick:
        i = perm[r]
        cols_i = m.colidx[m.rowptr[i]:m.rowptr[i + 1]]
        got = Cv[:, int(r)].cpu().numpy().astype(np.float64)
        if len(cols_i) == 0:
            worst = max(worst, float(np.abs(got).max()))
            continue
        a = m.vals[m.rowptr[i]:m.rowptr[i + 1]] if m.vals is not None else np.ones(len(cols_i), np.float32)
        a = _rounded(a, torch, tdt)
        bb = Bv[:, torch.from_numpy(cols_i.astype(np.int64)).cuda()].float().cpu().numpy().astype(np.float64)
        want, scale = bb @ a, np.abs(bb) @ np.abs(a) + 1e-30
        worst = max(worst, float((np.abs(got - want) / scale).max()))
    return worst


def _column_checksum_check(torch, m, C, n, B, ldb, tdt):
    """1^T C == (1^T A) B per column of B, in float64 on the device; returns max |diff| / (|1^T A| |B|)"""
    # column sums of (rounded) A and of |A|, accumulated on the device in float64, a chunk of the nonzeros at a time (the hub parts hold 1e9..2.5e9 of them)
    st = torch.zeros(m.cols, dtype=torch.float64, device="cuda")
    sat = torch.zeros(m.cols, dtype=torch.float64, device="cuda")
    nnz, step = m.nztot(), 1 << 27
    for lo in range(0, nnz, step):
        hi = min(nnz, lo + step)
        idx = torch.from_numpy(m.colidx[lo:hi].astype(np.int64)).cuda()
        v = torch.ones(hi - lo, dtype=torch.float64, device="cuda") if m.vals is None else torch.from_numpy(m.vals[lo:hi]).cuda().to(tdt).double()
        st.index_add_(0, idx, v)
        sat.index_add_(0, idx, v.abs())
        del idx, v
    Bd = B.view(n, ldb)[:, :m.cols].double()
    want, scale = Bd @ st, Bd.abs() @ sat + 1e-30
    got = C.view(n, -1).double().sum(dim=1)
    return float(((got - want).abs() / scale).max())


def _dense_rhs(torch, cols, n, tdt, seed):
    ldb = (cols + 7) // 8 * 8 if tdt != torch.float32 else cols
    g = torch.Generator(device="cuda").manual_seed(seed)
    B = torch.zeros(ldb * n, dtype=tdt, device="cuda")
    B.view(n, ldb)[:, :cols] = (torch.rand(n, cols, generator=g, device="cuda") - 0.5).to(tdt)
    return B, ldb


def _product(torch, d, B, ldb, n, fill=None):
    C = torch.empty(d.rows * n, dtype=torch.float32, device="cuda")
    if fill is not None:
        C.fill_(fill)
    d.spmm(B, C, n, accumulate=False, ldb=ldb)
    torch.cuda.synchronize()
    return C


def test_config1_flagship_blocking_every_row_against_the_oracle():
    """configs[1] exactly as bench.py runs it: cant-like 62 451^2, Keeper tau 0.6, 32-row blocks (-a 5 -B 32 -F 1), w = 32, N = 128 fp32,
    column-major B and C: EVERY element of C against the oracle's VBR::multiply within 1e-5 * sum|a||b|, for the product kernel the
    library picks (no-barrier) and for the LDS-staged one it can be told to use; the exact-order kernel bit for bit."""
    import os
    torch = _torch()
    m = sa.gen.cant_like(seed=2)
    w, n = 32, 128
    eng = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=32, force_fixed_size=True, sim_measure=1)
    g = eng.GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w, 32, True)
    assert (v.block_rows, len(v.jab), int(v.nztot)) == (1952, 21910, 22435840)        # the workload bench.py reports
    Bh = sa.gen.dense_rhs(v.cols, n, seed=9)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bh, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bh, n)
    B = torch.from_numpy(Bh).cuda()
    saved = os.environ.get("SPARTA_F32_PLAN")
    try:
        for plan in (None, "legacy"):
            if plan is None:
                os.environ.pop("SPARTA_F32_PLAN", None)
            else:
                os.environ["SPARTA_F32_PLAN"] = plan
            d = v.to_device(0)
            C1 = _product(torch, d, B, v.cols, n, fill=7.0)
            C2 = _product(torch, d, B, v.cols, n, fill=-3.0)
            assert torch.equal(C1, C2), "not reproducible (plan %s)" % plan
            err = np.abs(C1.cpu().numpy() - Co)
            assert np.all(err <= 1e-5 * bound + 1e-30), (plan, float((err / (bound + 1e-30)).max()))
            if plan is None:
                Ce = torch.zeros_like(C1)
                d.spmm(B, Ce, n, accumulate=False, algo=sa.SPMM_EXACT)
                torch.cuda.synchronize()
                assert np.array_equal(Ce.cpu().numpy(), Co), "exact-order kernel is not bit-identical to the oracle"
            d.close()
    finally:
        if saved is None:
            os.environ.pop("SPARTA_F32_PLAN", None)
        else:
            os.environ["SPARTA_F32_PLAN"] = saved


@pytest.mark.parametrize("arm", ["reorder-on", "reorder-off"])
def test_config2_ogbn_products_sized_fp16(arm):
    """configs[2]: 2 449 029^2, ~124 M nonzeros (values 1), B = 256 columns, fp16 storage / fp32 accumulation (the numerics of the
    reference's CUTLASS back-end, src/cuda/cutlass_bellpack_lib.cu:56-57)."""
    torch = _torch()
    m = sa.gen.ogbn_products_like(seed=3)
    assert m.rows == 2449029 and 1.2e8 < m.nztot() < 1.3e8
    w, n, tdt = 64, 256, torch.float16
    if arm == "reorder-on":
        g = sa.BlockingEngine(blocking_algo=7, tau=0.4, col_block_size=w).GetGrouping(m)
        d = sa.DeviceVBS.from_csr(m, g, w, device=0, dtype=sa.F16)
    else:
        g = sa.BlockingEngine(blocking_algo="fixed_size", row_block_size=64, col_block_size=w).GetGrouping(m)
        d = sa.DeviceVBS.from_csr(m, g, w, 64, False, device=0, dtype=sa.F16)
    B, ldb = _dense_rhs(torch, m.cols, n, tdt, seed=5)
    C1 = _product(torch, d, B, ldb, n, fill=1.0)
    C2 = _product(torch, d, B, ldb, n, fill=-2.0)
    assert torch.equal(C1, C2)
    assert _sampled_rows_check(torch, m, g, C1, n, B, ldb, tdt) <= 1e-5
    assert _column_checksum_check(torch, m, C1, n, B, ldb, tdt) <= 1e-5


def test_config3_rmat20_bf16_reorder_on_equals_reorder_off():
    """configs[3]: R-MAT 2^20 x 2^20, B = 512 columns, bf16 storage.  The test runs the 0.012 % density (~1.27e8 distinct nonzeros; the
    stated 0.1 % = 1.1e9 runs in bench.py --rmat-density 0.001 and takes minutes of host time) with BOTH arms of the reference's
    experiment -- clustered (blocking_algo 7) and fixed 64-row blocks -- and checks that the two products are the same matrix product:
    row perm_on[r] of C_on equals row perm_off[r'] of C_off for the same original row."""
    torch = _torch()
    n_side = 1 << 20
    m = sa.gen.rmat_device(20, target_nnz=int(1.0e-4 * n_side * n_side), seed=3, values="uniform")
    assert 1.0e8 < m.nztot() < 1.5e8            # the generator stops at the first round that reaches the target: ~1.27e8
    w, n, tdt = 64, 512, torch.bfloat16
    B, ldb = _dense_rhs(torch, m.cols, n, tdt, seed=6)
    g_off = sa.BlockingEngine(blocking_algo="fixed_size", row_block_size=64, col_block_size=w).GetGrouping(m)
    d_off = sa.DeviceVBS.from_csr(m, g_off, w, 64, False, device=0, dtype=sa.BF16)
    C_off = _product(torch, d_off, B, ldb, n, fill=3.0)
    assert torch.equal(C_off, _product(torch, d_off, B, ldb, n, fill=-1.0))
    assert _sampled_rows_check(torch, m, g_off, C_off, n, B, ldb, tdt) <= 1e-5
    assert _column_checksum_check(torch, m, C_off, n, B, ldb, tdt) <= 1e-5
    d_off.close()
    g_on = sa.BlockingEngine(blocking_algo=7, tau=0.4, col_block_size=w).GetGrouping(m)
    d_on = sa.DeviceVBS.from_csr(m, g_on, w, device=0, dtype=sa.BF16)
    C_on = _product(torch, d_on, B, ldb, n)
    assert _sampled_rows_check(torch, m, g_on, C_on, n, B, ldb, tdt, seed=2) <= 1e-5
    # same product through the two permutations: position of original row i in each ordering
    p_on, p_off = sa.get_permutation(g_on), sa.get_permutation(g_off)
    inv_on = np.empty(m.rows, np.int64); inv_on[p_on] = np.arange(m.rows)
    inv_off = np.empty(m.rows, np.int64); inv_off[p_off] = np.arange(m.rows)
    rows = np.random.Generator(np.random.PCG64(4)).integers(0, m.rows, 4096)
    a = C_on.view(n, -1)[:, torch.from_numpy(inv_on[rows]).cuda()]
    b = C_off.view(n, -1)[:, torch.from_numpy(inv_off[rows]).cuda()]
    # both sum the same exact products in fp32; only the order (hub rows: segments) may differ
    scale = b.abs().max().clamp_min(1e-30)
    assert float((a - b).abs().max() / scale) <= 2e-5


def test_config4_row_range_partition_on_one_gpu():
    """configs[4] in miniature on ONE GPU: the strong-scaling path of bench.py (--workload rmat --gpus N) with every rank played in
    turn -- block-row ranges by cost, a slab per rank built by sparta_vbs_create_from_csr, B row-sharded and handed over in the
    all-gather layout (sparta_vbs_spmm_gathered) -- against the single-handle product of the whole matrix."""
    torch = _torch()
    world, w, n = 4, 64, 128
    m = sa.gen.rmat_device(17, n_edges=10 << 17, seed=3, symmetrize=True, values="uniform")
    g = sa.BlockingEngine(blocking_algo=7, tau=0.4, col_block_size=w).GetGrouping(m)
    d = sa.DeviceVBS.from_csr(m, g, w, device=0)
    B, ldb = _dense_rhs(torch, m.cols, n, torch.float32, seed=8)
    C_full = _product(torch, d, B, ldb, n).view(n, -1)
    perm, part = sa.get_permutation(g), sa.get_partition(g)
    cost = np.add.reduceat(np.diff(m.rowptr)[perm].astype(np.float64) + 1.0, part[:-1])
    ranges = sa.dist.partition_by_cost(cost, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == len(part) - 1 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    shard_rows = sa.dist.padded_shard_rows(-(-m.cols // world), w)
    # the gathered buffer: `world` column-major slabs of shard_rows x n (rows past cols are zero)
    Bg = torch.zeros(world, n, shard_rows, dtype=torch.float32, device="cuda")
    Bfull = torch.zeros(n, world * shard_rows, dtype=torch.float32, device="cuda")
    Bfull[:, :m.cols] = B.view(n, ldb)[:, :m.cols]
    for q in range(world):
        Bg[q] = Bfull[:, q * shard_rows:(q + 1) * shard_rows]
    loads = []
    for r, (b0, b1) in enumerate(ranges):
        if b0 == b1:
            continue
        my_rows = perm[part[b0]:part[b1]]
        slab = sa.dist.row_slab(m, my_rows, world * shard_rows)
        gl = np.repeat(np.arange(b1 - b0, dtype=np.int64), np.diff(part[b0:b1 + 1]))
        ds = sa.DeviceVBS.from_csr(slab, gl, w, device=0)
        Cs = torch.empty(ds.rows * n, dtype=torch.float32, device="cuda")
        ds.spmm_gathered(Bg.view(-1), shard_rows, Cs, n, accumulate=False)
        torch.cuda.synchronize()
        # row k of the slab's C is original row my_rows[perm_local[k]]; the full product has it at the position of that row in `perm`
        pl = sa.get_permutation(gl)
        inv = np.empty(m.rows, np.int64); inv[perm] = np.arange(m.rows)
        want = C_full[:, torch.from_numpy(inv[my_rows[pl]]).cuda()]
        got = Cs.view(n, -1)
        scale = want.abs().max().clamp_min(1e-30)
        assert float((got - want).abs().max() / scale) <= 2e-5, r
        loads.append(float(cost[b0:b1].sum()))
        ds.close()
    assert max(loads) / (sum(loads) / world) < 1.5            # the cut by cost is balanced (hub block-rows are what limits it)


def test_matrix_market_file_through_the_whole_path(tmp_path):
    """What `bench.py --matrix cant.mtx` does with a staged SuiteSparse file (none can be fetched here), on a small one: a standard
    MatrixMarket coordinate file (real, general, 1-based, values) -> sparta_csr_read (IO_STRICT) -> Jaccard reorder -> VBS -> SpMM,
    every element against the oracle; and the reference-compatible reader (pattern-only, values 1) through the same path."""
    torch = _torch()
    src = sa.gen.fem3d(5, 4, 7, 3, seed=8)                                # 420 x 420, 3-dof FEM pattern with values
    path = tmp_path / "small_fem.mtx"
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% written by tests/test_configs_gpu.py\n")
        f.write("%d %d %d\n" % (src.rows, src.cols, src.nztot()))
        for i in range(src.rows):
            for k in range(src.rowptr[i], src.rowptr[i + 1]):
                f.write("%d %d %.9g\n" % (i + 1, src.colidx[k] + 1, src.vals[k]))
    m = sa.CSR.read_from_edgelist(str(path), mat_fmt=sa.FMT_MTX, mode=sa.IO_STRICT)
    assert (m.rows, m.cols) == (src.rows, src.cols) and np.array_equal(m.rowptr, src.rowptr) and np.array_equal(m.colidx, src.colidx)
    assert np.allclose(m.vals, src.vals, rtol=1e-7, atol=0)
    w, n = 32, 128
    g = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=32, force_fixed_size=True).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, w, 32, True)
    Bh = sa.gen.dense_rhs(v.cols, n, seed=3)
    Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bh, n)
    bound = U.abs_bound(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bh, n)
    d = v.to_device(0)
    C = _product(torch, d, torch.from_numpy(Bh).cuda(), v.cols, n, fill=5.0)
    assert np.all(np.abs(C.cpu().numpy() - Co) <= 1e-5 * bound + 1e-30)
    d.spmm(torch.from_numpy(Bh).cuda(), C, n, accumulate=False, algo=sa.SPMM_EXACT)
    torch.cuda.synchronize()
    assert np.array_equal(C.cpu().numpy(), Co)


def test_rmat_row_slab_is_sampled_without_the_rest_of_the_graph():
    """configs[4] is too large to generate on one GPU; a rank's rows are sampled alone (gen.rmat_device(row_slab=(k, parts))): the top bits of
    the row are fixed, the column bits of those levels come from the conditional distribution.  Structure of the slab, its share of the edges
    (row bits are independent: 0.76 / 0.24 per bit) and the column marginals of the fixed levels against the same rows of a whole graph."""
    _torch()
    scale, parts = 14, 4
    n = 1 << scale
    whole = sa.gen.rmat_device(scale, n_edges=40 << scale, seed=5, values=None)
    share = []
    for k in range(parts):
        m = sa.gen.rmat_device(scale, n_edges=int((40 << scale) * [0.76 * 0.76, 0.24 * 0.76, 0.76 * 0.24, 0.24 * 0.24][k]), seed=5, values="uniform", row_slab=(k, parts))
        assert (m.rows, m.cols) == (n // parts, n) and len(m.rowptr) == m.rows + 1 and m.rowptr[-1] == m.nztot()
        assert m.colidx.min() >= 0 and m.colidx.max() < n and np.all(np.diff(m.rowptr) >= 0)
        for i in (0, m.rows // 2, m.rows - 1):                # columns ascending and distinct inside a row (sorted, de-duplicated keys)
            c = m.colidx[m.rowptr[i]:m.rowptr[i + 1]]
            assert np.all(np.diff(c) > 0)
        lo, hi = whole.rowptr[k * m.rows], whole.rowptr[(k + 1) * m.rows]
        share.append((m.nztot(), int(hi - lo)))
        # column marginal of the top level (bit scale - 1 of the column) given the slab's top row bit: b / (a + b) = 0.25 for row bit 0, d / (c + d) = 0.208 for row bit 1
        top = float(np.mean(m.colidx >= n // 2))
        top_whole = float(np.mean(whole.colidx[lo:hi] >= n // 2))
        want = 0.05 / 0.24 if k >= 2 else 0.19 / 0.76        # (of the raw edges; removing duplicates shifts it towards the sparser half, equally in both)
        assert abs(top - top_whole) < 0.01 and abs(top - want) < 0.06, (k, top, top_whole, want)
    for k, (got, ref_cnt) in enumerate(share):                # same expected edge count as the same rows of the whole graph (duplicates removed in both)
        assert abs(got - ref_cnt) < 0.05 * ref_cnt + 500, (k, got, ref_cnt)


# ---- round 3: the stated densities / sizes through the canonical piece-wise graph (bench.py --workload rmat-part) ----------------------
def _gathered_view(torch, B_gath, P, shard_rows, n, shard_ld=None):
    """the gathered layout (P column-major slabs of shard_rows x n, columns shard_ld apart) as one n x cols view for the checks: (n, P, shard_rows) -> (n, P * shard_rows)"""
    shard_ld = shard_rows if shard_ld is None else shard_ld
    return B_gath.view(P, n, shard_ld)[:, :, :shard_rows].permute(1, 0, 2).reshape(n, P * shard_rows)


def _gathered_b(torch, P, shard_rows, n, tdt, padded):
    """B of the canonical graph in the layout the all-gather of P ranks leaves; padded: the columns of a slab shard_rows + 64 apart, as bench_parts.py allocates
    them (columns a power of two apart share cache sets and channels: sparta_vbs_spmm_gathered_ld)"""
    shard_ld = sa.dist.padded_shard_ld(shard_rows, 2 if tdt != torch.float32 else 4) if padded else shard_rows
    B_gath = torch.zeros(P * n * shard_ld, dtype=tdt, device="cuda")
    for s in range(P):
        B_gath.view(P, n, shard_ld)[s, :, :shard_rows] = sa.gen.dense_rhs_rows(s * shard_rows, (s + 1) * shard_rows, n, seed=7, dtype=tdt, device=0).view(n, shard_rows)
    return B_gath, shard_ld


def _part_checks(torch, m, g, d, B_gath, P, shard_rows, n, tdt, seed, shard_ld=None, keep=False, n_rows=64):
    C = torch.full((d.rows * n,), 2.5, dtype=torch.float32, device="cuda")
    d.spmm_gathered(B_gath, shard_rows, C, n, shard_ld=shard_ld)
    C2 = torch.full((d.rows * n,), -7.0, dtype=torch.float32, device="cuda")
    d.spmm_gathered(B_gath, shard_rows, C2, n, shard_ld=shard_ld)
    torch.cuda.synchronize()
    assert torch.equal(C, C2), "a second product into a dirty buffer must give the same bits"
    del C2
    Bflat = _gathered_view(torch, B_gath, P, shard_rows, n, shard_ld).contiguous().view(-1)      # column-major, ld = cols: what the check helpers read
    ldb = P * shard_rows
    assert _sampled_rows_check(torch, m, g, C, n, Bflat, ldb, tdt, n_rows=n_rows, seed=seed) <= 1e-5
    assert _column_checksum_check(torch, m, C, n, Bflat, ldb, tdt) <= 1e-5
    del Bflat
    return C if keep else None


def test_config3_rmat20_at_the_stated_density_0p1_percent_slab_streamed():
    """configs[3] at its stated 0.1 %: R-MAT 2^20 x 2^20, 1.1e9 distinct nonzeros, B = 512 columns, bf16 -- the reorder-OFF arm of the reference's
    experiment (-a 2 -F 1: fixed 64 x 64 grid, src/scripts/run_multiplication_experiments_fixed_cluster.sh:14-16), streamed through the GPU in the 8
    parts of equal expected cost that bench.py --workload rmat-part --rmat-scale 20 --rmat-density 1e-3 --slabs 8 runs: every part is generated alone
    (canonical graph), built (sparta_vbs_create_from_csr), multiplied against B in the gathered layout and checked -- sampled + heaviest rows against
    float64, the column checksum over every row, bit-reproducibility; the parts tile the graph and hold the stated number of nonzeros."""
    torch = _torch()
    scale, dens, P, n, w, tdt = 20, 1e-3, 8, 512, 64, torch.bfloat16
    n_side = 1 << scale
    E = sa.gen.rmat_raw_edges_for_density(scale, dens)
    cuts = sa.gen.rmat_cuts(scale, E, P, n_cols=n)
    assert cuts[0][0] == 0 and cuts[-1][1] == n_side and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    shard_rows = n_side // P
    B_gath = torch.cat([sa.gen.dense_rhs_rows(s * shard_rows, (s + 1) * shard_rows, n, seed=7, dtype=tdt, device=0) for s in range(P)])
    total, costs = 0, []
    for ip, (r0, r1) in enumerate(cuts):
        m = sa.gen.rmat_rows(scale, E, r0, r1, seed=3, device=0)
        total += m.nztot()
        g = np.arange(m.rows, dtype=np.int64) // 64
        st = sa.DeviceVBS.plan_stats(m, g, w, 64, False, dtype=sa.BF16)
        d = sa.DeviceVBS.from_csr(m, g, w, 64, False, device=0, dtype=sa.BF16)
        info, sp = d.info(), d.sparse_info()
        assert st["tile_blocks"] == info["nblocks"] and st["tile_area"] == info["nztot"]      # the dry run predicts exactly what is built
        assert abs(st["sparse_nnz"] - sp["nnz"]) <= 0.001 * max(sp["nnz"], 1)                 # (values that round to zero in bf16 leave the device copy)
        costs.append(sp["nnz"] + 25.0 * info["nblocks"] + 34.0 * m.rows)           # gen.rmat_piece_table's model
        _part_checks(torch, m, g, d, B_gath, P, shard_rows, n, tdt, seed=30 + ip)
        d.close()
        del m
        torch.cuda.empty_cache()
    assert abs(total - dens * n_side * n_side) / (dens * n_side * n_side) < 0.01, total
    costs = np.array(costs)
    assert costs.max() / costs.mean() < 1.08, costs                 # the cuts came from the marginals alone; the built parts carry equal cost


def test_config4_one_full_size_slab_of_the_8m_row_graph():
    """configs[4] at full size, one rank's share: part 0 of the 8 parts of the 8.4 M x 8.4 M R-MAT at 0.01 % (~1.1e9 of its 7.0e9 nonzeros: the hub),
    generated without the rest of the graph, B = 256 columns fp16 in the layout the all-gather of 8 ranks leaves (slabs of 2^20 rows) -- what rank 0 of
    `bench.py --gpus 8` and slab 0 of `bench.py --gpus 1 --slabs 8` run -- through the property checks."""
    torch = _torch()
    scale, dens, P, n, w, tdt = 23, 1e-4, 8, 256, 64, torch.float16
    n_side = 1 << scale
    E = sa.gen.rmat_raw_edges_for_density(scale, dens)
    cuts = sa.gen.rmat_cuts(scale, E, P, n_cols=n)
    r0, r1 = cuts[0]
    m = sa.gen.rmat_rows(scale, E, r0, r1, seed=3, device=0)
    assert m.cols == n_side and 0.10 < m.nztot() / (dens * n_side * n_side) < 0.20          # the hub part: few rows, a large share of the nonzeros
    assert (r1 - r0) * 8 < n_side // 4
    shard_rows = n_side // P
    B_gath = torch.cat([sa.gen.dense_rhs_rows(s * shard_rows, (s + 1) * shard_rows, n, seed=7, dtype=tdt, device=0) for s in range(P)])
    g = np.arange(m.rows, dtype=np.int64) // 64
    d = sa.DeviceVBS.from_csr(m, g, w, 64, False, device=0, dtype=sa.F16)
    assert d.info()["tiles64"] > 0 and d.sparse_info()["nnz"] > 0          # hub blocks on MFMA tiles, the tail on the sparse-row kernels
    _part_checks(torch, m, g, d, B_gath, P, shard_rows, n, tdt, seed=41)
    d.close()


# ---- round 4: the parts of the power-law configs that no test had reached ("configs not exercised") ------------------------------------------
def _canonical_part(scale, dens, P, ip, n_cols=512):
    E = sa.gen.rmat_raw_edges_for_density(scale, dens)
    r0, r1 = sa.gen.rmat_cuts(scale, E, P, n_cols=n_cols)[ip]            # (the parts bench.py --workload rmat-part cuts for a product of n_cols columns)
    return sa.gen.rmat_rows(scale, E, r0, r1, seed=3, device=0), (r0, r1)


@pytest.mark.parametrize("dens,P,ip", [(5e-2, 64, 0), (5e-2, 64, 63), (1e-2, 16, 0), (1e-2, 16, 15)], ids=["5pct-hub", "5pct-tail", "1pct-hub", "1pct-tail"])
def test_config3_rmat20_at_1_and_5_percent_hub_and_tail_parts(dens, P, ip):
    """configs[3] at its stated 1 % and 5 %: part 0 -- the hub: at 5 % 8-12 k rows of which 2048+ are fully dense, 0.8-1.1e10 stored elements in 16-bit tiles, the one
    place a dense tile stream of that size occurs (it runs on the hub plan: group tiles of four 64-row tiles through the GEMM-shaped kernel, k_hub16.hip) -- and the
    LAST part (the all-sparse end of the graph) of the 64 / 16 parts of equal expected cost that bench.py --workload rmat-part --rmat-density 0.05 / 0.01 streams,
    reorder OFF (-a 2 -F 1: src/scripts/run_multiplication_experiments_fixed_cluster.sh:14-16), B = 512 columns bf16 in the (padded) all-gather layout."""
    torch = _torch()
    scale, n, w, tdt = 20, 512, 64, torch.bfloat16
    m, (r0, r1) = _canonical_part(scale, dens, P, ip)
    shard_rows = (1 << scale) // P
    B_gath, shard_ld = _gathered_b(torch, P, shard_rows, n, tdt, padded=True)
    assert shard_ld == shard_rows + 64
    g = np.arange(m.rows, dtype=np.int64) // 64
    d = sa.DeviceVBS.from_csr(m, g, w, 64, False, device=0, dtype=sa.BF16)
    info, sp, hub = d.info(), d.sparse_info(), d.hub_info()
    if ip == 0:
        assert hub["steps"] > 100000 and hub["tiles_per_group"] == 4 and hub["tiles"] >= 0.9 * info["tiles64"], (hub, info)
        assert hub["union_area"] <= 1.15 * hub["stored_area"], hub          # the Jaccard grouping of the block-rows: the unions hold few blocks their members lack
        if dens == 5e-2:
            assert hub["stored_area"] > 7e9 and 8192 <= m.rows <= 16384, m.rows    # (8192 rows with sa.gen.rmat_cost_constants(512) as of r4c; 12288 / 9216 with earlier fits)
    else:
        assert sp["nnz"] > 0.25 * m.nztot()                                  # the tail: sparse rows carry a large share (at 5 % even the last part keeps tiles)
    _part_checks(torch, m, g, d, B_gath, P, shard_rows, n, tdt, seed=50 + ip, shard_ld=shard_ld, n_rows=16 if ip == 0 else 64)      # (a hub row holds 10^5..10^6 nonzeros)
    d.close()


@pytest.mark.parametrize("ip", [1, 6])
def test_config3_rmat20_at_0p1_percent_reorder_on_equals_off_on_a_part(ip):
    """configs[3] at its stated 0.1 %, BOTH arms of the reference's experiment (-a 2 -F 1 against the clustering arm) on parts 1 and 6 of the 8 that the bench
    streams: each arm through the property checks, and the two products are the same rows of the same product -- row inv_on[i] of C_on equals row inv_off[i] of
    C_off for every original row i sampled."""
    torch = _torch()
    scale, dens, P, n, w, tdt = 20, 1e-3, 8, 512, 64, torch.bfloat16
    m, _ = _canonical_part(scale, dens, P, ip)
    shard_rows = (1 << scale) // P
    B_gath, shard_ld = _gathered_b(torch, P, shard_rows, n, tdt, padded=True)
    g_off = np.arange(m.rows, dtype=np.int64) // 64
    d_off = sa.DeviceVBS.from_csr(m, g_off, w, 64, False, device=0, dtype=sa.BF16)
    C_off = _part_checks(torch, m, g_off, d_off, B_gath, P, shard_rows, n, tdt, seed=60 + ip, shard_ld=shard_ld, keep=True)
    d_off.close()
    g_on = sa.BlockingEngine(blocking_algo=7, tau=0.4, col_block_size=w, row_block_size=64, force_fixed_size=False, sim_measure=1).GetGrouping(m)
    d_on = sa.DeviceVBS.from_csr(m, g_on, w, 64, False, device=0, dtype=sa.BF16)
    C_on = _part_checks(torch, m, g_on, d_on, B_gath, P, shard_rows, n, tdt, seed=70 + ip, shard_ld=shard_ld, keep=True)
    d_on.close()
    p_on, p_off = sa.get_permutation(g_on), sa.get_permutation(g_off)
    inv_on = np.empty(m.rows, np.int64); inv_on[p_on] = np.arange(m.rows)
    inv_off = np.empty(m.rows, np.int64); inv_off[p_off] = np.arange(m.rows)
    rows = np.random.Generator(np.random.PCG64(4 + ip)).integers(0, m.rows, 4096)
    a = C_on.view(n, -1)[:, torch.from_numpy(inv_on[rows]).cuda()]
    b = C_off.view(n, -1)[:, torch.from_numpy(inv_off[rows]).cuda()]
    scale_ = b.abs().max().clamp_min(1e-30)
    assert float((a - b).abs().max() / scale_) <= 2e-5                       # the same exact products summed in fp32, in two orders


def test_config4_the_last_slab_of_the_8m_row_graph():
    """configs[4] at full size, part 7 of 8: 3.1 M rows, the all-sparse end of the 8.4 M x 8.4 M R-MAT at 0.01 % (what rank 7 of `bench.py --gpus 8` runs),
    B = 256 columns fp16 in the padded all-gather layout of 8 ranks."""
    torch = _torch()
    scale, dens, P, n, w, tdt = 23, 1e-4, 8, 256, 64, torch.float16
    m, (r0, r1) = _canonical_part(scale, dens, P, 7, n_cols=n)
    assert r1 == 1 << scale and r1 - r0 > 2500000
    shard_rows = (1 << scale) // P
    B_gath, shard_ld = _gathered_b(torch, P, shard_rows, n, tdt, padded=True)
    g = np.arange(m.rows, dtype=np.int64) // 64
    d = sa.DeviceVBS.from_csr(m, g, w, 64, False, device=0, dtype=sa.F16)
    assert d.sparse_info()["nnz"] > 0.8 * m.nztot()
    _part_checks(torch, m, g, d, B_gath, P, shard_rows, n, tdt, seed=47, shard_ld=shard_ld)
    d.close()
