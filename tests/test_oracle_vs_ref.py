"""CPU, only where oracle/_ref/libsparta_ref.so exists (built from /root/reference by `make -C oracle ref`):
three-way agreement reference == oracle == product host code on seeded random inputs beyond the committed goldens."""
import numpy as np
import pytest

from oracle import oracle as O, ref
import sparta_amd as sa

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (no /root/reference on this box)")


def _rc(m):
    return ref.RefCSR(m.rows, m.cols, m.rowptr, m.colidx.astype(np.int64), m.vals)


@pytest.mark.parametrize("seed", range(12))
def test_random_matrices_three_way(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    rows, cols = int(rng.integers(20, 400)), int(rng.integers(20, 400))
    nnz = int(rng.integers(rows, rows * min(cols, 12)))
    m = sa.gen.uniform_random(rows, cols, min(nnz, rows * cols // 2), seed=100 + seed, pattern_only=bool(seed % 3 == 0))
    w = int(rng.choice([1, 3, 8, 16, 40]))
    tau = float(rng.choice([0.1, 0.3, 0.5, 0.8, 1.0]))
    rbs = int(rng.choice([2, 5, 16]))
    rc = _rc(m)
    for algo in (3, 4, 0, 2, 5):
        for ff in (False, True):
            gr, st = rc.grouping(algo=algo, tau=tau, col_block_size=w, row_block_size=rbs, force_fixed_size=ff, use_groups=bool(seed & 1))
            e = sa.BlockingEngine(tau=tau, col_block_size=w, row_block_size=rbs, force_fixed_size=ff, blocking_algo=algo, use_groups=bool(seed & 1))
            gp = e.GetGrouping(m)
            assert np.array_equal(gp, gr), (algo, ff)
            if True:
                go, _ = O.get_grouping(m.rows, m.rowptr, m.colidx, algo, 1, tau, w, rbs, bool(seed & 1), True, ff)
                assert np.array_equal(go, gr), (algo, ff)
            rv = ref.RefVBR(rc, gr, w, rbs, ff)
            arr = rv.export()
            pv = sa.VBR().fill_from_CSR_inplace(m, gr, w, rbs, ff)
            ov = O.OracleVBR(m.rows, m.cols, m.rowptr, m.colidx, m.vals, gr, w, rbs, ff)
            for x, y, z in zip(arr, (pv.row_part, pv.nzcount, pv.jab, pv.mab), (ov.row_part, ov.nzcount, ov.jab, ov.mab)):
                assert np.array_equal(x, y) and np.array_equal(x, z)
            if not ff:      # padded cols: the reference driver passes a B sized for the unpadded matrix
                B = sa.gen.dense_rhs(rv.cols, 3, seed=seed)
                assert np.array_equal(rv.multiply(B, 3), O.vbr_multiply(ov.rows, ov.cols, w, ov.row_part, ov.nzcount, ov.jab, ov.mab, B, 3))


def test_reference_reader_fixture_matches_golden():
    import _util as U
    k = U.load("kat9.npz")
    c = ref.RefCSR.read("/root/reference/data/TEST_matrix_weighted.el")
    rp, ci, v = c.export()
    assert np.array_equal(rp, k["rowptr"]) and np.array_equal(ci, k["colidx"]) and np.array_equal(v, k["vals"])


@pytest.mark.parametrize("name", ["bcsstk18_r.el", "ca-HepPh_r.el"])
def test_real_world_edgelists(name):
    """the reference's own data/minitest matrices through its reader, pattern-only, default algorithm"""
    path = "/root/reference/data/minitest/" + name
    c = ref.RefCSR.read(path, pattern_only=True)
    rp, ci, _ = c.export()
    m = sa.CSR(c.rows, c.cols, rp, ci.astype(np.int32), None)
    gr, st = c.grouping(algo=3, tau=0.5, col_block_size=64)
    e = sa.BlockingEngine(tau=0.5, col_block_size=64)
    assert np.array_equal(e.GetGrouping(m), gr) and e.comparison_counter == st["comparison_counter"]
    rv = ref.RefVBR(c, gr, 64)
    pv = sa.VBR().fill_from_CSR_inplace(m, gr, 64)
    for x, y in zip(rv.export(), (pv.row_part, pv.nzcount, pv.jab, pv.mab)):
        assert np.array_equal(x, y)


@pytest.mark.skipif(not ref.available(), reason="compiled reference not present")
def test_structured_mn_random_cases_against_the_live_reference():
    """blocking_algo 1: product, oracle restatement and the compiled reference on random small matrices and m:n settings"""
    rng = np.random.Generator(np.random.PCG64(91))
    for case in range(150):
        rows, cols = int(rng.integers(2, 100)), int(rng.integers(2, 80))
        m = sa.gen.uniform_random(rows, cols, int(rng.integers(0, rows * cols // 3 + 1)), seed=int(rng.integers(1 << 30)))
        w, sm = int(rng.choice([1, 2, 3, 8])), int(rng.integers(0, 2))
        tau = float(rng.integers(1, 12)) if sm == 0 else float(rng.choice([0.2, 0.5, 0.8, 1.0]))
        mm, nn, ug, up = int(rng.integers(1, 4)), int(rng.integers(1, 6)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        e = sa.BlockingEngine(tau=tau, col_block_size=w, use_groups=ug, use_pattern=up, blocking_algo=1, sim_measure=sm,
                              structured_m=mm, structured_n=nn)
        g = e.GetGrouping(m)
        gr, st = ref.RefCSR(m.rows, m.cols, m.rowptr, m.colidx, m.vals).grouping(
            algo=1, tau=tau, col_block_size=w, use_groups=ug, use_pattern=up, sim_measure=sm, structured_m=mm, structured_n=nn)
        go, co = O.get_grouping(m.rows, m.rowptr, m.colidx, 1, sm, tau, w, 1, ug, up, False, mm, nn)
        assert np.array_equal(g, gr) and np.array_equal(go, gr), case
        assert e.comparison_counter == st["comparison_counter"] == co["comparison_counter"]
        assert e.merge_counter == st["merge_counter"] == co["merge_counter"]
