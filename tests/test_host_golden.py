"""CPU: the product's host-side C++ (reorder engine + VBS builder, through the C-ABI / sparta_amd mirror classes)
against the golden vectors from the compiled reference -- including algorithm 5 (IterativeBlockingKeeper) and 6."""
import os

import numpy as np
import pytest

import sparta_amd as sa
import _util as U


def test_similarity_and_merge_kat():
    A, B = [1, 2, 5, 10, 12, 20], [0, 2, 4, 10, 16]
    assert sa.row_distance(0, A, 1, B, 1, 1) == 7.0
    assert sa.row_distance(1, A, 1, B, 1, 1) == pytest.approx(0.777778, abs=1e-6)
    assert sa.row_distance(0, A, 1, B, 1, 3) == 3.0 and sa.row_distance(1, A, 1, B, 1, 4) == 0.5
    assert sa.merge_rows([1, 2, 3], [5]).tolist() == [5]
    assert sa.merge_rows([1, 4, 9], [2, 4, 7]).tolist() == [1, 2, 4, 7]
    assert sa.merge_rows([1, 4, 9], [2, 4, 7, 12]).tolist() == [1, 2, 4, 7, 12]


def test_appendix_b_kat():
    k = U.load("kat9.npz")
    m = sa.CSR(9, 9, k["rowptr"], k["colidx"], k["vals"])
    e = sa.BlockingEngine(tau=0.6, col_block_size=3, row_block_size=3)
    g = e.GetGrouping(m)
    assert g.tolist() == [0, 1, 1, 1, 0, 5, 0, 0, 8]
    assert (e.comparison_counter, e.merge_counter) == (13, 5)
    v = sa.VBR().fill_from_CSR_inplace(m, g, 3)
    assert (v.rows, v.cols, v.block_rows, v.block_cols, v.nztot) == (9, 9, 4, 3, 33)
    assert v.row_part.tolist() == [0, 4, 7, 8, 9] and v.nzcount.tolist() == [0, 3, 1, 1] and v.jab.tolist() == [0, 1, 2, 2, 0]
    assert np.array_equal(v.mab, k["mab"])
    e.CollectBlockingInfo(m)
    assert (e.VBR_nzcount, e.VBR_nzblocks_count, e.VBR_longest_row) == (33, 5, 3)
    assert e.VBR_average_height == pytest.approx(2.2)
    assert sa.BlockingEngine(tau=0.6, col_block_size=3, row_block_size=3, force_fixed_size=True).GetGrouping(m).tolist() == \
        [0, 1, 1, 2, 0, 2, 0, 1, 2]
    assert sa.BlockingEngine(tau=0.6, col_block_size=3, row_block_size=3, force_fixed_size=True, blocking_algo=5).GetGrouping(m).tolist() == \
        [0, 1, 1, 1, 0, 2, 0, 2, 2] == k["g_a5_B3_F1"].tolist()


def test_primitives_golden():
    p = U.load("prims.npz")
    for t in range(int(p["n"])):
        A, B = p["%d/A" % t], p["%d/B" % t]
        ga, gb, bs = map(int, p["%d/par" % t])
        assert np.array_equal(sa.merge_rows(A, B), p["%d/merged" % t]), t
        d = p["%d/dist" % t]
        assert np.float32(sa.row_distance(0, A, ga, B, gb, bs)) == d[0] and np.float32(sa.row_distance(1, A, ga, B, gb, bs)) == d[1], t
    for t in range(int(p["nperm"])):
        g = p["perm%d/g" % t]
        assert np.array_equal(sa.get_permutation(g), p["perm%d/perm" % t])
        assert np.array_equal(sa.get_partition(g), p["perm%d/part" % t])
        assert np.array_equal(sa.get_fixed_size_grouping(g, 5), p["perm%d/fixed5" % t])


@pytest.mark.parametrize("key,name,cfg", U.case_list(), ids=[c[0] for c in U.case_list()])
def test_case_golden(key, name, cfg):
    m = U.matrices()[name]
    f = U.case_fields(key)
    w, rbs, ff = cfg["w"], cfg.get("rbs", 1), cfg.get("ff", False)
    e = sa.BlockingEngine(tau=cfg["tau"], col_block_size=w, row_block_size=rbs, use_groups=cfg.get("use_groups", False),
                          use_pattern=cfg.get("use_pattern", True), force_fixed_size=ff, blocking_algo=cfg["algo"],
                          sim_measure=cfg.get("sim", 1))
    g = e.GetGrouping(m)
    assert np.array_equal(g, f["grouping"]), "grouping differs from the reference"
    if cfg["algo"] in (0, 3, 4, 5):
        assert [e.comparison_counter, e.merge_counter] == f["counters"].tolist()
    assert np.array_equal(sa.get_permutation(g), f["perm"])
    v = sa.VBR().fill_from_CSR_inplace(m, g, w, rbs, ff)
    assert [v.rows, v.cols, v.block_rows, v.block_cols, v.nztot] == f["dims"].tolist()
    assert np.array_equal(v.row_part, f["row_part"]) and np.array_equal(v.nzcount, f["nzcount"]) and np.array_equal(v.jab, f["jab"])
    assert U.sha(v.mab) == str(f["mab_sha"]), "mab differs from the reference"
    e.CollectBlockingInfo(m)
    assert [e.VBR_nzcount, e.VBR_nzblocks_count, e.VBR_longest_row] == f["info"].tolist()
    assert np.float32(e.VBR_average_height) == f["avg_height"]


def _mn_cases():
    z = U.load("mn.npz")
    return sorted({tuple(k.split("/")[:2]) for k in z.files})


@pytest.mark.parametrize("name,k", _mn_cases())
def test_structured_mn_blocking_matches_the_reference(name, k):
    """blocking_algo 1 (IterativeBlockingPatternMN, blocking.cpp:19-87): product == oracle == compiled reference (mn.npz)"""
    import ast
    from oracle import oracle as O
    z = U.load("mn.npz")
    cfg = ast.literal_eval(str(z["%s/%s/cfg" % (name, k)]))
    m = U.matrices()[name]
    e = sa.BlockingEngine(tau=cfg["tau"], col_block_size=cfg["w"], row_block_size=cfg.get("rbs", 1), use_groups=cfg.get("use_groups", False),
                          use_pattern=cfg.get("use_pattern", True), force_fixed_size=cfg.get("ff", False), blocking_algo=1,
                          sim_measure=cfg.get("sim", 1), structured_m=cfg["m"], structured_n=cfg["n"])
    g = e.GetGrouping(m)
    want, cnt = z["%s/%s/grouping" % (name, k)], z["%s/%s/counters" % (name, k)]
    assert np.array_equal(g, want)
    assert [e.comparison_counter, e.merge_counter] == list(cnt)
    go, co = O.get_grouping(m.rows, m.rowptr, m.colidx, 1, cfg.get("sim", 1), cfg["tau"], cfg["w"], cfg.get("rbs", 1),
                            cfg.get("use_groups", False), cfg.get("use_pattern", True), cfg.get("ff", False), cfg["m"], cfg["n"])
    assert np.array_equal(go, want) and [co["comparison_counter"], co["merge_counter"]] == list(cnt)


@pytest.mark.parametrize("algo", [0, 1, 3, 4, 5])
def test_candidate_filter_is_exact(monkeypatch, algo):
    """The inverted-index candidate filter of the reorder engine only skips list walks whose outcome is known (no common
    block): groupings and counters are identical with the filter forced on, forced off, and left to its size threshold."""
    outs = []
    for mode in ("0", "1", None):
        if mode is None:
            monkeypatch.delenv("SPARTA_REORDER_FILTER", raising=False)
        else:
            monkeypatch.setenv("SPARTA_REORDER_FILTER", mode)
        res = []
        for m, w, tau in ((sa.gen.uniform_random(5000, 7000, 30000, seed=41), 16, 0.7), (sa.gen.rmat(12, 40000, seed=42, pattern_only=True), 8, 0.5),
                          (sa.gen.banded(3000, 9, 0.5, seed=43), 4, 0.4)):
            for sim in (1, 0):
                e = sa.BlockingEngine(tau=tau if sim else 6.0, col_block_size=w, row_block_size=24, blocking_algo=algo, sim_measure=sim,
                                      use_groups=bool(algo == 3))
                g = e.GetGrouping(m)
                res.append((g.copy(), e.comparison_counter, e.merge_counter, np.float32(e.average_merge_tau).tobytes(),
                            np.float32(e.average_row_distance).tobytes()))
        outs.append(res)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]


@pytest.mark.parametrize("algo", [3, 4])
def test_near_linear_clocked_form_is_exact(monkeypatch, algo):
    """clocked_sparse (two-class bookkeeping of distances[] + a heap of candidate rows) against the plain scan: same grouping,
    counters and float statistics, with and without pattern merging, with empty and duplicate rows, for tau from 0 to just below 1."""
    monkeypatch.setenv("SPARTA_REORDER_FILTER", "1")          # build the index for small inputs too
    rng = np.random.Generator(np.random.PCG64(17 + algo))
    for case in range(60):
        rows, cols = int(rng.integers(1, 500)), int(rng.integers(1, 400))
        m = sa.gen.uniform_random(rows, cols, int(rows * cols * float(rng.choice([0.003, 0.02, 0.1]))), seed=int(rng.integers(1 << 30)))
        if case % 3 == 0 and rows > 8:                          # knock out a quarter of the rows, repeat some others
            cnt = np.diff(m.rowptr)
            cnt[rng.integers(0, rows, rows // 4)] = 0
            src = rng.integers(0, rows, rows // 6)
            rp, ci = [0], []
            for r in range(rows):
                rr = int(src[r % len(src)]) if (r % 7 == 3 and len(src)) else r
                ci.extend(m.colidx[m.rowptr[rr]:m.rowptr[rr] + cnt[rr]].tolist())
                rp.append(len(ci))
            m = sa.CSR(rows, cols, rp, np.array(ci, np.int32), None)
        w, tau = int(rng.choice([1, 3, 8, 32])), float(rng.choice([0.0, 0.2, 0.5, 0.8, 0.99]))
        up = bool(rng.integers(0, 2))
        res = []
        for mode in ("0", "1"):
            monkeypatch.setenv("SPARTA_REORDER_SCALABLE", mode)
            e = sa.BlockingEngine(tau=tau, col_block_size=w, use_pattern=up, blocking_algo=algo)
            g = e.GetGrouping(m)
            res.append((g.tobytes(), e.comparison_counter, e.merge_counter, np.float32(e.average_merge_tau).tobytes(),
                        np.float32(e.average_row_distance).tobytes()))
        assert res[0] == res[1], (case, rows, cols, w, tau, up)


# ---- blocking_algo 7: LSH-bucketed clustering (an extension, approximate by design -- validated by QUALITY) --------------

def _blocks_area(m, g, w):
    v = sa.VBR().fill_from_CSR_inplace(m, g, w)
    return len(v.jab), int(v.nztot)


@pytest.mark.parametrize("name,w,tau", [("fem", 32, 0.6), ("fem", 64, 0.4), ("rmat", 64, 0.6), ("rmat", 32, 0.4), ("uniform", 64, 0.7), ("banded", 16, 0.5)])
def test_minhash_reorder_quality_against_the_exact_algorithm(name, w, tau):
    """the scalable reorder is judged by the number of nonzero blocks and the stored area (fill) it
    reaches relative to the exact algorithm 3 on inputs both can handle."""
    m = {"fem": lambda: sa.gen.fem3d(7, 7, 60, 3, 2, pattern_only=True), "rmat": lambda: sa.gen.rmat(13, 100000, seed=3, symmetrize=True),
         "uniform": lambda: sa.gen.uniform_random(6000, 6000, 60000, seed=1, pattern_only=True),
         "banded": lambda: sa.gen.banded(12000, 12, density=0.5, seed=4, pattern_only=True)}[name]()
    exact = sa.BlockingEngine(blocking_algo=3, tau=tau, col_block_size=w)
    lsh = sa.BlockingEngine(blocking_algo="minhash", tau=tau, col_block_size=w)
    ge, gl = exact.GetGrouping(m), lsh.GetGrouping(m)
    be, ae = _blocks_area(m, ge, w)
    bl, al = _blocks_area(m, gl, w)
    assert bl <= 1.15 * be and al <= 1.15 * ae, (be, bl, ae, al)            # measured: -17 % .. +13 % area, -0.2 % .. +8 % blocks
    assert lsh.comparison_counter < exact.comparison_counter / 10          # it looks at a small fraction of the pairs
    # a valid grouping in the reference's convention: the id of a group is its first (seed) row
    for gid in np.unique(gl):
        assert gl[gid] == gid and np.flatnonzero(gl == gid)[0] == gid
    # merges counted = rows that joined a seed
    assert lsh.merge_counter == m.rows - len(np.unique(gl))
    # deterministic
    assert np.array_equal(gl, sa.BlockingEngine(blocking_algo=7, tau=tau, col_block_size=w).GetGrouping(m))


def test_minhash_reorder_rules():
    # identical rows always end up together, whatever tau; empty rows form one group (distance 0 between empty rows)
    rows = [[0, 5, 9], [], [1, 2], [0, 5, 9], [], [40, 41], [1, 2], [0, 5, 9]]
    rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
    m = sa.CSR(len(rows), 64, rp, np.array([c for r in rows for c in r], np.int32), None)
    g = sa.BlockingEngine(blocking_algo=7, tau=0.0, col_block_size=1).GetGrouping(m)
    assert g.tolist() == [0, 1, 2, 0, 1, 5, 2, 0]
    # tau = 1 with Jaccard merges everything that shares a bucket with the pattern; a cluster cap stops the growth
    m2 = sa.gen.banded(400, 3, seed=2, pattern_only=True)
    e = sa.BlockingEngine(blocking_algo=7, tau=0.9, col_block_size=4, minhash_max_rows=8)
    g2 = e.GetGrouping(m2)
    sizes = np.bincount(g2)[np.unique(g2)]
    assert np.median(sizes) == 8 and sizes.max() < 16      # soft limit: a row's identical copies join together with it
    assert np.bincount(sa.BlockingEngine(blocking_algo=7, tau=0.9, col_block_size=4).GetGrouping(m2)).max() > 8
    # force_fixed_size re-chunks the order like for every other algorithm (blocking.cpp:670-673)
    g3 = sa.BlockingEngine(blocking_algo=7, tau=0.5, col_block_size=4, row_block_size=16, force_fixed_size=True).GetGrouping(m2)
    assert np.bincount(g3).tolist() == [16] * 25
    # unsorted rows are refused (the exact algorithms reproduce the reference's behaviour on them; this one does not pretend to)
    bad = sa.CSR(2, 8, [0, 2, 3], np.array([5, 1, 2], np.int32), None)
    with pytest.raises(sa.SpartaError):
        sa.BlockingEngine(blocking_algo=7, tau=0.5, col_block_size=2).GetGrouping(bad)
    # 0 rows
    assert sa.BlockingEngine(blocking_algo=7).GetGrouping(sa.CSR(0, 4, [0], np.zeros(0, np.int32), None)).tolist() == []
    # the other measures / options run too (Hamming, cluster weights, no pattern merging)
    for kw in (dict(sim_measure=0, tau=3.0), dict(use_groups=True, tau=0.5), dict(use_pattern=False, tau=0.5)):
        gk = sa.BlockingEngine(blocking_algo=7, col_block_size=4, **kw).GetGrouping(m2)
        assert len(gk) == 400 and (gk <= np.arange(400)).all()


def test_minhash_reorder_equals_the_exact_algorithm_at_tau_zero():
    """tau = 0 merges exactly the rows with the same block set as the seed: no approximation is left, algorithm 7 must return
    algorithm 3's grouping (and the same merge count) -- with empty rows, repeated rows and every block width"""
    rng = np.random.Generator(np.random.PCG64(77))
    for case in range(40):
        rows, cols = int(rng.integers(1, 400)), int(rng.integers(1, 60))
        m = sa.gen.uniform_random(rows, cols, int(rows * cols * float(rng.choice([0.02, 0.1, 0.4]))), seed=int(rng.integers(1 << 30)), pattern_only=True)
        w = int(rng.choice([1, 2, 5, 16]))
        e3 = sa.BlockingEngine(blocking_algo=3, tau=0.0, col_block_size=w)
        e7 = sa.BlockingEngine(blocking_algo=7, tau=0.0, col_block_size=w)
        g3, g7 = e3.GetGrouping(m), e7.GetGrouping(m)
        assert np.array_equal(g3, g7), (case, rows, cols, w)
        assert e3.merge_counter == e7.merge_counter


def test_minhash_regression_fixture():
    """blocking_algo 7 has no reference; tests/golden/minhash_regression.npz pins what it returns for seeded inputs (regression only)"""
    import hashlib
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_minhash", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden_minhash.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "minhash_regression.npz"))
    for i, c in enumerate(mod.CASES):
        g, cmp_, mrg = mod.run(c)
        assert bytes(fx["sha%d" % i]) == hashlib.sha256(g.tobytes()).digest(), c
        assert fx["stat%d" % i].tolist() == [cmp_, mrg, len(np.unique(g))]


# ---- VBR::fill_from_CSR / get_block_start / partition_check (vbr.cpp:239-321, 33-49, 108-118) --------------------------------
def _random_partition(rng, rows, n_cuts, repeats=0):
    cuts = np.sort(rng.choice(np.arange(1, rows), size=min(n_cuts, rows - 1), replace=False))
    part = np.concatenate([[0], cuts, [rows]]).astype(np.int64)
    if repeats:                                   # repeated entries = block-rows of height 0
        part = np.sort(np.concatenate([part, rng.choice(part, size=repeats)]))
    return part


@pytest.mark.parametrize("name,bs", [("u256", 16), ("rect", 7), ("band1k", 32), ("fem", 3)])
def test_fill_from_csr_partition_matches_the_compiled_reference(name, bs):
    from oracle import ref
    if not ref.available():
        pytest.skip("compiled reference not present")
    m = U.matrices()[name]
    rng = np.random.default_rng(5)
    for trial in range(3):
        part = _random_partition(rng, m.rows, 9 + 20 * trial, repeats=2 * trial)
        vb = sa.VBR().fill_from_CSR(m, part, bs)
        rc = ref.RefCSR(m.rows, m.cols, m.rowptr, m.colidx.astype(np.int64), m.vals)
        rv = ref.RefVBR(rc, None, bs, row_partition=part)
        rp, nz, jab, mab = rv.export()
        assert (vb.rows, vb.cols, vb.block_rows, vb.block_cols, vb.block_col_size, vb.nztot) == \
               (rv.rows, rv.cols, rv.block_rows, rv.block_cols, rv.block_col_size, rv.nztot)
        assert np.array_equal(vb.row_part, rp) and np.array_equal(vb.nzcount, nz) and np.array_equal(vb.jab, jab)
        assert np.array_equal(vb.mab, mab)
        for ib in (0, 1, vb.block_rows // 2, vb.block_rows - 1, vb.block_rows, vb.block_rows + 3):
            assert vb.get_block_start(ib) == rv.block_start(ib), ib
        for cand in (part, part[:-1], part[::-1].copy(), np.zeros(0, np.int64), np.array([0, m.rows], np.int64)):
            assert vb.partition_check(cand) == rv.partition_check(cand)


def test_fill_from_csr_partition_properties():
    """no compiled reference needed: the partition build holds exactly the matrix (rows in place, column-major blocks, vbr.cpp:307),
    and bad partitions are refused"""
    m = U.matrices()["u256"]
    part = np.arange(0, m.rows + 1, 8, dtype=np.int64)
    w = 16
    vb = sa.VBR().fill_from_CSR(m, part, w)
    dense = np.zeros((m.rows, m.cols), np.float32)
    for i in range(m.rows):
        dense[i, m.colidx[m.rowptr[i]:m.rowptr[i + 1]]] = m.vals[m.rowptr[i]:m.rowptr[i + 1]]
    back = np.zeros_like(dense)
    jo = mo = 0
    for ib in range(vb.block_rows):
        r0, r1 = int(vb.row_part[ib]), int(vb.row_part[ib + 1])
        for b in range(int(vb.nzcount[ib])):
            jb = int(vb.jab[jo + b])
            blk = vb.mab[mo:mo + (r1 - r0) * w].reshape(w, r1 - r0).T          # column-major h x w
            back[r0:r1, jb * w:(jb + 1) * w] = blk[:, :min(w, m.cols - jb * w)]
            mo += (r1 - r0) * w
        jo += int(vb.nzcount[ib])
    assert np.array_equal(back, dense) and mo == vb.nztot
    assert vb.partition_check(part) == 0 and vb.partition_check(part[:-1]) == 2 and vb.partition_check(part[::-1].copy()) == 2
    assert vb.partition_check(np.array([0, 9, 5, m.rows], np.int64)) == 3 and vb.partition_check(np.zeros(0, np.int64)) == 1
    with pytest.raises(sa.SpartaError):
        sa.VBR().fill_from_CSR(m, part[:-1], 16)
