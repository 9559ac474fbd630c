"""CPU: the oracle (plain-C restatement) against the reference's own known answers and the golden vectors that
tests/golden/make_golden.py generated from the compiled reference.  This is what PINS the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
import _util as U

ORACLE_ALGOS = (0, 2, 3, 4, 5)


def test_similarity_kat():
    # test/general/TEST_similarities.cpp:14-22 (values: SURVEY.md section 4, verified by run)
    A, B = [1, 2, 5, 10, 12, 20], [0, 2, 4, 10, 16]
    assert O.distance(0, A, 1, B, 1, 1) == 7.0
    assert O.distance(1, A, 1, B, 1, 1) == pytest.approx(0.777778, abs=1e-6)
    for w in (3, 4):
        assert O.distance(0, A, 1, B, 1, w) == 3.0
        assert O.distance(1, A, 1, B, 1, w) == 0.5
    # empty-row rules (blocking.cpp:863-864, 926-927)
    assert O.distance(1, [], 1, [], 1, 4) == 0.0 and O.distance(1, [], 1, [3], 1, 4) == 1.0
    assert O.distance(0, [], 2, [3, 9], 3, 4) == 6.0 and O.distance(0, [1, 2, 3], 2, [], 3, 4) == 6.0


def test_merge_rows_is_lossy():
    # src/general/utilities.cpp:145-173 (verified by run)
    assert O.merge_rows([1, 2, 3], [5]).tolist() == [5]
    assert O.merge_rows([1, 4, 9], [2, 4, 7]).tolist() == [1, 2, 4, 7]
    assert O.merge_rows([1, 4, 9], [2, 4, 7, 12]).tolist() == [1, 2, 4, 7, 12]
    assert O.merge_rows([1, 4, 9], []).tolist() == []
    assert O.merge_rows([], [2, 3]).tolist() == [2, 3]


def test_appendix_b_inline_kat():
    """-b 3 -t 0.6 on data/TEST_matrix_weighted.el (first data line dropped by the reader)."""
    k = U.load("kat9.npz")
    rows, cols = int(k["rows"]), int(k["cols"])
    assert (rows, cols, len(k["colidx"])) == (9, 9, 12)
    g, cnt = O.get_grouping(rows, k["rowptr"], k["colidx"], 3, 1, 0.6, 3, 3)
    assert g.tolist() == [0, 1, 1, 1, 0, 5, 0, 0, 8] == k["g_b3_t06"].tolist()
    assert cnt == dict(comparison_counter=13, merge_counter=5)
    assert O.get_permutation(g).tolist() == [0, 4, 6, 7, 1, 2, 3, 5, 8]
    v = O.OracleVBR(rows, cols, k["rowptr"], k["colidx"], k["vals"], g, 3)
    assert (v.rows, v.cols, v.block_rows, v.block_cols, v.nztot) == (9, 9, 4, 3, 33)
    assert v.row_part.tolist() == [0, 4, 7, 8, 9] and v.nzcount.tolist() == [0, 3, 1, 1] and v.jab.tolist() == [0, 1, 2, 2, 0]
    want_mab = [0, 0, 0, 0, 0, 1, 5, 0, 0, 0, 0, 1, 0, 0, 0, 8, 1, 0, 0, 1, 0, 0, 0, 3, 7, 1, 8, 2, 0, 0, 0, 5, 0]
    assert v.mab.tolist() == want_mab == k["mab"].tolist()
    C = O.vbr_multiply(9, 9, 3, v.row_part, v.nzcount, v.jab, v.mab, np.arange(1, 19, dtype=np.float32), 2)
    assert C.tolist() == [0, 0, 0, 0, 126, 22, 102, 14, 10, 0, 0, 0, 0, 306, 49, 219, 32, 55] == k["C_B1to18"].tolist()
    info = O.collect_blocking_info(rows, cols, k["rowptr"], k["colidx"], g, 3)
    assert (info["VBR_nzcount"], info["VBR_nzblocks_count"], info["VBR_longest_row"]) == (33, 5, 3)
    assert info["VBR_average_height"] == pytest.approx(2.2)
    gF, _ = O.get_grouping(rows, k["rowptr"], k["colidx"], 3, 1, 0.6, 3, 3, force_fixed_size=True)
    assert gF.tolist() == [0, 1, 1, 2, 0, 2, 0, 1, 2] == k["g_F1_B3"].tolist()
    gK, _ = O.get_grouping(rows, k["rowptr"], k["colidx"], 5, 1, 0.6, 3, 3, force_fixed_size=True)     # -a 5 -B 3 -F 1
    assert gK.tolist() == [0, 1, 1, 1, 0, 2, 0, 2, 2] == k["g_a5_B3_F1"].tolist()


def test_vbr_equals_csr_multiply_fixed_blocking():
    """test/general/TEST_matrices.cpp:44-54: VBR::multiply == CSR::multiply bit-exactly, fixed blocking, B = ones."""
    k = U.load("kat9.npz")
    g, _ = O.get_grouping(9, k["rowptr"], k["colidx"], 2, 1, 0.6, 3, 3)
    v = O.OracleVBR(9, 9, k["rowptr"], k["colidx"], k["vals"], g, 3)
    ones = np.ones(45, np.float32)
    Cv = O.vbr_multiply(9, 9, 3, v.row_part, v.nzcount, v.jab, v.mab, ones, 5)
    Cc = O.csr_multiply(9, k["rowptr"], k["colidx"], k["vals"], ones, 9, 5)
    assert np.array_equal(Cv, Cc) and np.array_equal(Cv, k["C_fixed_ones_vbr"]) and np.array_equal(Cc, k["C_fixed_ones_csr"])


def test_primitives_golden():
    p = U.load("prims.npz")
    for t in range(int(p["n"])):
        A, B = p["%d/A" % t], p["%d/B" % t]
        ga, gb, bs = map(int, p["%d/par" % t])
        assert np.array_equal(O.merge_rows(A, B), p["%d/merged" % t]), t
        d = p["%d/dist" % t]
        assert np.float32(O.distance(0, A, ga, B, gb, bs)) == d[0] and np.float32(O.distance(1, A, ga, B, gb, bs)) == d[1], t
    for t in range(int(p["nperm"])):
        g = p["perm%d/g" % t]
        assert np.array_equal(O.get_permutation(g), p["perm%d/perm" % t]), "introsort tie order, n=%d" % len(g)
        assert np.array_equal(O.get_partition(g), p["perm%d/part" % t])
        assert np.array_equal(O.get_fixed_size_grouping(g, 5), p["perm%d/fixed5" % t])


@pytest.mark.parametrize("key,name,cfg", U.case_list(), ids=[c[0] for c in U.case_list()])
def test_case_golden(key, name, cfg):
    m = U.matrices()[name]
    f = U.case_fields(key)
    w, rbs, ff = cfg["w"], cfg.get("rbs", 1), cfg.get("ff", False)
    g = f["grouping"].astype(np.int64)
    if cfg["algo"] in ORACLE_ALGOS:
        go, cnt = O.get_grouping(m.rows, m.rowptr, m.colidx, cfg["algo"], cfg.get("sim", 1), cfg["tau"], w, rbs,
                                 cfg.get("use_groups", False), cfg.get("use_pattern", True), ff)
        assert np.array_equal(go, g)
        if cfg["algo"] != 2:
            assert [cnt["comparison_counter"], cnt["merge_counter"]] == f["counters"].tolist()
    # everything downstream of the grouping is restated for every algorithm
    assert np.array_equal(O.get_permutation(g), f["perm"])
    v = O.OracleVBR(m.rows, m.cols, m.rowptr, m.colidx, m.vals, g, w, rbs, ff)
    assert [v.rows, v.cols, v.block_rows, v.block_cols, v.nztot] == f["dims"].tolist()
    assert np.array_equal(v.row_part, f["row_part"]) and np.array_equal(v.nzcount, f["nzcount"]) and np.array_equal(v.jab, f["jab"])
    assert U.sha(v.mab) == str(f["mab_sha"])
    import sparta_amd as sa
    n = f["C"].size // v.rows
    B = sa.gen.dense_rhs(v.cols, n, seed=77)
    C = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
    assert np.array_equal(C, f["C"]), "oracle VBR::multiply is not bit-identical to the reference"
    info = O.collect_blocking_info(m.rows, m.cols, m.rowptr, m.colidx, g, w)
    assert [info["VBR_nzcount"], info["VBR_nzblocks_count"], info["VBR_longest_row"]] == f["info"].tolist()
    assert np.float32(info["VBR_average_height"]) == f["avg_height"]
    # C_vbs[r] == C_csr[perm[r]] (bit-exact: the VBS adds exact zeros in the same column order)
    if m.rows == m.cols and not ff:
        Cc = O.csr_multiply(m.rows, m.rowptr, m.colidx, m.vals, B, m.cols, n).reshape(n, m.rows)
        assert np.array_equal(C.reshape(n, v.rows), Cc[:, f["perm"].astype(np.int64)])
