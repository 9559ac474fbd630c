"""On-disk formats either side of the hot path: edge-list / MatrixMarket readers, edge-list
writer, grouping file, the 32-column CSV row, reorder_by_degree.

Three-way parity: product (C-ABI, through `sparta_amd`) == Python restatement (oracle/io_oracle.py) == the compiled
reference (golden vectors in tests/golden/io.npz made by tests/golden/make_golden_io.py; live where oracle/_ref exists).
"""
import ast
import os

import numpy as np
import pytest

import sparta_amd as sa
from oracle import io_oracle as IO
from oracle import oracle as O
from oracle import ref
import _util as U

Z = U.load("io.npz")
REF_DATA = os.path.join(U.GOLDEN, "ref_data")
EL_CASES = sorted({k.split("/")[1] for k in Z.files if k.startswith("el/")})
MTX_CASES = sorted({k.split("/")[1] for k in Z.files if k.startswith("mtx/")})


def _same_csr(m, rows, cols, rowptr, colidx, vals):
    assert (m.rows, m.cols) == (int(rows), int(cols))
    assert np.array_equal(m.rowptr, rowptr)
    assert np.array_equal(m.colidx, colidx)
    if vals is None:
        assert m.vals is None
    else:
        assert m.vals is not None and np.array_equal(m.vals, np.asarray(vals, np.float32))


# ---- the reference's own fixture ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("po", [0, 1])
def test_reference_fixture_is_read_like_the_reference_reads_it(po):
    """data/TEST_matrix_weighted.el has 13 lines; the reference's reader drops the first one: 9 x 9, 12 nonzeros"""
    path = os.path.join(REF_DATA, "TEST_matrix_weighted.el")
    want = (Z["fixture/po%d/dims" % po][0], Z["fixture/po%d/dims" % po][1], Z["fixture/po%d/rowptr" % po],
            Z["fixture/po%d/colidx" % po], None if po else Z["fixture/po%d/vals" % po])
    assert int(want[2][-1]) == 12
    _same_csr(sa.CSR.read_from_edgelist(path, pattern_only=bool(po)), *want)
    r = IO.read_el(open(path).read(), " ", bool(po), False)
    assert (r[0], r[1]) == (int(want[0]), int(want[1])) and np.array_equal(r[2], want[2]) and np.array_equal(r[3], want[3])
    if not po:
        assert np.array_equal(r[4], want[4])
    strict = sa.CSR.read_from_edgelist(path, pattern_only=bool(po), mode=sa.IO_STRICT)
    assert strict.nztot() == 13                     # documented format: nothing dropped


def test_kat9_fixture_matches_the_golden_matrix_of_the_other_tests():
    k = U.load("kat9.npz")
    m = sa.CSR.read_from_edgelist(os.path.join(REF_DATA, "TEST_matrix_weighted.el"))
    assert np.array_equal(m.rowptr, k["rowptr"]) and np.array_equal(m.colidx, k["colidx"]) and np.array_equal(m.vals, k["vals"])


# ---- edge lists: golden (compiled reference) == oracle == product --------------------------------------------------------
@pytest.mark.parametrize("name", EL_CASES)
def test_edge_list_reader_and_writer(name, tmp_path):
    text = str(Z["el/%s/text" % name])
    delim, po, sym = [str(x) for x in Z["el/%s/args" % name]]
    po, sym = bool(int(po)), bool(int(sym))
    rows, cols = Z["el/%s/dims" % name]
    rowptr, colidx = Z["el/%s/rowptr" % name], Z["el/%s/colidx" % name]
    vals = None if po else Z["el/%s/vals" % name]
    r = IO.read_el(text, delim, po, sym)
    assert (r[0], r[1]) == (int(rows), int(cols)) and np.array_equal(r[2], rowptr) and np.array_equal(r[3], colidx)
    if not po:
        assert np.array_equal(r[4], vals)
    p = tmp_path / (name + ".el")
    p.write_text(text)
    m = sa.CSR.read_from_edgelist(p, delim, po, sa.FMT_EL, sym)
    _same_csr(m, rows, cols, rowptr, colidx, vals)
    w = tmp_path / (name + ".out")
    m.save_to_edgelist(w, delim)
    saved = str(Z["el/%s/saved" % name])
    assert w.read_text() == saved == IO.save_to_edgelist(m.rows, m.rowptr, m.colidx, delim)
    if ref.available():
        live = ref.RefCSR.read(str(p), delim, po, 0, sym)
        rp, ci, v = live.export()
        _same_csr(m, live.rows, live.cols, rp, ci, None if po else v)


def test_edge_list_round_trip_through_the_writer(tmp_path):
    """save_to_edgelist writes no header: reading it back in the reference's mode loses the first entry, in strict mode nothing"""
    m = sa.gen.uniform_random(50, 60, 300, seed=3)
    p = tmp_path / "rt.el"
    m.save_to_edgelist(p)
    back = sa.CSR.read_from_edgelist(p, pattern_only=True, mode=sa.IO_STRICT)
    assert back.nztot() == m.nztot() and np.array_equal(back.colidx, m.colidx)
    assert np.array_equal(back.rowptr[:back.rows + 1], m.rowptr[:back.rows + 1])
    compat = sa.CSR.read_from_edgelist(p, pattern_only=True)
    assert compat.nztot() == m.nztot() - 1
    mt = tmp_path / "rt_mtx.el"
    m.save_to_edgelist(mt, mat_fmt=sa.FMT_MTX)                  # "j i" per entry (csr.cpp:175)
    first = mt.read_text().split("\n")[0].split()
    assert [int(first[0]), int(first[1])] == [int(m.colidx[0]), 0 if m.rowptr[1] > 0 else int(np.searchsorted(m.rowptr, 1, side="right") - 1)]


@pytest.mark.parametrize("text,po,why", [
    ("h\n3 1 1.0\n2 0 1.0\n", False, "ascending"),              # std::invalid_argument in the reference (csr.cpp:259)
    ("h\n\n1 2 3\n", False, "bad row id"),                      # stoi("") throws
    ("h\nx 2 3\n", False, "bad row id"),
    ("h\n1 y 3\n", False, "bad column id"),
    ("h\n1 2 z\n", False, "bad value"),
    ("h\n-1 2 3\n", False, "negative"),
])
def test_inputs_the_reference_aborts_on_are_errors_not_crashes(tmp_path, text, po, why):
    p = tmp_path / "bad.el"
    p.write_text(text)
    with pytest.raises(sa.SpartaError) as e:
        sa.CSR.read_from_edgelist(p, pattern_only=po)
    assert why in str(e.value)
    with pytest.raises(IO.RefUndefined):
        IO.read_el(text, " ", po, False)


def test_reader_argument_errors(tmp_path):
    with pytest.raises(sa.SpartaError):
        sa.CSR.read_from_edgelist(tmp_path / "does_not_exist.el")
    p = tmp_path / "w.el"
    p.write_text("h\n0 1 2.0\n1 2 3.0\n2 2 1.0\n")
    with pytest.raises(sa.SpartaError) as e:                    # symmetrize is only defined for patterns (csr.cpp:277-280)
        sa.CSR.read_from_edgelist(p, symmetrize=True, pattern_only=False)
    assert "unweighted" in str(e.value)
    with pytest.raises(IO.RefUndefined):
        IO.read_el(p.read_text(), " ", False, True)
    ok = sa.CSR.read_from_edgelist(p, symmetrize=True, pattern_only=True)
    assert ok.to_scipy().toarray().tolist() == [[0, 1, 0], [1, 0, 1], [0, 1, 1]]
    with pytest.raises(sa.SpartaError):
        sa.CSR.read_from_edgelist(p, mat_fmt=7)
    with pytest.raises(sa.SpartaError):
        sa.CSR.read_from_edgelist(p, mode=9)


# ---- MatrixMarket ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", MTX_CASES)
def test_matrix_market_reference_mode(name, tmp_path):
    text = str(Z["mtx/%s/text" % name])
    rows, cols = Z["mtx/%s/dims" % name]
    rowptr, colidx = Z["mtx/%s/rowptr" % name], Z["mtx/%s/colidx" % name]
    r = IO.read_mtx(text)
    assert (r[0], r[1]) == (int(rows), int(cols)) and np.array_equal(r[2], rowptr) and np.array_equal(r[3], colidx)
    p = tmp_path / (name + ".mtx")
    p.write_text(text)
    _same_csr(sa.CSR.read_from_edgelist(p, mat_fmt=sa.FMT_MTX), rows, cols, rowptr, colidx, None)


def test_matrix_market_standard_file_in_both_modes(tmp_path):
    text = ("%%MatrixMarket matrix coordinate real symmetric\n% a comment\n4 4 4\n1 1 2.0\n2 1 -1.0\n3 2 0.5\n4 4 7\n")
    p = tmp_path / "s.mtx"
    p.write_text(text)
    # reference mode: one entry line is skipped and the last read runs dry -> undefined in the reference, an error here
    with pytest.raises(sa.SpartaError) as e:
        sa.CSR.read_from_edgelist(p, mat_fmt=sa.FMT_MTX)
    assert "announced" in str(e.value)
    with pytest.raises(IO.RefUndefined):
        IO.read_mtx(text)
    m = sa.CSR.read_from_edgelist(p, mat_fmt=sa.FMT_MTX, mode=sa.IO_STRICT)
    d = m.to_scipy().toarray()
    want = np.array([[2, -1, 0, 0], [-1, 0, 0.5, 0], [0, 0.5, 0, 0], [0, 0, 0, 7]], np.float32)
    assert np.array_equal(d, want)
    mp = sa.CSR.read_from_edgelist(p, mat_fmt=sa.FMT_MTX, mode=sa.IO_STRICT, pattern_only=True)
    assert mp.vals is None and mp.nztot() == 6
    (tmp_path / "p.mtx").write_text("%%MatrixMarket matrix coordinate pattern general\n2 3 2\n1 3\n2 1\n")
    g = sa.CSR.read_from_edgelist(tmp_path / "p.mtx", mat_fmt=sa.FMT_MTX, mode=sa.IO_STRICT)
    assert g.vals is None and (g.rows, g.cols) == (2, 3) and list(g.colidx) == [2, 0]
    (tmp_path / "short.mtx").write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1\n")
    with pytest.raises(sa.SpartaError):
        sa.CSR.read_from_edgelist(tmp_path / "short.mtx", mat_fmt=sa.FMT_MTX, mode=sa.IO_STRICT)


# ---- grouping file -----------------------------------------------------------------------------------------------------------
def test_grouping_file_of_the_reference_and_round_trip(tmp_path):
    want = np.array([3, 1, 0, 2], np.int64)                      # data/TEST/TEST.g
    path = os.path.join(REF_DATA, "TEST.g")
    assert np.array_equal(sa.read_grouping_file(path), want)
    assert np.array_equal(IO.read_grouping_file(open(path).read()), want)
    assert np.array_equal(sa.read_grouping_file(path, rows=4), want)
    assert np.array_equal(sa.read_grouping_file(path, rows=3), want[1:])       # Matrix_Analysis.cpp:78: leading count dropped
    assert np.array_equal(IO.read_grouping_file(open(path).read(), rows=3), want[1:])
    with pytest.raises(sa.SpartaError):
        sa.read_grouping_file(path, rows=9)
    g = np.array([5, 5, 0, -1, 7, 123456789012], np.int64)[:5]
    p = tmp_path / "x.g"
    sa.save_grouping(p, g)
    assert p.read_text() == IO.grouping_file(g) == "5\n5\n0\n-1\n7\n"
    assert np.array_equal(sa.read_grouping_file(p, rows=5), g)
    (tmp_path / "junk.g").write_text("4\nabc\n 2 trailing\n\n1\n")
    assert list(sa.read_grouping_file(tmp_path / "junk.g")) == [4, 2, 1] == list(IO.read_grouping_file("4\nabc\n 2 trailing\n\n1\n"))


def test_grouping_file_feeds_the_vbs_builder(tmp_path):
    m = sa.gen.uniform_random(300, 300, 3000, seed=8)
    eng = sa.BlockingEngine(tau=0.5, col_block_size=16)
    g = eng.GetGrouping(m)
    sa.save_grouping(tmp_path / "m.g", g)
    g2 = sa.read_grouping_file(tmp_path / "m.g", rows=m.rows)
    a, b = sa.VBR().fill_from_CSR_inplace(m, g, 16), sa.VBR().fill_from_CSR_inplace(m, g2, 16)
    assert np.array_equal(a.jab, b.jab) and np.array_equal(a.mab, b.mab) and np.array_equal(a.row_part, b.row_part)


# ---- CSV row -------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [0, 1, 2])
def test_csv_row_matches_save_blocking_data(k, tmp_path):
    kw = ast.literal_eval(str(Z["csv/%d/kwargs" % k]))
    want_csv, want_g = str(Z["csv/%d/csv" % k]), str(Z["csv/%d/gfile" % k])
    header_want, values_want = want_csv.split("\n")[0], want_csv.split("\n")[1]
    assert header_want.count(",") == 32 and values_want.count(",") == 32
    m = sa.CSR.read_from_edgelist(os.path.join(REF_DATA, "TEST_matrix_weighted.el"))
    eng = sa.BlockingEngine(tau=kw["tau"], col_block_size=kw["col_block_size"], row_block_size=kw["row_block_size"],
                            use_groups=bool(kw.get("sim_use_groups", 0)), use_pattern=bool(kw.get("use_pattern", 1)),
                            force_fixed_size=bool(kw.get("force_fixed_size", 0)), blocking_algo=kw["blocking_algo"],
                            sim_measure=kw.get("sim_measure", 1))
    g = eng.GetGrouping(m)
    assert np.array_equal(g, Z["csv/%d/grouping" % k])
    eng.timer_total, eng.timer_merges, eng.timer_comparisons = 1234.5, 77.25, 901.0       # the generator pinned the clocks
    eng.multiplication_timer_avg, eng.multiplication_timer_std = 0.125 * (k + 1), 0.001
    out = tmp_path / "res.txt"
    header, values = sa.save_blocking_data(out, eng, m, g, matrix="data/TEST_matrix_weighted.el", exp_name="exp%d" % k,
                                           symmetrize=kw.get("symmetrize", 0), reorder=kw.get("reorder", 0), b_cols=kw.get("b_cols", 1024),
                                           warmup=1, exp_repetitions=kw.get("exp_repetitions", 5),
                                           multiplication_algo=kw.get("multiplication_algo", 0), n_streams=kw.get("n_streams", 4))
    assert header == header_want
    assert values == values_want
    assert out.read_text() == want_csv
    assert (tmp_path / "res.txt.g").read_text() == want_g == IO.grouping_file(g)
    # the restatement formats the same numbers the same way
    vals = dict(zip(header_want.rstrip(",").split(","), values_want.rstrip(",").split(",")))
    typed = {c: (vals[c] if c in ("matrix", "exp_name") else float(vals[c]) if "." in vals[c] else int(vals[c])) for c in IO.CSV_COLUMNS}
    assert IO.csv_row(**typed) == (header_want, values_want)
    assert tuple(header_want.rstrip(",").split(",")) == sa.CSV_COLUMNS == IO.CSV_COLUMNS


def test_csv_row_rejects_unknown_columns():
    with pytest.raises(TypeError):
        sa.blocking_csv_row(not_a_column=1)


# ---- reorder_by_degree ------------------------------------------------------------------------------------------------------------
DEG_CASES = sorted({tuple(k.split("/")[1:3]) for k in Z.files if k.startswith("deg/")})


@pytest.mark.parametrize("name,desc", DEG_CASES)
def test_reorder_by_degree(name, desc):
    deg = Z["deg/%s/%s/degrees" % (name, desc)]
    descending = desc == "desc1"
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    colidx = np.concatenate([np.arange(d) for d in deg] + [np.zeros(0, int)]).astype(np.int32)
    tag = np.repeat(np.arange(len(deg)), deg).astype(np.float32)
    m = sa.CSR(len(deg), int(max(deg.max(), 1)), rowptr, colidx, tag)
    m.reorder_by_degree(descending)
    assert np.array_equal(m.rowptr, Z["deg/%s/%s/new_rowptr" % (name, desc)])
    want = Z["deg/%s/%s/perm_where_nonempty" % (name, desc)]
    got = np.array([int(m.vals[m.rowptr[k]]) if m.rowptr[k + 1] > m.rowptr[k] else -1 for k in range(m.rows)])
    assert np.array_equal(got, want)
    perm = IO.degree_permutation(rowptr, descending, O.get_permutation)
    assert np.array_equal(np.diff(rowptr)[perm], np.diff(m.rowptr))
    assert np.array_equal(np.where(np.diff(rowptr)[perm] > 0, perm, -1), want)


def test_permute_rows_and_reorder_match_the_reference_semantics():
    m = sa.gen.uniform_random(120, 90, 900, seed=21)
    g = sa.BlockingEngine(tau=0.6, col_block_size=8).GetGrouping(m)
    perm = sa.get_permutation(g)
    d0 = m.to_scipy().toarray()
    m.reorder(g)
    assert np.array_equal(m.to_scipy().toarray(), d0[perm])
    if ref.available():
        m0 = sa.gen.uniform_random(120, 90, 900, seed=21)
        r = ref.RefCSR(m0.rows, m0.cols, m0.rowptr, m0.colidx, m0.vals)
        r.reorder(g)
        rp, ci, v = r.export()
        assert np.array_equal(rp, m.rowptr) and np.array_equal(ci, m.colidx) and np.array_equal(v, m.vals)
    with pytest.raises(ValueError):
        m.permute_rows(np.arange(5))


@pytest.mark.skipif(not ref.available(), reason="compiled reference not present")
def test_random_edge_lists_against_the_live_reference(tmp_path):
    rng = np.random.Generator(np.random.PCG64(2024))
    for case in range(40):
        rows, cols, nnz = int(rng.integers(1, 60)), int(rng.integers(1, 60)), int(rng.integers(0, 300))
        delim = [" ", ",", "\t", "::"][case % 4]
        po = bool(case % 3 == 0)
        rr = np.sort(rng.integers(0, rows, nnz))
        lines = ["whatever first line\n"]
        for r in rr:
            c = int(rng.integers(0, cols))
            lines.append("%d%s%d\n" % (r, delim, c) if po else "%d%s%d%s%.3f\n" % (r, delim, c, delim, rng.normal()))
        p = tmp_path / ("r%d.el" % case)
        p.write_text("".join(lines))
        live = ref.RefCSR.read(str(p), delim, po, 0, False)
        rp, ci, v = live.export()
        _same_csr(sa.CSR.read_from_edgelist(p, delim, po), live.rows, live.cols, rp, ci, None if po else v)
        o = IO.read_el(p.read_text(), delim, po, False)
        assert np.array_equal(o[2], rp) and np.array_equal(o[3], ci) and (po or np.array_equal(o[4], v))


# ---- binary VBS container ------------------------------------------------------------------------------------------------------
def _fnv1a(chunks):
    h = 14695981039346656037
    for c in chunks:
        for b in c.tobytes():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_vbs_container_round_trip_and_layout(tmp_path):
    import struct
    m = sa.gen.uniform_random(90, 130, 700, seed=4)
    g = sa.BlockingEngine(tau=0.6, col_block_size=16).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, 16)
    p = tmp_path / "a.vbs"
    v.save(p)
    back = sa.VBR.load(p)
    for name in ("rows", "cols", "block_rows", "block_cols", "block_col_size", "nztot"):
        assert getattr(back, name) == getattr(v, name), name
    for name in ("row_part", "nzcount", "jab", "mab"):
        assert np.array_equal(getattr(back, name), getattr(v, name)), name
    # the documented layout, parsed independently
    raw = p.read_bytes()
    assert raw[:8] == b"SPARTAVB" and struct.unpack_from("<II", raw, 8) == (1, 96)
    dims = struct.unpack_from("<8q", raw, 16)
    assert dims[:7] == (v.rows, v.cols, v.block_rows, v.block_cols, v.block_col_size, v.nztot, len(v.jab))
    chk, payload = struct.unpack_from("<QQ", raw, 80)
    assert payload == len(raw) - 96 == 8 * (v.block_rows + 1) + 8 * v.block_rows + 8 * len(v.jab) + 4 * v.nztot
    assert chk == _fnv1a([v.row_part, v.nzcount, v.jab, v.mab])
    off = 96
    assert np.array_equal(np.frombuffer(raw, "<i8", v.block_rows + 1, off), v.row_part)
    off += 8 * (v.block_rows + 1) + 8 * v.block_rows + 8 * len(v.jab)
    assert np.array_equal(np.frombuffer(raw, "<f4", v.nztot, off), v.mab)


def test_vbs_container_rejects_damage(tmp_path):
    m = sa.gen.uniform_random(40, 40, 200, seed=6)
    v = sa.VBR().fill_from_CSR_inplace_fixed(m, 8, 8)
    p = tmp_path / "a.vbs"
    v.save(p)
    raw = bytearray(p.read_bytes())
    cases = {"flip": bytes(raw[:200]) + bytes([raw[200] ^ 1]) + bytes(raw[201:]), "trunc": bytes(raw[:-5]), "magic": b"NOTAVBS!" + bytes(raw[8:]),
             "short": bytes(raw[:40]), "version": bytes(raw[:8]) + (2).to_bytes(4, "little") + bytes(raw[12:])}
    for name, data in cases.items():
        q = tmp_path / (name + ".vbs")
        q.write_bytes(data)
        with pytest.raises(sa.SpartaError):
            sa.VBR.load(q)
    with pytest.raises(sa.SpartaError):
        sa.VBR.load(tmp_path / "missing.vbs")
    empty = sa.VBR.from_arrays(5, 7, 4, np.array([0, 5]), np.array([0]), np.zeros(0, np.int64), np.zeros(0, np.float32))
    empty.save(tmp_path / "e.vbs")
    e2 = sa.VBR.load(tmp_path / "e.vbs")
    assert e2.rows == 5 and e2.nztot == 0 and len(e2.jab) == 0


# ---- Blocked-ELL view (prepare_cusparse_BLOCKEDELLPACK, cuda_utilities.cpp:1656-1710) ---------------------------------------------

@pytest.mark.parametrize("rows,cols,nnz,bs", [(24, 36, 60, 4), (64, 64, 300, 8), (30, 18, 0, 6), (16, 48, 200, 16)])
def test_blocked_ell_view(rows, cols, nnz, bs):
    m = sa.gen.uniform_random(rows, cols, nnz, seed=rows + cols + bs)
    v = sa.VBR().fill_from_CSR_inplace_fixed(m, bs, bs)
    b, ind, val = v.to_blocked_ell()
    bo, indo, valo = IO.blocked_ell(v.rows, v.cols, bs, v.nzcount, v.jab, v.mab)
    assert b == bo == bs and np.array_equal(ind, indo) and np.array_equal(val, valo)
    # the Blocked-ELL arrays expand to the matrix itself; padding blocks are -1 / zeros
    dense = np.zeros((rows, cols), np.float32)
    for i in range(rows):
        dense[i, m.colidx[m.rowptr[i]:m.rowptr[i + 1]]] = m.vals[m.rowptr[i]:m.rowptr[i + 1]]
    back = np.zeros((v.rows, v.cols), np.float32)
    for k in range(ind.shape[0]):
        for j in range(ind.shape[1]):
            if ind[k, j] >= 0:
                back[k * bs:(k + 1) * bs, ind[k, j] * bs:(ind[k, j] + 1) * bs] = val[k * bs:(k + 1) * bs, j * bs:(j + 1) * bs]
            else:
                assert not val[k * bs:(k + 1) * bs, j * bs:(j + 1) * bs].any()
    perm = sa.get_permutation(np.arange(rows) // bs)                  # rows of a VBS are in get_permutation order (unstable sort: even a fixed grid permutes inside its groups)
    assert np.array_equal(back, dense[perm])
    assert ind.shape == (rows // bs, int(v.nzcount.max()) if len(v.nzcount) else 0)
    for k in range(ind.shape[0]):
        assert (ind[k] >= 0).sum() == v.nzcount[k] and list(ind[k][:v.nzcount[k]]) == sorted(ind[k][:v.nzcount[k]])


def test_blocked_ell_view_rejects_what_the_reference_cannot_handle():
    m = sa.gen.uniform_random(30, 20, 80, seed=3)
    # sizes that are not multiples of the block size: the reference prints and exits (:1666-1672)
    with pytest.raises(sa.SpartaError) as ei:
        sa.VBR().fill_from_CSR_inplace_fixed(m, 4, 4).to_blocked_ell()
    assert "multiple of ell_blocksize" in str(ei.value)
    with pytest.raises(IO.RefUndefined):
        IO.blocked_ell(30, 20, 4, [], [], [])
    # block-rows that are not bs tall: the reference indexes as if they were (garbage); refused here
    g = sa.BlockingEngine(tau=0.5, col_block_size=4).GetGrouping(sa.gen.uniform_random(32, 20, 80, seed=3))
    with pytest.raises(sa.SpartaError):
        sa.VBR().fill_from_CSR_inplace(sa.gen.uniform_random(32, 20, 80, seed=3), g, 4).to_blocked_ell()
    # force_fixed_size pads rows / cols up to multiples: then it works for any size
    v = sa.VBR().fill_from_CSR_inplace(m, np.arange(30) // 4, 4, 4, True)
    bs, ind, val = v.to_blocked_ell()
    assert (v.rows, v.cols) == (32, 20) and ind.shape[0] == 8 and val.shape == (32, ind.shape[1] * 4)
