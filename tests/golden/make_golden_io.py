#!/usr/bin/env python3
"""Generates tests/golden/io.npz from the REAL reference (oracle/_ref/libsparta_ref.so): what the reference's readers,
its edge-list writer, reorder_by_degree and save_blocking_data produce for a set of small text inputs.  Runs only where
/root/reference exists; the .npz holds inputs (file texts) and expected outputs and is committed.

    python tests/golden/make_golden_io.py

tests/golden/ref_data/ holds two data files of the reference's own tests, byte for byte (data/TEST_matrix_weighted.el,
data/TEST/TEST.g).
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402


def el_text(rng, rows, cols, nnz, delim=" ", weighted=True, comments=0, sort_cols=True, first_line="header"):
    rr = np.sort(rng.integers(0, rows, nnz))
    out = ["# comment %d\n" % k if k % 2 == 0 else "% other\n" for k in range(comments)]
    out.append(first_line + "\n")
    cur, cs = -1, []
    lines = []
    for r in rr:
        lines.append((int(r), int(rng.integers(0, cols))))
    if sort_cols:
        lines.sort()
    for r, c in lines:
        if weighted:
            out.append("%d%s%d%s%s\n" % (r, delim, c, delim, repr(float(np.float32(rng.integers(-40, 40) / 8.0)))))
        else:
            out.append("%d%s%d\n" % (r, delim, c))
    return "".join(out)


def texts():
    rng = np.random.Generator(np.random.PCG64(77))
    t = {}
    t["w_space"] = (el_text(rng, 30, 40, 160), " ", False, False)
    t["w_comma"] = (el_text(rng, 25, 25, 120, delim=","), ",", False, False)
    t["w_multi"] = (el_text(rng, 20, 33, 90, delim=" ; ", comments=3), " ; ", False, False)
    t["p_space"] = (el_text(rng, 40, 40, 200, weighted=False), " ", True, False)
    t["p_from_weighted"] = (el_text(rng, 30, 30, 100), " ", True, False)          # values present but ignored
    t["p_unsorted_cols"] = (el_text(rng, 30, 30, 150, weighted=False, sort_cols=False), " ", True, False)
    t["gaps"] = ("skipped\n3 1 2.5\n3 9 -1\n7 0 4\n", " ", False, False)           # empty rows 0-2, 4-6
    t["two_fields_weighted"] = ("x\n1 5\n2 7\n", " ", False, False)               # value = the column text parsed again
    t["one_field"] = ("x\n4\n6\n", " ", True, False)                              # (4,4), (6,6)
    t["tabs"] = ("h\n0\t3\t1.5\n2\t1\t-2\n", "\t", False, False)
    # symmetrize: upper-triangular pattern
    up = ["first\n"]
    for i in range(12):
        for j in sorted(set(int(x) for x in rng.integers(i, 12, 3))):
            up.append("%d %d\n" % (i, j))
    up.append("11 11\n")
    t["sym_upper"] = ("".join(up), " ", True, True)
    t["sym_not_triangular"] = ("f\n0 1\n1 0\n2 2\n", " ", True, True)             # not triangular: left alone
    return t


def mtx_texts():
    return {
        "with_extra_line": "%%MatrixMarket matrix coordinate pattern general\n% c\n4 5 3\nEXTRA LINE SKIPPED\n1 2\n3 5\n4 1\n",
        "values_ignored": "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 9.0\n2 3 1.5\n3 1 -2\n",   # first entry lost, reads 2
    }


def main():
    out = {}
    tmp = tempfile.mkdtemp()
    for name, (text, delim, pattern_only, symmetrize) in texts().items():
        p = os.path.join(tmp, name + ".el")
        open(p, "w").write(text)
        m = ref.RefCSR.read(p, delim, pattern_only, 0, symmetrize)
        rp, ci, v = m.export()
        out["el/%s/text" % name] = np.array(text)
        out["el/%s/args" % name] = np.array([delim, str(int(pattern_only)), str(int(symmetrize))])
        out["el/%s/dims" % name] = np.array([m.rows, m.cols], np.int64)
        out["el/%s/rowptr" % name] = rp
        out["el/%s/colidx" % name] = ci
        if not pattern_only:
            out["el/%s/vals" % name] = v
        w = os.path.join(tmp, name + ".out")
        m.save_to_edgelist(w, delim, pattern_only, 0)
        out["el/%s/saved" % name] = np.array(open(w).read())
    for name, text in mtx_texts().items():
        p = os.path.join(tmp, name + ".mtx")
        open(p, "w").write(text)
        m = ref.RefCSR.read(p, " ", True, 1, False)
        rp, ci, _ = m.export()
        out["mtx/%s/text" % name] = np.array(text)
        out["mtx/%s/dims" % name] = np.array([m.rows, m.cols], np.int64)
        out["mtx/%s/rowptr" % name] = rp
        out["mtx/%s/colidx" % name] = ci
    # the reference's own fixture, weighted and pattern-only
    fx = os.path.join(HERE, "ref_data", "TEST_matrix_weighted.el")
    for po in (0, 1):
        m = ref.RefCSR.read(fx, " ", bool(po), 0, False)
        rp, ci, v = m.export()
        out["fixture/po%d/dims" % po] = np.array([m.rows, m.cols], np.int64)
        out["fixture/po%d/rowptr" % po] = rp
        out["fixture/po%d/colidx" % po] = ci
        out["fixture/po%d/vals" % po] = v
    # reorder_by_degree: distinct degrees (both directions), ties ascending, ties descending with <= 16 rows
    rng = np.random.Generator(np.random.PCG64(5))
    cases = {"distinct40": rng.permutation(40), "ties200": rng.integers(0, 6, 200), "ties16": rng.integers(0, 3, 16),
             "ties9": rng.integers(0, 2, 9)}
    for name, deg in cases.items():
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        colidx = np.concatenate([np.arange(d) for d in deg] + [np.zeros(0, int)]).astype(np.int64)
        tag = np.repeat(np.arange(len(deg)), deg).astype(np.float32)       # value = original row id
        for desc in (0, 1):
            if desc and name == "ties200":
                continue                                                    # undefined in the reference
            m = ref.RefCSR(len(deg), int(max(deg.max(), 1)), rowptr, colidx, tag)
            m.reorder_by_degree(bool(desc))
            rp, ci, v = m.export()
            perm = np.full(len(deg), -1, np.int64)
            for k in range(len(deg)):
                if rp[k + 1] > rp[k]:
                    perm[k] = int(v[rp[k]])
            out["deg/%s/desc%d/degrees" % (name, desc)] = deg.astype(np.int64)
            out["deg/%s/desc%d/new_rowptr" % (name, desc)] = rp
            out["deg/%s/desc%d/perm_where_nonempty" % (name, desc)] = perm
    # save_blocking_data on the fixture (README example flags) and on a generated matrix
    m = ref.RefCSR.read(fx, " ", False, 0, False)
    rows = []
    for k, kw in enumerate([dict(tau=0.6, col_block_size=3, row_block_size=3, blocking_algo=3),
                            dict(tau=0.5, col_block_size=2, row_block_size=4, blocking_algo=5, force_fixed_size=1, b_cols=64, n_streams=8,
                                 exp_repetitions=10, multiplication_algo=6, sim_use_groups=1, symmetrize=1, reorder=-1),
                            dict(tau=0.25, col_block_size=4, row_block_size=2, blocking_algo=2, sim_measure=0, use_pattern=0)]):
        csv, g = m.save_blocking_data(filename="data/TEST_matrix_weighted.el", exp_name="exp%d" % k, timers=[1234.5, 77.25, 901.0],
                                      mult=[0.125 * (k + 1), 0.001], **kw)
        grp, st = m.grouping(algo=kw["blocking_algo"], tau=kw["tau"], col_block_size=kw["col_block_size"],
                             row_block_size=kw["row_block_size"], use_groups=bool(kw.get("sim_use_groups", 0)),
                             use_pattern=bool(kw.get("use_pattern", 1)), force_fixed_size=bool(kw.get("force_fixed_size", 0)),
                             sim_measure=kw.get("sim_measure", 1), with_info=True)
        out["csv/%d/kwargs" % k] = np.array(repr(kw))
        out["csv/%d/csv" % k] = np.array(csv)
        out["csv/%d/gfile" % k] = np.array(g)
        out["csv/%d/grouping" % k] = grp
        out["csv/%d/stats" % k] = np.array(repr(st))
    np.savez_compressed(os.path.join(HERE, "io.npz"), **out)
    print("wrote io.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
