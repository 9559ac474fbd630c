"""shared helpers for the tests: golden fixtures, seeded matrices, tolerance bound"""
import ast
import hashlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


_mats = None


def matrices():
    """the seeded matrices of tests/golden/make_golden.py (regenerated, then checked against the stored checksums)"""
    global _mats
    if _mats is None:
        import sparta_amd as sa
        from golden.make_golden import unsorted_rows
        _mats = {
            "unsorted": unsorted_rows(),
            "u256": sa.gen.uniform_random(256, 256, 2000, seed=11),
            "band1k": sa.gen.banded(1000, 12, 0.6, seed=12),
            "rmat2k": sa.gen.rmat(11, 30000, seed=13, pattern_only=True),
            "rect": sa.gen.uniform_random(300, 517, 6000, seed=14),
            "fem": sa.gen.fem3d(4, 4, 9, 3, seed=15),
            "c1": sa.gen.config1(),
        }
        cases = load("cases.npz")
        for k, m in _mats.items():
            want = str(cases["%s/csr_sha" % k])
            got = sha(m.rowptr) + sha(m.colidx) + (sha(m.vals) if m.vals is not None else "")
            assert got == want, "seeded generator drifted for %s: golden fixtures no longer match their inputs" % k
    return _mats


def case_list():
    cases = load("cases.npz")
    out = []
    for key, name, cfg in cases["index"]:
        out.append((str(key), str(name), dict(ast.literal_eval(str(cfg)))))
    return out


def case_fields(key):
    cases = load("cases.npz")
    pre = key + "/"
    return {k[len(pre):]: cases[k] for k in cases.files if k.startswith(pre)}


def abs_bound(rows, cols, w, row_part, nzcount, jab, mab, B, n):
    """sum_k |a||b| per element of C (oracle arithmetic): the scale of the fp32 tolerance 1e-5 * sum|a||b|"""
    from oracle import oracle as O
    return O.vbr_multiply(rows, cols, w, row_part, nzcount, jab, np.abs(mab), np.abs(B), n)
