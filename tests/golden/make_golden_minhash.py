"""REGRESSION fixture for blocking_algo 7 (an extension: there is no reference to take vectors from).  Pins the groupings the
algorithm returns today for seeded inputs, so that an unintended change of the hashing / candidate order shows up as a diff.
    python tests/golden/make_golden_minhash.py      -> tests/golden/minhash_regression.npz"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sparta_amd as sa

CASES = [("rmat", dict(scale=12, n_edges=40000, seed=3, symmetrize=True), 32, 0.4), ("rmat", dict(scale=12, n_edges=40000, seed=3, symmetrize=True), 64, 0.6),
         ("fem", dict(nx=5, ny=5, nz=30, dof=3, seed=2, pattern_only=True), 32, 0.6), ("uniform", dict(n_rows=3000, n_cols=3000, nnz=30000, seed=1, pattern_only=True), 16, 0.7)]


def matrix(kind, kw):
    return {"rmat": sa.gen.rmat, "fem": sa.gen.fem3d, "uniform": sa.gen.uniform_random}[kind](**kw)


def run(case):
    kind, kw, w, tau = case
    e = sa.BlockingEngine(blocking_algo=7, tau=tau, col_block_size=w)
    g = e.GetGrouping(matrix(kind, kw))
    return g, e.comparison_counter, e.merge_counter


if __name__ == "__main__":
    out = {}
    for i, c in enumerate(CASES):
        g, cmp_, mrg = run(c)
        out["sha%d" % i] = np.frombuffer(hashlib.sha256(g.tobytes()).digest(), np.uint8)
        out["stat%d" % i] = np.array([cmp_, mrg, len(np.unique(g))], np.int64)
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "minhash_regression.npz"), **out)
    print({k: (v.tolist() if k.startswith("stat") else bytes(v).hex()[:16]) for k, v in out.items()})
