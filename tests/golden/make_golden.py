#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REAL reference (oracle/_ref/libsparta_ref.so =
/root/reference/src/general/*.cpp compiled unmodified, see oracle/Makefile).  Runs only where /root/reference
exists (this container); the resulting .npz files are data (inputs + expected outputs) and are committed.

    python tests/golden/make_golden.py

Fixtures:
  kat9.npz      the reference's own 9x9 test matrix (data/TEST_matrix_weighted.el) as parsed by the reference's
                reader, with the README example configurations
  cases.npz     seeded matrices x {algo, w, tau, force_fixed} -> grouping, counters, permutation, VBS index arrays,
                SHA-256 of mab, blocking statistics and C = VBR::multiply(B) for a seeded B
  prims.npz     random cases of merge_rows / Hamming / Jaccard distances / get_permutation / get_fixed_size_grouping
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402
import sparta_amd as sa  # noqa: E402  (only its seeded generators are used here)

N_COLS = 6


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def unsorted_rows(seed=16):
    """a matrix whose rows are NOT ascending (like data/minitest/bcsstk18_r.el): exercises the literal reorder path"""
    m = sa.gen.uniform_random(200, 230, 2400, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    ci, v = m.colidx.copy(), m.vals.copy()
    for i in range(0, m.rows, 3):
        lo, hi = m.rowptr[i], m.rowptr[i + 1]
        p = rng.permutation(hi - lo)
        ci[lo:hi], v[lo:hi] = ci[lo:hi][p], v[lo:hi][p]
    return sa.CSR(m.rows, m.cols, m.rowptr, ci, v)


def matrices():
    return {
        "unsorted": unsorted_rows(),
        "u256": sa.gen.uniform_random(256, 256, 2000, seed=11),
        "band1k": sa.gen.banded(1000, 12, 0.6, seed=12),
        "rmat2k": sa.gen.rmat(11, 30000, seed=13, pattern_only=True),
        "rect": sa.gen.uniform_random(300, 517, 6000, seed=14),
        "fem": sa.gen.fem3d(4, 4, 9, 3, seed=15),
        "c1": sa.gen.config1(),
    }


def configs(name):
    if name == "c1":
        return [dict(algo=3, tau=0.5, w=64), dict(algo=2, tau=0.5, w=64, rbs=64), dict(algo=5, tau=0.5, w=64, rbs=64)]
    out = []
    for w in (16, 64):
        for tau in (0.2, 0.5):
            out.append(dict(algo=3, tau=tau, w=w))
        out.append(dict(algo=2, tau=0.5, w=w, rbs=24))
        out.append(dict(algo=5, tau=0.5, w=w, rbs=8))
        out.append(dict(algo=5, tau=0.2, w=w, rbs=32, ff=True))
    out.append(dict(algo=3, tau=0.4, w=7, ff=True, rbs=10))
    out.append(dict(algo=3, tau=0.5, w=16, use_groups=True))
    out.append(dict(algo=3, tau=0.5, w=16, use_pattern=False))
    out.append(dict(algo=0, tau=0.5, w=16))
    out.append(dict(algo=4, tau=0.5, w=16))
    out.append(dict(algo=3, tau=6.0, w=16, sim=0))
    out.append(dict(algo=6, tau=0.5, w=16))
    return out


def run_case(m, cfg):
    rc = ref.RefCSR(m.rows, m.cols, m.rowptr, m.colidx.astype(np.int64), m.vals)
    w, rbs, ff = cfg["w"], cfg.get("rbs", 1), cfg.get("ff", False)
    g, st = rc.grouping(algo=cfg["algo"], tau=cfg["tau"], col_block_size=w, row_block_size=rbs,
                        use_groups=cfg.get("use_groups", False), use_pattern=cfg.get("use_pattern", True),
                        force_fixed_size=ff, sim_measure=cfg.get("sim", 1), with_info=True)
    v = ref.RefVBR(rc, g, w, rbs, ff)
    row_part, nzcount, jab, mab = v.export()
    B = sa.gen.dense_rhs(v.cols, N_COLS, seed=77)
    C = v.multiply(B, N_COLS)
    return dict(grouping=g.astype(np.int32), perm=ref.get_permutation(g).astype(np.int32),
                counters=np.array([st["comparison_counter"], st["merge_counter"]], np.int64),
                info=np.array([st["VBR_nzcount"], st["VBR_nzblocks_count"], st["VBR_longest_row"]], np.int64),
                avg_height=np.float32(st["VBR_average_height"]),
                dims=np.array([v.rows, v.cols, v.block_rows, v.block_cols, v.nztot], np.int64),
                row_part=row_part.astype(np.int32), nzcount=nzcount.astype(np.int32), jab=jab.astype(np.int32),
                mab_sha=np.array(sha(mab)), C=C.astype(np.float32))


def main():
    if not ref.available():
        raise SystemExit("oracle/_ref/libsparta_ref.so missing: run `make -C oracle ref` first")
    # ---- the reference's own fixture -------------------------------------------------------------------------------
    c = ref.RefCSR.read("/root/reference/data/TEST_matrix_weighted.el")
    rp, ci, vals = c.export()
    kat = dict(rows=c.rows, cols=c.cols, rowptr=rp, colidx=ci, vals=vals)
    g, _ = c.grouping(algo=3, tau=0.6, col_block_size=3, row_block_size=3)
    kat["g_b3_t06"] = g
    v = ref.RefVBR(c, g, 3)
    a = v.export()
    kat.update(row_part=a[0], nzcount=a[1], jab=a[2], mab=a[3])
    kat["C_B1to18"] = v.multiply(np.arange(1, 19, dtype=np.float32), 2)
    kat["g_F1_B3"] = c.grouping(algo=3, tau=0.6, col_block_size=3, row_block_size=3, force_fixed_size=True)[0]
    kat["g_a5_B3_F1"] = c.grouping(algo=5, tau=0.6, col_block_size=3, row_block_size=3, force_fixed_size=True)[0]
    # TEST_matrices.cpp: fixed blocking -a 2 -b 3 -B 3, B = ones (5 columns): VBR::multiply == CSR::multiply
    gf, _ = c.grouping(algo=2, tau=0.6, col_block_size=3, row_block_size=3)
    vf = ref.RefVBR(c, gf, 3)
    ones = np.ones(9 * 5, np.float32)
    kat["C_fixed_ones_vbr"] = vf.multiply(ones, 5)
    kat["C_fixed_ones_csr"] = c.multiply(ones, 5)
    np.savez_compressed(os.path.join(HERE, "kat9.npz"), **kat)

    # ---- seeded cases ----------------------------------------------------------------------------------------------
    out = {}
    index = []
    for name, m in matrices().items():
        out["%s/csr_sha" % name] = np.array(sha(m.rowptr) + sha(m.colidx) + (sha(m.vals) if m.vals is not None else ""))
        for k, cfg in enumerate(configs(name)):
            key = "%s/%02d" % (name, k)
            res = run_case(m, cfg)
            for f, val in res.items():
                out["%s/%s" % (key, f)] = val
            index.append((key, name, repr(sorted(cfg.items()))))
            print(key, cfg, "block_rows", int(res["dims"][2]), "nztot", int(res["dims"][4]))
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(HERE, "cases.npz"), **out)

    # ---- primitives ------------------------------------------------------------------------------------------------
    rng = np.random.Generator(np.random.PCG64(2024))
    prim = {}
    rows = []
    for t in range(300):
        na, nb = int(rng.integers(0, 12)), int(rng.integers(0, 12))
        A = np.sort(rng.choice(40, size=na, replace=False)).astype(np.int64)
        Bv = np.sort(rng.choice(40, size=nb, replace=False)).astype(np.int64)
        ga, gb, bs = int(rng.integers(1, 5)), int(rng.integers(1, 3)), int(rng.integers(1, 9))
        mr = ref.merge_rows(A, Bv)
        rows.append((A, Bv, ga, gb, bs, mr, ref.distance(0, A, ga, Bv, gb, bs), ref.distance(1, A, ga, Bv, gb, bs)))
    prim["n"] = np.array(len(rows))
    for t, (A, Bv, ga, gb, bs, mr, dh, dj) in enumerate(rows):
        prim["%d/A" % t], prim["%d/B" % t], prim["%d/merged" % t] = A, Bv, mr
        prim["%d/par" % t] = np.array([ga, gb, bs], np.int64)
        prim["%d/dist" % t] = np.array([dh, dj], np.float32)
    # permutations with many ties (introsort tie order) at sizes around the 16-element threshold and beyond
    for t, n in enumerate((1, 2, 15, 16, 17, 33, 100, 1000, 5000)):
        g = rng.integers(0, max(2, n // 7), size=n).astype(np.int64)
        prim["perm%d/g" % t] = g
        prim["perm%d/perm" % t] = ref.get_permutation(g)
        prim["perm%d/part" % t] = ref.get_partition(g)
        prim["perm%d/fixed5" % t] = ref.get_fixed_size_grouping(g, 5)
    prim["nperm"] = np.array(9)
    np.savez_compressed(os.path.join(HERE, "prims.npz"), **prim)
    for f in ("kat9.npz", "cases.npz", "prims.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def main_mn():
    """mn.npz: IterativeBlockingPatternMN (blocking_algo 1) on three of the seeded matrices, m:n and flag variants"""
    mats = matrices()
    out = {}
    cfgs = [dict(tau=0.5, w=4, m=2, n=4), dict(tau=0.8, w=1, m=1, n=3), dict(tau=0.6, w=16, m=3, n=2, use_groups=True),
            dict(tau=0.5, w=8, m=2, n=4, use_pattern=False), dict(tau=0.7, w=8, m=1, n=1, ff=True, rbs=6), dict(tau=9.0, w=4, m=2, n=5, sim=0)]
    for name in ("u256", "rect", "unsorted"):
        m = mats[name]
        rc = ref.RefCSR(m.rows, m.cols, m.rowptr, m.colidx.astype(np.int64), m.vals)
        for k, cfg in enumerate(cfgs):
            g, st = rc.grouping(algo=1, tau=cfg["tau"], col_block_size=cfg["w"], row_block_size=cfg.get("rbs", 1),
                                use_groups=cfg.get("use_groups", False), use_pattern=cfg.get("use_pattern", True),
                                force_fixed_size=cfg.get("ff", False), sim_measure=cfg.get("sim", 1),
                                structured_m=cfg["m"], structured_n=cfg["n"])
            out["%s/%d/cfg" % (name, k)] = np.array(repr(cfg))
            out["%s/%d/grouping" % (name, k)] = g.astype(np.int32)
            out["%s/%d/counters" % (name, k)] = np.array([st["comparison_counter"], st["merge_counter"]], np.int64)
    np.savez_compressed(os.path.join(HERE, "mn.npz"), **out)
    print("wrote mn.npz with", len(out), "arrays")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mn":
    main_mn()
