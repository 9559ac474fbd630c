#!/usr/bin/env python3
"""Golden vectors for the reference's own real-world inputs (data/minitest/*.el: bcsstk18 = SuiteSparse HB/bcsstk18, ca-HepPh, wiki-Vote,
social_location, ia-wikiquote-user-edits), produced by the COMPILED reference (oracle/_ref/libsparta_ref.so).  The .el files are DATA and
are committed under tests/golden/ref_data/minitest/ (nothing under /root/reference travels to the GPU box).

    python tests/golden/make_golden_real.py            -> tests/golden/real.npz

Per matrix (read by the reference's reader, pattern-only = its `-P 1`; bcsstk18 also with its values) and per blocking
  a3   : -a 3 -t 0.5 -b 64                 (IterativeBlockingPatternCLOCKED, the default algorithm)
  a5F1 : -a 5 -t 0.6 -b 32 -B 32 -F 1      (Keeper + fixed-size re-chunking: the flags of the multiplication experiments, batch/VBR_batch_a5:36)
the fixture holds: the CSR's checksums as the reference read it, grouping, counters, the VBS index arrays, SHA-256 of mab, and
C = VBR::multiply(B) for the seeded B (N = 8) as SHA-256 + its first 4096 values.
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402
import sparta_amd as sa  # noqa: E402  (seeded dense_rhs only)

N_COLS = 8
MATS = ["bcsstk18_r.el", "ca-HepPh_r.el", "wiki-Vote_r.el", "social_location.el", "ia-wikiquote-user-edits-nodup.el"]
BLOCKINGS = {"a3": dict(algo=3, tau=0.5, w=64, rbs=1, ff=False), "a5F1": dict(algo=5, tau=0.6, w=32, rbs=32, ff=True)}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def variants():
    for f in MATS:
        yield f, True
    yield "bcsstk18_r.el", False


def main():
    if not ref.available():
        raise SystemExit("oracle/_ref/libsparta_ref.so missing: run `make -C oracle ref` first")
    out, index = {}, []
    for f, pattern in variants():
        src = os.path.join("/root/reference/data/minitest", f)
        mine = os.path.join(HERE, "ref_data", "minitest", f)
        assert open(src, "rb").read() == open(mine, "rb").read(), "the committed copy of %s differs from the reference's file" % f
        c = ref.RefCSR.read(src, " ", pattern)
        rp, ci, v = c.export()
        key0 = "%s/%s" % (f, "pattern" if pattern else "values")
        out[key0 + "/dims"] = np.array([c.rows, c.cols, c.nnz], np.int64)
        out[key0 + "/csr_sha"] = np.array(sha(rp) + sha(ci.astype(np.int32)) + ("" if pattern else sha(v)))
        for bname, cfg in BLOCKINGS.items():
            t0 = time.time()
            g, st = c.grouping(algo=cfg["algo"], tau=cfg["tau"], col_block_size=cfg["w"], row_block_size=cfg["rbs"], force_fixed_size=cfg["ff"], with_info=True)
            vb = ref.RefVBR(c, g, cfg["w"], cfg["rbs"], cfg["ff"])
            row_part, nzcount, jab, mab = vb.export()
            B = sa.gen.dense_rhs(vb.cols, N_COLS, seed=77)
            C = vb.multiply(B, N_COLS)
            key = key0 + "/" + bname
            out[key + "/grouping"] = g.astype(np.int32)
            out[key + "/counters"] = np.array([st["comparison_counter"], st["merge_counter"]], np.int64)
            out[key + "/info"] = np.array([st["VBR_nzcount"], st["VBR_nzblocks_count"], st["VBR_longest_row"]], np.int64)
            out[key + "/dims"] = np.array([vb.rows, vb.cols, vb.block_rows, vb.block_cols, vb.nztot], np.int64)
            out[key + "/row_part"] = row_part.astype(np.int32)
            out[key + "/nzcount"] = nzcount.astype(np.int32)
            out[key + "/jab"] = jab.astype(np.int32)
            out[key + "/mab_sha"] = np.array(sha(mab))
            out[key + "/C_sha"] = np.array(sha(C.astype(np.float32)))
            out[key + "/C_head"] = C.astype(np.float32)[:4096]
            index.append((key, f, "pattern" if pattern else "values", bname))
            print(key, "rows", vb.rows, "block_rows", vb.block_rows, "blocks", len(jab), "nztot", vb.nztot, "%.1f s" % (time.time() - t0), flush=True)
    out["index"] = np.array(index)
    np.savez_compressed(os.path.join(HERE, "real.npz"), **out)
    print("real.npz", os.path.getsize(os.path.join(HERE, "real.npz")), "bytes")


if __name__ == "__main__":
    main()
