"""Lab: where the resident-column kernel's time goes -- the same product with phases switched off (SPARTA_COLRES_PROBE: 1 no loads of B, 2 no stream of A, 4 no stores of C)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import sparta_amd as sa  # noqa: E402
import bench_suite as S  # noqa: E402
from r4_colres import timed  # noqa: E402,F401

DATA = S.DATA
for f, n in (("bcsstk18_r.el", 8192), ("wiki-Vote_r.el", 8192), ("bcsstk18_r.el", 128)):
    m = S._sorted_rows(sa, sa.CSR.read_from_edgelist(os.path.join(DATA, f), pattern_only=True))
    g = np.arange(m.rows, dtype=np.int64) // 64
    d = sa.DeviceVBS.from_csr(m, g, 64, 64, True, device=0)
    Bt = torch.from_numpy(sa.gen.dense_rhs(d.cols, n, seed=1)).cuda()
    Ct = torch.zeros(d.rows * n, dtype=torch.float32, device="cuda")
    for nc in (1, 2, 3, 4):
        os.environ["SPARTA_COLRES_NC"] = str(nc)
        row = []
        for probe in (0, 1, 2, 4, 3, 5, 6, 7):
            os.environ["SPARTA_COLRES_PROBE"] = str(probe)
            ms = timed(d, Bt, Ct, n, 20 if n >= 8192 else 100)
            if d.colres_info()["nc"] != nc:
                break
            row.append("probe%d %.4f" % (probe, ms))
        if row:
            print(f, "N", n, "nc", nc, " ".join(row), flush=True)
    os.environ.pop("SPARTA_COLRES_NC"); os.environ.pop("SPARTA_COLRES_PROBE")
    d.close()
