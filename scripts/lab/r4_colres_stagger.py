"""Lab: start offsets between groups of CUs in the first dispatch round (SPARTA_COLRES_STAGGER_US x SPARTA_COLRES_GROUPS; 0 = all start together)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import sparta_amd as sa  # noqa: E402
import bench_suite as S  # noqa: E402
from r4_colres import timed  # noqa: E402

for f, n in (("bcsstk18_r.el", 8192), ("wiki-Vote_r.el", 8192), ("ca-HepPh_r.el", 8192), ("bcsstk18_r.el", 1024)):
    m = S._sorted_rows(sa, sa.CSR.read_from_edgelist(os.path.join(S.DATA, f), pattern_only=True))
    g = np.arange(m.rows, dtype=np.int64) // 64
    d = sa.DeviceVBS.from_csr(m, g, 64, 64, True, device=0)
    Bt = torch.from_numpy(sa.gen.dense_rhs(d.cols, n, seed=1)).cuda()
    Ct = torch.zeros(d.rows * n, dtype=torch.float32, device="cuda")
    for groups in (2, 3, 4):
        os.environ["SPARTA_COLRES_GROUPS"] = str(groups)
        row = []
        for us in ("0", "2", "4", "6", "8", "12", "16", "auto"):
            if us == "auto":
                os.environ.pop("SPARTA_COLRES_STAGGER_US", None)
            else:
                os.environ["SPARTA_COLRES_STAGGER_US"] = us
            ms = timed(d, Bt, Ct, n, 20 if n >= 8192 else 100)
            row.append("%s %.4f" % (us, ms))
        print(f, "N", n, "nc", d.colres_info()["nc"], "groups", groups, " ".join(row), flush=True)
    os.environ.pop("SPARTA_COLRES_GROUPS")
    os.environ.pop("SPARTA_COLRES_STAGGER_US", None)
    ms = timed(d, Bt, Ct, n, 20 if n >= 8192 else 100)
    print(f, "N", n, "library choice nc", d.colres_info()["nc"], "%.4f" % ms, flush=True)
    d.close()
