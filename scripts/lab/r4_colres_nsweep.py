import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, sparta_amd as sa, bench_suite as S
from r4_colres import timed
for f in ("bcsstk18_r.el", "wiki-Vote_r.el", "ca-HepPh_r.el"):
    m = S._sorted_rows(sa, sa.CSR.read_from_edgelist(os.path.join(S.DATA, f), pattern_only=True))
    g = np.arange(m.rows, dtype=np.int64) // 64
    d = sa.DeviceVBS.from_csr(m, g, 64, 64, True, device=0)
    for n in (128, 256, 512, 768, 1024, 2048, 4096, 8192):
        Bt = torch.rand(d.cols * n, device="cuda") - 0.5
        Ct = torch.zeros(d.rows * n, dtype=torch.float32, device="cuda")
        row = []
        for nc in (1, 2, 3, 4):
            os.environ["SPARTA_COLRES_NC"] = str(nc)
            ms = timed(d, Bt, Ct, n, 20 if n >= 4096 else 100)
            if d.colres_info()["nc"] == nc: row.append("nc%d %.4f" % (nc, ms))
        os.environ.pop("SPARTA_COLRES_NC")
        ms = timed(d, Bt, Ct, n, 20 if n >= 4096 else 100)
        print(f, "N", n, " ".join(row), "| choice nc %d %.4f" % (d.colres_info()["nc"], ms), flush=True)
