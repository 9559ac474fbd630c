"""Lab: the resident-column kernel (k_colres.hip) against the row gather on the reference's real matrices at N = 128 / 1024 / 8192, per columns-per-workgroup.
usage: python scripts/lab/r4_colres.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import sparta_amd as sa  # noqa: E402
import bench_suite as S  # noqa: E402

DATA = S.DATA
NS = [int(x) for x in sys.argv[1:]] or [128, 1024, 8192]


def timed(d, Bt, Ct, n, reps):
    for _ in range(3):
        d.spmm(Bt, Ct, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        d.spmm(Bt, Ct, n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if __name__ == "__main__":
  for f in ["bcsstk18_r.el", "ca-HepPh_r.el", "wiki-Vote_r.el", "ia-wikiquote-user-edits-nodup.el", "social_location.el"]:
      m = S._sorted_rows(sa, sa.CSR.read_from_edgelist(os.path.join(DATA, f), pattern_only=True))
      g = np.arange(m.rows, dtype=np.int64) // 64
      for n in NS:
          row = {}
          os.environ["SPARTA_COLRES"] = "0"
          d0 = sa.DeviceVBS.from_csr(m, g, 64, 64, True, device=0)
          Bt = torch.from_numpy(sa.gen.dense_rhs(d0.cols, n, seed=1)).cuda()
          Ct = torch.zeros(d0.rows * n, dtype=torch.float32, device="cuda")           # (-F 1 pads the rows to whole blocks)
          row["gather"] = timed(d0, Bt, Ct, n, 20 if n >= 8192 else 100)
          ref = Ct.clone()
          d0.close()
          os.environ["SPARTA_COLRES"] = "1"
          d = sa.DeviceVBS.from_csr(m, g, 64, 64, True, device=0)
          info = d.colres_info()
          if info["slices"]:
              for nc in (1, 2, 3, 4):
                  os.environ["SPARTA_COLRES_NC"] = str(nc)
                  ms = timed(d, Bt, Ct, n, 20 if n >= 8192 else 100)
                  if d.colres_info()["nc"] == nc:
                      row["nc%d" % nc] = ms
                      err = float((Ct - ref).abs().max())
                      row["err"] = max(row.get("err", 0.0), err)
              os.environ.pop("SPARTA_COLRES_NC")
          d.close()
          bound_us = (m.rows + m.cols) * n * 4 / 8e12 * 1e6
          print(f, "rows", m.rows, "nnz", m.nztot(), "N", n, "entries", info["entries"], "long", info["long_rows"], "lmax", info["lmax"],
                " ".join("%s %.4f" % (k, v) for k, v in row.items()), "bound_us %.1f" % bound_us, flush=True)
