import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SPARTA_DBG_TIMES"] = "1"
import torch
import sparta_amd as sa
from sparta_amd._lib import lib
m = sa.gen.cant_like()
N = 128
for fixed in (64, 32):
    g = np.arange(m.rows) // fixed
    vb = sa.VBR().fill_from_CSR_inplace(m, g, 64)
    d = vb.to_device(0)
    B = torch.from_numpy(sa.gen.dense_rhs(vb.cols, N, seed=3)).cuda()
    Cc = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
    for _ in range(5):
        d.spmm(B, Cc, N)
    torch.cuda.synchronize()
    P = d.info()["stream_workers"]
    t = np.zeros(8192 * 8, np.uint64)
    lib.sparta_dbg_read_times.argtypes = [C.c_void_p, C.c_int]
    assert lib.sparta_dbg_read_times(t.ctypes.data_as(C.c_void_p), P) == 0
    t = t[:4 * P].reshape(P, 4).astype(np.int64)
    steps = d.info()["stream_steps"] / P
    print("fixed", fixed, "steps/worker %.1f total cycles p50 %.0f (%.0f/step); epilogue+tail-of-step cycles p50 %.0f, epilogues p50 %.1f -> %.0f cycles each" % (
        steps, np.median(t[:, 2]), np.median(t[:, 2]) / steps, np.median(t[:, 0]), np.median(t[:, 1]), np.median(t[:, 0]) / max(np.median(t[:, 1]), 1)))
