"""Developer script: time the cant-like configurations (device-resident)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
m = sa.gen.cant_like()
N = int(os.environ.get("NCOLS", "128"))
cfgs = [(0.2, 0), (0.3, 0), (0.4, 0), (None, 32), (None, 48), (None, 64)]
for tau, fixed in cfgs:
    g = (np.arange(m.rows) // fixed) if fixed else sa.BlockingEngine(tau=tau, col_block_size=64).GetGrouping(m)
    vb = sa.VBR().fill_from_CSR_inplace(m, g, 64)
    d = vb.to_device(0)
    B = torch.from_numpy(sa.gen.dense_rhs(vb.cols, N, seed=3)).cuda()
    C = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
    for _ in range(5):
        d.spmm(B, C, N)
    torch.cuda.synchronize()
    d.set_class_timing(True)
    ts, cls = [], []
    for _ in range(30):
        ts.append(d.spmm(B, C, N, timed=True)); cls.append(list(d.class_times().values())[:3] + [0.0] * (3 - len(d.class_times())))
    t = float(np.median(ts)); c = np.median(np.array(cls), axis=0)
    i = d.info()
    print('tau=%s fixed=%s: %.1f us exec %.1f TF useful %.2f TF | kernels us %s | steps %d workers %d split %d path %d' % (tau, fixed, t * 1e3, 2 * vb.nztot * N / t / 1e9, 2 * m.nztot() * N / t / 1e9, np.round(c * 1e3, 1), i['stream_steps'], i['stream_workers'], i['split_tiles'], i['last_path']))
