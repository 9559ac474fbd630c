"""Developer script: the banded 200k matrix of the suite (short block-rows: 2 blocks per 28-row tile), ms per product under the current env."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
m = sa.gen.banded(200000, 16, density=0.5, seed=4)
eng = sa.BlockingEngine(col_block_size=32, blocking_algo=7, tau=0.5, minhash_max_rows=32, row_block_size=32, force_fixed_size=True)
g = eng.GetGrouping(m)
d = sa.DeviceVBS.from_csr(m, g, 32, 32, True, device=0)
N = 128
B = torch.rand(d.cols * N, device="cuda") - 0.5
C = torch.zeros(d.rows * N, device="cuda")
import time
t_pre = time.time()
while time.time() - t_pre < 0.3:
    for _ in range(20): d.spmm(B, C, N)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300): d.spmm(B, C, N)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 300
d.set_class_timing(True)
ts = []
for _ in range(20):
    d.spmm(B, C, N); ts.append(d.class_times())
d.set_class_timing(False)
ct = {k: float(np.median([t[k] for t in ts])) for k in ts[0]}
i = d.info()
print("%-20s ms %.5f class %s tiles %s/%s/%s steps %s split %s sparse_rows %s" % (sys.argv[1] if len(sys.argv) > 1 else "", ms, {k: round(x, 5) for k, x in ct.items()},
      i["tiles16"], i["tiles32"], i["tiles64"], i["stream_steps"], i["split_tiles"], i["sparse_rows"]))
