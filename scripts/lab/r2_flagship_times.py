"""Developer script: flagship (cant-like, Keeper 32, w 32, N 128 fp32) -- ms per product and the per-launch-class times of the current env."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
m = sa.gen.cant_like(seed=2)
eng = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=32, row_block_size=32, force_fixed_size=True, sim_measure=1)
v = sa.VBR().fill_from_CSR_inplace(m, eng.GetGrouping(m), 32, 32, True)
d = v.to_device(0)
N = 128
B = torch.rand(v.cols * N, device="cuda") - 0.5
C = torch.zeros(v.rows * N, device="cuda")
for _ in range(200): d.spmm(B, C, N)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(500): d.spmm(B, C, N)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 500
d.set_class_timing(True)
ts = []
for _ in range(20):
    d.spmm(B, C, N); ts.append(d.class_times())
d.set_class_timing(False)
ct = {k: float(np.median([t[k] for t in ts])) for k in ts[0]}
i = d.info()
print("%-24s ms %.5f  class %s  aligned %s split %s" % (sys.argv[1] if len(sys.argv) > 1 else "", ms, {k: round(x, 5) for k, x in ct.items()}, i.get("plan_aligned"), i.get("n_split")))
