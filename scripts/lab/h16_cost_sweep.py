"""Developer script: 16-bit stream kernel, sweep of the plan's cost model (SPARTA_COST_MODEL=c2,c1,ct) and workers per CU."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa

m = sa.gen.cant_like(seed=2)
for (w, rb, N) in [(32, 32, 128), (64, 64, 128), (32, 32, 256)]:
    eng = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=rb, force_fixed_size=True, sim_measure=1)
    g = eng.GetGrouping(m)
    vb = sa.VBR().fill_from_CSR_inplace(m, g, w, rb, True)
    ldb = (vb.cols + 7) // 8 * 8
    B = ((torch.rand(ldb * N) - 0.5).to(torch.float16)).cuda()
    C = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
    for per_cu in ("2", "3"):
        for cm in ("12,10,6", "12,10,12", "12,10,20", "12,10,30", "12,10,45", "20,10,20", "16,10,20"):
            os.environ["SPARTA_COST_MODEL"] = cm
            os.environ["SPARTA_WORKERS_PER_CU"] = per_cu
            d = vb.to_device(0, dtype=sa.F16)
            ts = []
            for rep in range(3):
                for _ in range(20):
                    d.spmm(B, C, N, accumulate=False, ldb=ldb)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(300):
                    d.spmm(B, C, N, accumulate=False, ldb=ldb)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 300 * 1e3)
            i = d.info()
            print("w=%d N=%d workers/CU=%s cost=%s: us %s split %d" % (w, N, per_cu, cm, np.round(ts, 1), i["split_tiles"]), flush=True)
            del d
