"""Developer script: 16-bit stream kernel, register pipeline depth 2 vs 4 (SPARTA_H16_DEPTH), alternated inside one process."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa

m = sa.gen.cant_like(seed=2)
for (w, rb, N, dt) in [(32, 32, 128, "f16"), (64, 64, 128, "f16"), (32, 32, 256, "bf16"), (64, 64, 256, "f16")]:
    eng = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=rb, force_fixed_size=True, sim_measure=1)
    g = eng.GetGrouping(m)
    vb = sa.VBR().fill_from_CSR_inplace(m, g, w, rb, True)
    d = vb.to_device(0, dtype=sa.F16 if dt == "f16" else sa.BF16)
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    ldb = (vb.cols + 7) // 8 * 8
    B = ((torch.rand(ldb * N) - 0.5).to(tdt)).cuda()
    C = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
    res = {2: [], 4: []}
    ref = None
    for rep in range(6):
        for depth in (2, 4):
            os.environ["SPARTA_H16_DEPTH"] = str(depth)
            for _ in range(20):
                d.spmm(B, C, N, accumulate=False, ldb=ldb)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                d.spmm(B, C, N, accumulate=False, ldb=ldb)
            e1.record()
            torch.cuda.synchronize()
            res[depth].append(e0.elapsed_time(e1) / 300 * 1e3)
            if ref is None:
                ref = C.clone()
            else:
                assert torch.equal(ref, C), "depth 2 and depth 4 must give the same bits"
    print("w=%d rb=%d N=%d %s area=%d: depth2 us %s | depth4 us %s" % (w, rb, N, dt, vb.nztot, np.round(res[2], 1), np.round(res[4], 1)), flush=True)
