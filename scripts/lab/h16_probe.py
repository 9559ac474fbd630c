"""Developer script: which stream bounds a step of the stream kernels?  SPARTA_DBG_PROBE bits: 1 = one hot B panel, 2 = hot A, 4 = one C tile.
Timing only (the products are wrong under a probe): the probes exist in the developer build only -
make -C sparta_amd/csrc timeline; SPARTA_AMD_LIB=sparta_amd/libsparta_amd_tl.so python scripts/h16_probe.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa

m = sa.gen.cant_like(seed=2)
w = rb = int(os.environ.get("WB", "32")); N = int(os.environ.get("NCOLS", "128"))
eng = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=rb, force_fixed_size=True, sim_measure=1)
g = eng.GetGrouping(m)
vb = sa.VBR().fill_from_CSR_inplace(m, g, w, rb, True)
for dt in ("f16", "f32"):
    for probe in range(8):
        os.environ["SPARTA_DBG_PROBE"] = str(probe)
        d = vb.to_device(0, dtype={"f16": sa.F16, "f32": sa.F32}[dt])
        if dt == "f16":
            ldb = (vb.cols + 7) // 8 * 8
            B = ((torch.rand(ldb * N) - 0.5).to(torch.float16)).cuda()
        else:
            ldb = vb.cols
            B = (torch.rand(ldb * N) - 0.5).cuda()
        C = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
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
        print("%s w=%d N=%d probe=%d (hotB=%d hotA=%d oneC=%d): us %s" % (dt, w, N, probe, probe & 1, (probe >> 1) & 1, (probe >> 2) & 1, np.round(ts, 1)), flush=True)
        del d
