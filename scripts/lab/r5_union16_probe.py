"""lab (round 5): what the 16-bit column-compacted tile kernel (vbs_union_h16_kernel) spends its time on -- the matrix of r5_union_l2.py (2000 true clusters x 48 rows, ~300 list columns
+ 6 of its own per row), bf16 handle, prepared B, N = 128 and 512; developer probes (timing only): 0 full, 8 no tails, 4 no MFMAs, 1 no loads of B, 2 no loads of A, 16 no epilogue.
python scripts/lab/r5_union16_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
import bench_suite as bs
m = bs._clustered(sa, 2000, 48, 60000, 300, 6, 5)
rng = np.random.Generator(np.random.PCG64(5)); order = rng.permutation(2000 * 48)
g = np.empty(m.rows, np.int64)
for gi in range(2000):
    rows = order[gi * 48:(gi + 1) * 48]; g[rows] = rows.min()
for dt, tdt, nm in ((sa.BF16, torch.bfloat16, "bf16"), (sa.F32, torch.float32, "f32")):
    d = sa.DeviceVBS.from_csr(m, g, 32, device=0, dtype=dt)
    for N in [int(x) for x in os.environ.get("NS", "128,512").split(",")]:
        ldb = (d.cols + 7) // 8 * 8
        B = (torch.rand(ldb * N, device="cuda") - 0.5).to(tdt)
        C = torch.zeros(d.rows * N, device="cuda")
        Bp = d.prepare_b(B, N, ldb=ldb)
        out = []
        for probe in (0, 8, 4, 1, 2, 16, 1 | 2 | 4 | 8 | 16):
            os.environ["SPARTA_UNION_PROBE"] = str(probe)
            for _ in range(10): d.spmm_prepared(Bp, C)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): d.spmm_prepared(Bp, C)
            e1.record(); torch.cuda.synchronize()
            out.append("probe %d: %.1f" % (probe, e0.elapsed_time(e1) / 100 * 1e3))
        os.environ["SPARTA_UNION_PROBE"] = "0"
        print("%s N = %d, us per product:  %s" % (nm, N, "   ".join(out)), flush=True)
        Bp.close()
    d.close()
