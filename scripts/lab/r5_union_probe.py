"""lab (round 5): knock-out probes of the column-compacted tile kernel on the `clustered` family (timing only): SPARTA_UNION_PROBE bits 1 no B loads, 2 no A loads,
4 no MFMAs, 8 no tails, 16 no epilogue.   python scripts/lab/r5_union_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
import bench_suite as bs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
name, kind, make, kw, w = [c for c in bs.cases(sa) if c[0].startswith("clustered")][0]
m = make()
g = sa.BlockingEngine(col_block_size=w, **kw).GetGrouping(m)
for tail in os.environ.get("TAILS", "16,0").split(","):
    os.environ["SPARTA_UNION_TAIL"] = tail
    d = sa.DeviceVBS.from_csr(m, g, w, device=0)
    print("tail cap", tail, d.union_info(), d.sparse_info(), flush=True)
    B = torch.rand(d.cols * N, device="cuda") - 0.5
    C = torch.zeros(d.rows * N, device="cuda")
    Bp = d.prepare_b(B, N)
    for probe in (0, 8, 16, 1, 2, 3, 4, 7, 31):
        os.environ["SPARTA_UNION_PROBE"] = str(probe)
        for _ in range(20): d.spmm_prepared(Bp, C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): d.spmm_prepared(Bp, C)
        e1.record(); torch.cuda.synchronize()
        print("  probe %2d: %.1f us per product (prepared B: no transpose)" % (probe, e0.elapsed_time(e1) / 200 * 1e3), flush=True)
    os.environ["SPARTA_UNION_PROBE"] = "0"
    Bp.close(); d.close()
