#!/usr/bin/env python3
"""the 1 % hub part of configs[3] (bf16, N = 512) and part 0 of configs[4] (fp16, N = 256): stream / sparse kernel times under the current library"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, sparta_amd as sa
for scale, dens, P, N, sdt, tdt in ((20, 1e-2, 16, 512, sa.BF16, torch.bfloat16), (23, 1e-4, 8, 256, sa.F16, torch.float16)):
    E = sa.gen.rmat_raw_edges_for_density(scale, dens)
    r0, r1 = sa.gen.rmat_cuts(scale, E, P)[0]
    m = sa.gen.rmat_rows(scale, E, r0, r1, device=0)
    g = np.arange(m.rows) // 64
    B = sa.gen.dense_rhs_rows(0, 1 << scale, N, dtype=tdt, device=0)
    C = torch.zeros(m.rows * N, dtype=torch.float32, device="cuda")
    d = sa.DeviceVBS.from_csr(m, g, 64, 64, False, device=0, dtype=sdt)
    d.spmm(B, C, N); torch.cuda.synchronize()
    # spot check of 8 rows against float64
    perm = sa.get_permutation(g)
    worst = 0.0
    for r in np.random.Generator(np.random.PCG64(1)).integers(0, m.rows, 8):
        i = perm[r]; cols_i = m.colidx[m.rowptr[i]:m.rowptr[i + 1]]
        a = torch.from_numpy(m.vals[m.rowptr[i]:m.rowptr[i + 1]]).to(tdt).double().numpy()
        bb = B.view(N, -1)[:, torch.from_numpy(cols_i.astype(np.int64)).cuda()].double().cpu().numpy()
        got = C.view(N, -1)[:, int(r)].double().cpu().numpy()
        worst = max(worst, float((np.abs(got - bb @ a) / (np.abs(bb) @ np.abs(a) + 1e-30)).max()))
    d.set_class_timing(True)
    ts = []
    for _ in range(5):
        d.spmm(B, C, N); ts.append(d.class_times())
    print("scale %d: stream %.3f ms sparse %.3f ms, check %.2e" % (scale, np.mean([t["stream"] for t in ts]), np.mean([t["sparse"] for t in ts]), worst), flush=True)
    d.close(); del B, C, m
    torch.cuda.empty_cache()
