#!/usr/bin/env python3
"""hub plan on / off on the hub parts of the power-law configs: part 0 of configs[3] at 5 % and at 1 % (bf16, N = 512), part 0 of configs[4] (fp16, N = 256).
Prints the stream / fix-up / sparse kernel times, the plan, the executed PFLOP/s on the stored tile area."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, sparta_amd as sa
which = sys.argv[1] if len(sys.argv) > 1 else "5,1,c4"
only = which.endswith(",only")           # only the default plan (profiling runs)
if only: which = which[:-5]
variants = [("hub G=4", {})] if only else [("hub off", {"SPARTA_HUB": "0"}), ("hub G=4", {}), ("hub G=2", {"SPARTA_HUB_G": "2"})]
extra = [a for a in sys.argv[2:]]
for e in extra:                      # NAME=VALUE,NAME=VALUE label
    kv = dict(x.split("=") for x in e.split(","))
    variants.append((e, kv))
cfgs = {"5": (20, 5e-2, 64, 512, sa.BF16, torch.bfloat16), "1": (20, 1e-2, 16, 512, sa.BF16, torch.bfloat16), "c4": (23, 1e-4, 8, 256, sa.F16, torch.float16)}
for key in which.split(","):
    scale, dens, P, N, sdt, tdt = cfgs[key]
    E = sa.gen.rmat_raw_edges_for_density(scale, dens)
    part = int(os.environ.get("HUB_PART", "0"))
    r0, r1 = sa.gen.rmat_cuts(scale, E, P, n_cols=N)[part]          # the cuts bench_parts.py runs (cost constants per operand width)
    t0 = time.time()
    m = sa.gen.rmat_rows(scale, E, r0, r1, device=0)
    g = np.arange(m.rows) // 64
    n = 1 << scale
    ldb = n + (64 if os.environ.get("HUB_PAD_B", "1") != "0" else 0)        # columns not a power of two apart (sparta_vbs_spmm_gathered_ld's comment)
    B = torch.zeros(ldb * N, dtype=tdt, device="cuda")
    B.view(N, ldb)[:, :n] = sa.gen.dense_rhs_rows(0, n, N, dtype=tdt, device=0).view(N, n)
    C = torch.zeros(m.rows * N, dtype=torch.float32, device="cuda")
    print("config %s part %d: rows %d nnz %d (generate %.1f s)" % (key, part, m.rows, m.nztot(), time.time() - t0), flush=True)
    perm = sa.get_permutation(g)
    ref = None
    for name, env in variants:
        for k, v in env.items(): os.environ[k] = v
        t0 = time.time()
        d = sa.DeviceVBS.from_csr(m, g, 64, 64, False, device=0, dtype=sdt)
        tb = time.time() - t0
        info, hi = d.info(), d.hub_info()
        d.spmm(B, C, N, ldb=ldb); torch.cuda.synchronize()
        worst = 0.0
        for r in np.random.Generator(np.random.PCG64(1)).integers(0, m.rows, 12):
            i = perm[r]; cols_i = m.colidx[m.rowptr[i]:m.rowptr[i + 1]]
            a = torch.from_numpy(m.vals[m.rowptr[i]:m.rowptr[i + 1]]).to(tdt).double().numpy()
            bb = B.view(N, ldb)[:, torch.from_numpy(cols_i.astype(np.int64)).cuda()].double().cpu().numpy()
            got = C.view(N, -1)[:, int(r)].double().cpu().numpy()
            worst = max(worst, float((np.abs(got - bb @ a) / (np.abs(bb) @ np.abs(a) + 1e-30)).max()))
        d.set_class_timing(True)
        ts = []
        for _ in range(5):
            d.spmm(B, C, N, ldb=ldb); ts.append(d.class_times())
        st = np.mean([t["stream"] for t in ts]); fx = np.mean([t["fixup"] for t in ts]); sp = np.mean([t["sparse"] for t in ts])
        area = info["nztot"]
        print("  %-34s build %.1f s | stream %.3f ms fixup %.3f sparse %.3f | tile area %.4g -> %.0f TFLOP/s on the stored area | hub: %s | check %.1e" %
              (name, tb, st, fx, sp, area, 2.0 * area * N / (max(st, 1e-9) * 1e-3) / 1e12, hi, worst), flush=True)
        si = d.sparse_info()
        print("      sparse rows: %s -> %.1f Gnnz/s" % (si, si["nnz"] / max(sp, 1e-9) / 1e6), flush=True)
        d.close()
        for k in env: os.environ.pop(k, None)
    del B, C, m
    torch.cuda.empty_cache()
