#!/usr/bin/env python3
"""one part of configs[4] (fp16, N = 256): sparse-row kernel time with the long rows' segments cut at column windows and processed window by window
(SPARTA_SP_WINDOW_COLS / SPARTA_SP_LONG / SPARTA_SP_MINSEG, read when the handle is built).  usage: r3_sparse_windows.py part 'window columns,LONG,MINSEG' ...   (empty = the library's default)"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, sparta_amd as sa
part = int(sys.argv[1])
scale, dens, P, N = int(os.environ.get("SCALE", 23)), float(os.environ.get("DENS", 1e-4)), int(os.environ.get("PARTS", 8)), int(os.environ.get("NCOLS", 256))
bf = os.environ.get("DT", "f16") == "bf16"
E = sa.gen.rmat_raw_edges_for_density(scale, dens)
r0, r1 = sa.gen.rmat_cuts(scale, E, P)[part]
m = sa.gen.rmat_rows(scale, E, r0, r1, device=0)
g = np.arange(m.rows) // 64
B = sa.gen.dense_rhs_rows(0, 1 << scale, N, dtype=torch.bfloat16 if bf else torch.float16, device=0)
C = torch.zeros(m.rows * N, dtype=torch.float32, device="cuda")
ref = None
for spec in sys.argv[2:]:
    W, LONG, MINSEG = (spec.split(",") + ["", ""])[:3]
    for k, v in (("SPARTA_SP_WINDOW_COLS", W), ("SPARTA_SP_LONG", LONG), ("SPARTA_SP_MINSEG", MINSEG)):
        if v: os.environ[k] = v
        else: os.environ.pop(k, None)
    t0 = time.time()
    d = sa.DeviceVBS.from_csr(m, g, 64, 64, False, device=0, dtype=sa.BF16 if bf else sa.F16)
    tb = time.time() - t0
    C.zero_()
    d.spmm(B, C, N); torch.cuda.synchronize()
    if ref is None: ref = C.clone(); err = 0.0
    else: err = float(((C - ref).abs().max() / ref.abs().max()).item())
    d.set_class_timing(True)
    ts = []
    for _ in range(4):
        d.spmm(B, C, N); ts.append(d.class_times())
    sp = d.sparse_info()
    print("scale %d part %d  window cols,LONG,MINSEG = %-14s  sparse %.3f ms  stream %.3f ms  (build %.1f s; hub rows %d, short rows %d; vs first %.1e)" % (
        scale, part, spec, np.mean([t["sparse"] for t in ts]), np.mean([t["stream"] for t in ts]), tb, sp["hub_rows"], sp["short_rows"], err), flush=True)
    d.close()
