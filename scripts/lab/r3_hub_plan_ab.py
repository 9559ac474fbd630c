#!/usr/bin/env python3
"""A/B of plan variants on the MFMA-carried hub of a power-law part (configs[3] at 1 %, part 0 of 16; bf16, N = 512): which tile order / cut lets the
B panels be shared through an XCD's L2.  Prints stream-kernel ms per variant."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sparta_amd as sa

scale, dens, P, N = 20, float(sys.argv[1]) if len(sys.argv) > 1 else 1e-2, int(sys.argv[2]) if len(sys.argv) > 2 else 16, 512
E = sa.gen.rmat_raw_edges_for_density(scale, dens)
r0, r1 = sa.gen.rmat_cuts(scale, E, P)[0]
m = sa.gen.rmat_rows(scale, E, r0, r1, device=0)
g = np.arange(m.rows) // 64
n = 1 << scale
B = sa.gen.dense_rhs_rows(0, n, N, dtype=torch.bfloat16, device=0)
C = torch.zeros(m.rows * N, dtype=torch.float32, device="cuda")
print("part 0: rows", m.rows, "nnz", m.nztot(), flush=True)
variants = [("default", {}), ("quad off", {"SPARTA_H16_QUAD": "0"}), ("interleave 1 (matrix order)", {"SPARTA_STREAM_INTERLEAVE": "1"}),
            ("interleave 0 (contiguous)", {"SPARTA_STREAM_INTERLEAVE": "0"}), ("align 1 (whole tiles)", {"SPARTA_STREAM_ALIGN": "1"}),
            ("align 1 + interleave 1", {"SPARTA_STREAM_ALIGN": "1", "SPARTA_STREAM_INTERLEAVE": "1"}), ("align 0 (split)", {"SPARTA_STREAM_ALIGN": "0"}),
            ("2 workgroups per CU", {"SPARTA_WORKERS_PER_CU": "2"})]
for name, env in variants:
    for k, v in env.items():
        os.environ[k] = v
    d = sa.DeviceVBS.from_csr(m, g, 64, 64, False, device=0, dtype=sa.BF16)
    info = d.info()
    d.spmm(B, C, N)
    torch.cuda.synchronize()
    d.set_class_timing(True)
    ts = []
    for _ in range(5):
        d.spmm(B, C, N)
        ts.append(d.class_times())
    st = np.mean([t["stream"] for t in ts]); fx = np.mean([t["fixup"] for t in ts]); sp = np.mean([t["sparse"] for t in ts])
    print("%-32s stream %.3f ms fixup %.3f sparse %.3f | tiles64 %d split %d steps %d area %.3g -> %.0f TFLOP/s executed" %
          (name, st, fx, sp, info["tiles64"], info["split_tiles"], info["stream_steps"], info["nztot"], 2.0 * info["nztot"] * N / (st * 1e-3) / 1e12), flush=True)
    d.close()
    for k in env:
        os.environ.pop(k, None)
