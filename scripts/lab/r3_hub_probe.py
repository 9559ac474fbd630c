#!/usr/bin/env python3
"""which operand bounds the four-accumulator hub kernel: the 1 % hub part of configs[3] with developer builds whose steps read a cache-hot A slice / B panel
(SPARTA_H16_PROBE 128 / 256 / 384: timing only, results wrong).  One process per library (SPARTA_AMD_LIB is read at import)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import torch, sparta_amd as sa
scale, dens, P, N = 20, 1e-2, 16, 512
E = sa.gen.rmat_raw_edges_for_density(scale, dens)
r0, r1 = sa.gen.rmat_cuts(scale, E, P)[0]
m = sa.gen.rmat_rows(scale, E, r0, r1, device=0)
g = np.arange(m.rows) // 64
B = sa.gen.dense_rhs_rows(0, 1 << scale, N, dtype=torch.bfloat16, device=0)
C = torch.zeros(m.rows * N, dtype=torch.float32, device="cuda")
d = sa.DeviceVBS.from_csr(m, g, 64, 64, False, device=0, dtype=sa.BF16)
d.spmm(B, C, N); torch.cuda.synchronize()
d.set_class_timing(True)
ts = []
for _ in range(5):
    d.spmm(B, C, N); ts.append(d.class_times()["stream"])
print("%%-28s stream %%.3f ms" %% (os.environ.get("SPARTA_AMD_LIB", "default").split("/")[-1], float(np.mean(ts))), flush=True)
''' % ROOT
for lib in ("", "libsparta_amd_p128.so", "libsparta_amd_p256.so", "libsparta_amd_p384.so"):
    env = dict(os.environ)
    if lib:
        env["SPARTA_AMD_LIB"] = os.path.join(ROOT, "sparta_amd", lib)
    subprocess.run([sys.executable, "-c", CHILD], env=env)
