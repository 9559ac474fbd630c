"""lab (round 5): where a wave of vbs_union_f32_kernel spends its cycles -- libraries built with -DSPARTA_UNION_STATS=1 (scripts/lab/r5_union_ilv.sh names them) write per-wave
cycle sums over the head of C: waiting for its own loads / at the barrier / in the multiply phase (MFMAs with the next step's loads between them) / in tile epilogues.
SPARTA_AMD_LIB=.../libsparta_amd_st.so python scripts/lab/r5_union_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparta_amd as sa
import bench_suite as bs
N = 128
cols = int(os.environ.get("COLS", "60000"))
NCL = int(os.environ.get("NCL", "2000"))
os.environ.setdefault("SPARTA_SPARSE_MIN_STEPS", "0"); os.environ.setdefault("SPARTA_LAUNCH_NNZ", "0")
m = bs._clustered(sa, NCL, 48, cols, 300, 6, 5)
rng = np.random.Generator(np.random.PCG64(5)); order = rng.permutation(NCL * 48)
g = np.empty(m.rows, np.int64)
for gi in range(NCL):
    rows = order[gi * 48:(gi + 1) * 48]; g[rows] = rows.min()
d = sa.DeviceVBS.from_csr(m, g, 32, device=0)
B = torch.rand(d.cols * N, device="cuda") - 0.5
C = torch.zeros(d.rows * N, device="cuda")
Bp = d.prepare_b(B, N)
W = d.union_info()["workers"]
for probe in [int(x) for x in os.environ.get("PROBES", "0,8").split(",")]:
    os.environ["SPARTA_UNION_PROBE"] = str(probe)
    for _ in range(5): d.spmm_prepared(Bp, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); d.spmm_prepared(Bp, C); e1.record(); torch.cuda.synchronize()
    s = C[:W * 4 * 8].cpu().numpy().reshape(W, 4, 8).astype(np.float64)
    tot = s[:, :, 4]
    print("probe %d: %.1f us; per wave, mean over %d workers x 4 waves, cycles (share of the wave's total): wait %.0f (%.2f)  barrier %.0f (%.2f)  multiply %.0f (%.2f)  epilogue %.0f (%.2f)  total %.0f (max %.0f)  steps %.1f (max %d)"
          % (probe, e0.elapsed_time(e1) * 1e3, W, s[:, :, 0].mean(), (s[:, :, 0] / tot).mean(), s[:, :, 1].mean(), (s[:, :, 1] / tot).mean(), s[:, :, 2].mean(), (s[:, :, 2] / tot).mean(),
             s[:, :, 3].mean(), (s[:, :, 3] / tot).mean(), tot.mean(), tot.max(), s[:, :, 5].mean(), int(s[:, :, 5].max())), flush=True)
    per_step = s[:, :, :4].sum(axis=(0, 1)) / s[:, :, 5].sum()
    print("         per step and wave: wait %.0f  barrier %.0f  multiply %.0f  epilogue %.0f cycles" % tuple(per_step), flush=True)
