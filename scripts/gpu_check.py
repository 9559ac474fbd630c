"""Developer script: first-contact correctness + timing of the HIP path on a GPU box (uses oracle/_ref as checker)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparta_amd as sa
from oracle import ref

def ref_c(m, g, w, B, N, rbs=0, ff=False):
    rc = ref.RefCSR(m.rows, m.cols, m.rowptr, m.colidx.astype(np.int64), m.vals)
    vr = ref.RefVBR(rc, g, w, rbs, ff)
    return vr.multiply(B, N)

def check(name, m, g, w, N, rbs=0, ff=False):
    vb = sa.VBR().fill_from_CSR_inplace(m, g, w, rbs, ff)
    B = sa.gen.dense_rhs(vb.cols, N, seed=11)
    Cr = ref_c(m, g, w, B, N, rbs, ff)
    d = vb.to_device(0)
    print(name, d.info())
    out = {}
    for algo, an in ((sa.SPMM_MFMA, 'mfma'), (sa.SPMM_EXACT, 'exact')):
        for vec in ('0', '1'):
            os.environ['SPARTA_NO_VEC'] = vec
            C = np.zeros(vb.rows * N, np.float32)
            dt = d.spmm_host(B, N, C, accumulate=True, algo=algo)
            err = np.abs(C - Cr).max()
            scale = np.abs(Cr).max()
            bit = bool(np.array_equal(C, Cr))
            print('  %-5s novec=%s dt=%.3f ms maxerr=%.3e (scale %.3e) bitexact=%s' % (an, vec, dt, err, scale, bit))
            out[an + vec] = dict(dt=dt, err=float(err), bit=bit)
    os.environ['SPARTA_NO_VEC'] = '0'
    # overwrite mode + accumulate onto nonzero C
    C0 = sa.gen.dense_rhs(vb.rows, N, seed=5)
    C = C0.copy(); d.spmm_host(B, N, C, accumulate=True)
    C2 = np.full(vb.rows * N, 7.0, np.float32); d.spmm_host(B, N, C2, accumulate=False)
    print('  acc-onto-C err %.3e ; overwrite err %.3e' % (np.abs(C - (C0 + Cr)).max(), np.abs(C2 - Cr).max()))
    return out

res = {}
c = sa.gen.uniform_random(9, 9, 14, seed=1)
res['tiny'] = check('tiny9 w3', c, sa.BlockingEngine(tau=0.6, col_block_size=3).GetGrouping(c), 3, 2)
m = sa.gen.config1()
g = sa.BlockingEngine(tau=0.5, col_block_size=64).GetGrouping(m)
res['c1'] = check('C1 tau.5 w64 N64', m, g, 64, 64)
g = np.arange(m.rows) // 64
res['c1f'] = check('C1 fixed64 N128', m, g, 64, 128)
m2 = sa.gen.uniform_random(1000, 777, 30000, seed=9)
res['rect'] = check('rect 1000x777 w48 N100', m2, sa.BlockingEngine(tau=0.7, col_block_size=48).GetGrouping(m2), 48, 100)
res['rect2'] = check('rect 1000x777 w100 fixed 200 N33', m2, np.arange(1000) // 200, 100, 33)

# timing on the cant-like workload, device-resident
m = sa.gen.cant_like()
N = 128
for tau, fixed in ((0.2, 0), (0.4, 0), (None, 32), (None, 64), (None, 128)):
    if fixed:
        g = np.arange(m.rows) // fixed
    else:
        g = sa.BlockingEngine(tau=tau, col_block_size=64).GetGrouping(m)
    vb = sa.VBR().fill_from_CSR_inplace(m, g, 64)
    d = vb.to_device(0)
    B = torch.from_numpy(sa.gen.dense_rhs(vb.cols, N, seed=3)).cuda()
    C = torch.zeros(vb.rows * N, dtype=torch.float32, device='cuda')
    for _ in range(3):
        d.spmm(B, C, N)
    torch.cuda.synchronize()
    ts = [d.spmm(B, C, N, timed=True) for _ in range(20)]
    t = float(np.median(ts))
    info = d.info()
    print('cant-like tau=%s fixed=%s: %.1f us  exec %.1f TF (padded %.1f TF)  useful %.2f TF  tiles %s' % (
        tau, fixed, t * 1e3, 2 * vb.nztot * N / t / 1e9, 2 * info['exec_area'] * N / t / 1e9, 2 * m.nztot() * N / t / 1e9,
        [info[k] for k in ('tiles16', 'tiles32', 'tiles64', 'sparse_rows')]))
    res['cant_%s_%s' % (tau, fixed)] = dict(ms=t, area=vb.nztot)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/gpu_check.json', 'w'), indent=1)
