"""debug: pair kernel determinism / equality between two handles of the same matrix (mfma-only rmat case of the tests)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPARTA_SPARSE_K"] = "0"
os.environ["SPARTA_PATH"] = "stream"
import torch
import sparta_amd as sa

n = 128
m, w = sa.gen.rmat(13, 60000, seed=5, symmetrize=True, pattern_only=True), 32
eng = sa.BlockingEngine(blocking_algo=7, tau=0.5, col_block_size=w, row_block_size=1, force_fixed_size=False)
g = eng.GetGrouping(m)
v = sa.VBR().fill_from_CSR_inplace(m, g, w, 0, False)
B = sa.gen.dense_rhs(v.cols, n, seed=31)
Bt = torch.from_numpy(B).cuda()
def run(d, reps=3):
    outs = []
    for _ in range(reps):
        C = torch.full((v.rows * n,), 0.25, dtype=torch.float32, device="cuda")
        d.spmm(Bt, C, n)
        torch.cuda.synchronize()
        outs.append(C.clone())
    return outs
d1 = v.to_device(0)
d2 = sa.DeviceVBS.from_csr(m, g, w, 0, False, device=0)
print("info1", d1.info()); print("info2", d2.info())
o1, o2 = run(d1), run(d2)
print("d1 run-to-run equal:", [bool(torch.equal(o1[0], x)) for x in o1[1:]])
print("d2 run-to-run equal:", [bool(torch.equal(o2[0], x)) for x in o2[1:]])
diff = (o1[0] - o2[0]).abs()
print("d1 vs d2: max abs diff %.3e, differing elements %d of %d" % (float(diff.max()), int((diff > 0).sum()), diff.numel()))
idx = torch.nonzero(diff.view(n, v.rows).amax(0) > 0).flatten().cpu().numpy()
print("differing C rows (first 40):", idx[:40], "count", len(idx))
rp = np.asarray(v.row_part)
br = np.searchsorted(rp, idx, side="right") - 1
print("their block-rows:", np.unique(br)[:40])
print("heights of those block-rows:", np.diff(rp)[np.unique(br)][:40], "nzcount", np.asarray(v.nzcount)[np.unique(br)][:40])
from oracle import oracle as O
Co = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, B, n)
for name, o in (("d1", o1[0]), ("d2", o2[0])):
    print(name, "vs oracle max abs err %.3e" % float(np.abs(o.cpu().numpy() - Co).max()))
print("---- neighbourhood of the differing rows")
h_all = np.diff(rp); nz_all = np.asarray(v.nzcount)
Cd1 = o1[0].view(n, v.rows).cpu().numpy(); Cref = Co.reshape(n, v.rows)
for r in idx[:8]:
    ib = int(np.searchsorted(rp, r, side="right") - 1)
    lo, hi = max(0, ib - 3), min(len(h_all), ib + 4)
    print("row", int(r), "block-row", ib, "| block-rows", lo, "..", hi - 1, "row_part", rp[lo:hi + 1].tolist(), "h", h_all[lo:hi].tolist(), "nb", nz_all[lo:hi].tolist())
    bad = np.nonzero(np.abs(Cd1[:, r] - Cref[:, r]) > 1e-4)[0]
    print("   bad columns of d1 vs oracle:", bad.tolist()[:64])
    if len(bad):
        j = bad[0]
        print("   col %d: got %.6f want %.6f ; neighbours' wanted values: %s" % (j, Cd1[j, r], Cref[j, r], [float(Cref[j, rr]) for rr in range(max(0, r - 3), min(v.rows, r + 4))]))
