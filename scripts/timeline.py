"""Developer tool: per-step s_memtime timeline of one workgroup of the fp32 stream kernel (build: make -C sparta_amd/csrc timeline;
run with SPARTA_AMD_LIB=sparta_amd/libsparta_amd_tl.so).  usage: python scripts/timeline.py [long|flagship]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparta_amd as sa
from sparta_amd._lib import lib

what = sys.argv[1] if len(sys.argv) > 1 else "long"
N = 128
if what == "long":
    m = sa.gen.uniform_random(65536, 8192, int(65536 * 8192 * 0.01), seed=1)
    vb = sa.VBR().fill_from_CSR_inplace_fixed(m, 64, 64)
else:
    m = sa.gen.cant_like(seed=2)
    g = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=32, row_block_size=32, force_fixed_size=True).GetGrouping(m)
    vb = sa.VBR().fill_from_CSR_inplace(m, g, 32, 32, True)
os.environ["SPARTA_PATH"] = "stream"
h16 = len(sys.argv) > 2 and sys.argv[2] == "h16"          # python scripts/timeline.py flagship h16: the 16-bit stream kernel
d = vb.to_device(0, dtype=sa.F16 if h16 else sa.F32)
ldb = (vb.cols + 7) // 8 * 8 if h16 else vb.cols
if h16:
    B = ((torch.rand(ldb * N) - 0.5).to(torch.float16)).cuda()
else:
    B = torch.from_numpy(sa.gen.dense_rhs(vb.cols, N, seed=3)).cuda()
Cc = torch.zeros(vb.rows * N, dtype=torch.float32, device="cuda")
for _ in range(20):
    d.spmm(B, Cc, N, ldb=ldb)
d.set_class_timing(True)
for _ in range(3):
    d.spmm(B, Cc, N, ldb=ldb)
torch.cuda.synchronize()
print("kernel ms", d.class_times(), "MHz", d.clock_mhz())
out = np.zeros(4 * 64 * 8 + 2048, np.int64)
lib.sparta_debug_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
assert lib.sparta_debug_timeline(d.h, out.ctypes.data_as(C.POINTER(C.c_longlong))) == 0
t = out[:4 * 64 * 8].reshape(4, 64, 8)
wk = out[4 * 64 * 8:].reshape(1024, 2)
wk = wk[wk[:, 1] > 0]
if len(wk):
    t0 = wk[:, 0].min()
    st, en = (wk[:, 0] - t0) / 100.0, (wk[:, 1] - t0) / 100.0       # 100 MHz ticks -> us
    dur = en - st
    print('workers: %d; start spread %.1f us; duration mean %.1f min %.1f max %.1f us; last end %.1f us' % (len(wk), st.max(), dur.mean(), dur.min(), dur.max(), en.max()))
    ids = np.nonzero(out[4 * 64 * 8:].reshape(1024, 2)[:, 1] > 0)[0]
    for x in range(8):
        sel = ids % 8 == x
        print('   XCD %d: duration mean %.1f max %.1f, end mean %.1f max %.1f' % (x, dur[sel].mean(), dur[sel].max(), en[sel].mean(), en[sel].max()))
    order = np.argsort(dur)
    print('   slowest workers', ids[order[-6:]].tolist(), 'fastest', ids[order[:6]].tolist())
names = ["round0", "round1", "round2", "round3+loads", "epilogue", "barrier", "to next step"]
if h16:
    names = ["LDS write", "issue loads", "frag + MFMA", "-", "epilogue", "barrier", "to next step"]
for wv in range(4):
    tw = t[wv]
    ok = tw[:, 0] > 0
    if ok.sum() < 4:
        print("wave", wv, "no data"); continue
    tw = tw[ok]
    seg = np.diff(tw[:, :7], axis=1)                       # segments inside a step
    nxt = tw[1:, 0] - tw[:-1, 6]                           # barrier exit -> next step's first stamp
    step = tw[1:, 0] - tw[:-1, 0]
    print("wave %d: %d steps; step period mean %.0f (min %d max %d) s_memtime ticks" % (wv, len(tw), step.mean(), step.min(), step.max()))
    for k in range(6):
        print("    %-14s mean %7.0f  min %6d  max %6d" % (names[k], seg[:, k].mean(), seg[:, k].min(), seg[:, k].max()))
    print("    %-14s mean %7.0f  min %6d  max %6d" % (names[6], nxt.mean(), nxt.min(), nxt.max()))
print("first 6 steps of wave 0 (relative ticks):")
w0 = t[0][t[0][:, 0] > 0]
for r in w0[:6]:
    print("   ", (r[:7] - w0[0][0]).tolist(), "flags %x" % r[7])
# steps that end a tile (epilogue: 16-32 stores per wave) and the steps right after them, against the rest
STEP_LAST = 1 << 17
for wv in range(1):
    tw = t[wv][t[wv][:, 0] > 0]
    if len(tw) < 8:
        break
    last = (tw[:, 7] & STEP_LAST) != 0
    period = tw[1:, 0] - tw[:-1, 0]                        # period[k]: start of step k -> start of step k+1
    seg = np.diff(tw[:, :7], axis=1)
    after1 = np.roll(last, 1); after1[0] = False
    after2 = np.roll(last, 2); after2[:2] = False
    for name, sel in (("tile-end steps", last), ("1st step after a tile end", after1 & ~last), ("2nd step after", after2 & ~last & ~after1), ("other steps", ~last & ~after1 & ~after2)):
        s = sel[:-1]
        if s.sum() == 0:
            continue
        print("%-28s n=%3d period mean %6.0f | rounds %s | epilogue %5.0f barrier %5.0f" % (name, s.sum(), period[s].mean(), np.round(seg[:-1][s][:, :4].mean(axis=0)).astype(int).tolist(),
              seg[:-1][s][:, 4].mean(), seg[:-1][s][:, 5].mean()))
