"""Developer check: does replaying the SpMM step from a HIP graph shorten the step?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import sparta_amd as sa
m = sa.gen.cant_like(seed=2)
eng = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=32, row_block_size=32, force_fixed_size=True)
g = eng.GetGrouping(m)
vb = sa.VBR().fill_from_CSR_inplace(m, g, 32, 32, True)
d = vb.to_device(0)
N = 128
B = torch.rand(vb.cols * N, device="cuda") - 0.5
C = torch.zeros(vb.rows * N, device="cuda")
for _ in range(20):
    d.spmm(B, C, N)
torch.cuda.synchronize()
def timeit(fn, k=300):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e6
print("eager  us/step", timeit(lambda: d.spmm(B, C, N)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    d.spmm(B, C, N)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, stream=s):
        d.spmm(B, C, N)
torch.cuda.synchronize()
print("graph1 us/step", timeit(gph.replay))
g10 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g10, stream=s):
        for _ in range(10): d.spmm(B, C, N)
torch.cuda.synchronize()
print("graph10 us/step", timeit(g10.replay, 30) / 10)
# dependency check: C1 = A.B ; C2 = A.C1 (C1 read as a B with ld = rows), eager vs one graph holding both launches
ldc = vb.rows
C1e = torch.zeros(vb.rows * N, device="cuda"); C2e = torch.zeros(vb.rows * N, device="cuda")
d.spmm(B, C1e, N); d.spmm(C1e, C2e, N, ldb=ldc)
torch.cuda.synchronize()
C1g = torch.zeros(vb.rows * N, device="cuda"); C2g = torch.zeros(vb.rows * N, device="cuda")
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    d.spmm(C1g, C2g, N, ldb=ldc)        # warm the (ldb) autotune key outside the capture
    torch.cuda.synchronize()
    with torch.cuda.graph(g2, stream=s):
        for _ in range(5):
            d.spmm(B, C1g, N); d.spmm(C1g, C2g, N, ldb=ldc)
torch.cuda.synchronize()
C1g.zero_(); C2g.zero_()
for _ in range(3): g2.replay()
torch.cuda.synchronize()
print("dependent steps in a graph identical to eager:", bool(torch.equal(C1g, C1e) and torch.equal(C2g, C2e)))
print("graph(5 x 2 dependent) us/step", timeit(g2.replay, 30) / 10)
