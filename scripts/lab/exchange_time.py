"""Developer script: the two products of RowBlockExchange for a middle rank of the weak-scaling bench workload, timed on ONE GPU
(the all-to-all itself needs the other GPUs; here the receive buffer just holds random tiles)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
world, rank, w, N = 3, 1, 32, 128
m, n_local, S = sa.gen.fem3d_slab(9, 9, 257, rank, world, dof=3, pad_to=w, seed=2)
g = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=32, force_fixed_size=True).GetGrouping(m)
vb = sa.VBR().fill_from_CSR_inplace(m, g, w, 32, True)
need = []
for r in range(world):
    mr, _, _ = sa.gen.fem3d_slab(9, 9, 257, r, world, dof=3, pad_to=w, seed=2)
    gr = sa.BlockingEngine(blocking_algo=5, tau=0.6, col_block_size=w, row_block_size=32, force_fixed_size=True).GetGrouping(mr)
    need.append(sa.dist.needed_blocks(sa.VBR().fill_from_CSR_inplace(mr, gr, w, 32, True).jab, w, S, world))
ex = sa.dist.RowBlockExchange(vb, rank, world, S, N, device=0, all_need=need)
print("own blocks %d, remote blocks %d in %d block-rows; recv %d tiles (%.1f KB), send %d; needed fraction %.4f" % (
    len(ex.own.jab), len(ex.remote.jab), int((ex.remote.nzcount > 0).sum()), ex.n_recv, ex.n_recv * w * N * 4 / 1024, ex.n_send, ex.needed_fraction))
Bt = torch.rand(S * N, device="cuda") - 0.5
ex.recv_buf.copy_(torch.rand_like(ex.recv_buf) - 0.5)
C = torch.zeros(vb.rows * N, device="cuda")
def t(fn, reps=1000):
    for _ in range(50): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("pack        %.2f us" % t(lambda: ex._pack(Bt)))
print("own         %.2f us" % t(lambda: ex._product("own", Bt, C, False)))
print("remote      %.2f us" % t(lambda: ex._product("remote", ex.recv_buf, C, True)))
def both():
    ex._pack(Bt); ex._product("own", Bt, C, False); ex._product("remote", ex.recv_buf, C, True)
print("pack+own+remote %.2f us" % t(both))
print(ex.d_rem.info())

# a stand-in for the collective: a 256 KB device copy on a side stream, issued before the own product like the all-to-all is
side = torch.cuda.Stream()
src = torch.rand_like(ex.recv_buf)
def overlapped():
    ex._pack(Bt)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ex.recv_buf.copy_(src)
    ex._product("own", Bt, C, False)
    torch.cuda.current_stream().wait_stream(side)
    ex._product("remote", ex.recv_buf, C, True)
print("pack -> (copy on a side stream || own) -> remote %.2f us" % t(overlapped))
