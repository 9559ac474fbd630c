"""CPU, world_size 2 over gloo: the row-partitioned multi-GPU path minus the GPU kernel -- slab generation, ONE
all-gather of the B shards into the gathered layout, and the oracle's multiply of each rank's slab against it.  The
result must equal the same rows computed from the assembled global problem on one process."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD, W, N = 2, 16, 12


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import sparta_amd as sa
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n_local, n_pad = sa.gen.fem3d_slab(3, 3, 4, rank, world, dof=3, pad_to=W, seed=9)
        shard = torch.from_numpy(sa.gen.dense_rhs(n_pad, N, seed=70 + rank))
        gathered = torch.empty(world * n_pad * N, dtype=torch.float32)
        sa.dist.allgather_B(shard, gathered)                       # the one exchange step
        g = sa.BlockingEngine(tau=0.4, col_block_size=W).GetGrouping(m)
        v = sa.VBR().fill_from_CSR_inplace(m, g, W)
        Bfull = sa.dist.gathered_to_colmajor(gathered.numpy(), world, n_pad, N)
        C = O.vbr_multiply(v.rows, v.cols, W, v.row_part, v.nzcount, v.jab, v.mab, Bfull, N)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), C=C, perm=sa.get_permutation(g), gathered=gathered.numpy(),
                 n_pad=n_pad, rows=v.rows)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_row_partition_allgather(tmp_path):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)

    sys.path.insert(0, ROOT)
    import sparta_amd as sa
    from oracle import oracle as O
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(WORLD)]
    n_pad = int(res[0]["n_pad"])
    # every rank ended up with the same gathered B = [shard0 | shard1]
    want = np.concatenate([sa.gen.dense_rhs(n_pad, N, seed=70 + r) for r in range(WORLD)])
    for r in range(WORLD):
        assert np.array_equal(res[r]["gathered"], want)
    Bfull = sa.dist.gathered_to_colmajor(want, WORLD, n_pad, N)
    # single-process truth: CSR product of each slab (global padded column numbering) with the full B
    for r in range(WORLD):
        m, n_local, _ = sa.gen.fem3d_slab(3, 3, 4, r, WORLD, dof=3, pad_to=W, seed=9)
        Cc = O.csr_multiply(m.rows, m.rowptr, m.colidx, m.vals, Bfull, m.cols, N).reshape(N, m.rows)
        C = res[r]["C"].reshape(N, int(res[r]["rows"]))
        assert np.array_equal(C, Cc[:, res[r]["perm"]]), "rank %d" % r
    # the slabs really couple to the neighbour's shard (otherwise the all-gather would be untested)
    m0, _, _ = sa.gen.fem3d_slab(3, 3, 4, 0, WORLD, dof=3, pad_to=W, seed=9)
    assert (m0.colidx >= n_pad).any()


def test_partition_block_rows_balances_work():
    import sparta_amd as sa
    rng = np.random.Generator(np.random.PCG64(1))
    h = rng.integers(1, 200, size=500)
    row_part = np.concatenate([[0], np.cumsum(h)])
    nz = rng.integers(0, 60, size=500)
    for world in (1, 2, 4, 8):
        parts = sa.dist.partition_block_rows(row_part, nz, 64, world)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == 500
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        work = np.array([(h[a:b] * 64 * nz[a:b]).sum() for a, b in parts], float)
        assert work.max() <= 1.25 * work.sum() / world + (h * 64 * nz).max()
