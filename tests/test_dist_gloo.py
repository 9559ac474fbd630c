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


# ---- sparsity-aware exchange (RowBlockExchange): plan + ONE all-to-all of the needed row-blocks ---------------------

def _cpu_hooks(ex_box, N, W_):
    """numpy-backed stand-ins for the two HIP calls of RowBlockExchange (pack kernel, SpMM), for CPU tensors over gloo"""
    import torch
    from oracle import oracle as O
    import sparta_amd as sa

    def pack(B_tiles):
        ex = ex_box[0]
        t = B_tiles.view(-1, W_ * N)
        ex.send_buf[:ex.n_send * W_ * N].view(-1, W_ * N).copy_(t[torch.from_numpy(ex.send_ids_host.astype(np.int64))])

    def product(which, B_tiles, C_out, accumulate):
        ex = ex_box[0]
        v = ex.own if which == "own" else ex.remote
        Bc = sa.dist.from_block_tiles(B_tiles.numpy()[:v.cols * N], v.cols, N, W_)
        C0 = C_out.numpy().copy() if accumulate else None
        Cn = O.vbr_multiply(v.rows, v.cols, W_, v.row_part, v.nzcount, v.jab, v.mab, Bc, N, C_in=C0)
        C_out.copy_(torch.from_numpy(np.asarray(Cn, np.float32).reshape(-1)))
    return pack, product


def _worker_blocks(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import sparta_amd as sa
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n_local, n_pad = sa.gen.fem3d_slab(3, 3, 6, rank, world, dof=3, pad_to=W, seed=9)
        g = sa.BlockingEngine(tau=0.4, col_block_size=W).GetGrouping(m)
        v = sa.VBR().fill_from_CSR_inplace(m, g, W)
        shard = sa.gen.dense_rhs(n_pad, N, seed=70 + rank)                       # column-major n_pad x N
        tiles = torch.from_numpy(sa.dist.to_block_tiles(shard, n_pad, N, W))
        box = [None]
        pack, product = _cpu_hooks(box, N, W)
        ex = sa.dist.RowBlockExchange(v, rank, world, n_pad, N, device=None, pack=pack, product=product)
        box[0] = ex
        C = torch.full((v.rows * N,), 7.0)                                        # overwritten (accumulate = False)
        ex.step(tiles, C)
        C2 = C.clone()
        ex.step(tiles, C2, accumulate=True)                                       # C2 = 2 * C
        np.savez(os.path.join(out_dir, "blk%d.npz" % rank), C=C.numpy(), C2=C2.numpy(), perm=sa.get_permutation(g), rows=v.rows, n_pad=n_pad,
                 n_send=ex.n_send, n_recv=ex.n_recv, frac=ex.needed_fraction, recv=ex.recv_buf.numpy()[:ex.n_recv * W * N],
                 need=np.concatenate([a + q * (n_pad // W) for q, a in enumerate(ex.need) if q != rank] + [np.zeros(0, np.int64)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_block_exchange_all_to_all(tmp_path, world):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_blocks, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import sparta_amd as sa
    from oracle import oracle as O
    res = [np.load(os.path.join(str(tmp_path), "blk%d.npz" % r)) for r in range(world)]
    n_pad = int(res[0]["n_pad"])
    shards = [sa.gen.dense_rhs(n_pad, N, seed=70 + r) for r in range(world)]
    Bfull = sa.dist.gathered_to_colmajor(np.concatenate(shards), world, n_pad, N).reshape(N, world * n_pad)
    for r in range(world):
        m, _, _ = sa.gen.fem3d_slab(3, 3, 6, r, world, dof=3, pad_to=W, seed=9)
        Cc = O.csr_multiply(m.rows, m.rowptr, m.colidx, m.vals, Bfull.reshape(-1), m.cols, N).reshape(N, m.rows)
        C = res[r]["C"].reshape(N, int(res[r]["rows"]))
        want = Cc[:, res[r]["perm"]]
        # two partial sums (own + received blocks) instead of one ascending sum: equal up to fp32 rounding of the re-association
        assert np.allclose(C, want, rtol=0, atol=1e-5 * np.abs(want).max()), "rank %d" % r
        assert np.allclose(res[r]["C2"].reshape(N, -1), 2 * C, rtol=0, atol=2e-5 * np.abs(want).max())
        # what arrived is exactly the needed row-blocks of the peers, in (source rank, block) order
        need = res[r]["need"]
        got = res[r]["recv"].reshape(len(need), N, W)
        for k, jb in enumerate(need):
            assert np.array_equal(got[k], Bfull[:, jb * W:(jb + 1) * W]), (r, jb)
        # a slab only needs the halo planes of its neighbours, far less than the all-gather moves
        assert 0 < res[r]["n_recv"] < (world - 1) * (n_pad // W)
        assert 0 < float(res[r]["frac"]) < 1
    # a middle rank of three talks to both neighbours; the end ranks to one
    if world == 3:
        assert int(res[1]["n_recv"]) > int(res[0]["n_recv"])


def test_split_own_remote_is_a_column_partition():
    import sparta_amd as sa
    from oracle import oracle as O
    world, rank = 3, 1
    m, n_local, n_pad = sa.gen.fem3d_slab(3, 3, 5, rank, world, dof=3, pad_to=W, seed=4)
    g = sa.BlockingEngine(tau=0.5, col_block_size=W).GetGrouping(m)
    v = sa.VBR().fill_from_CSR_inplace(m, g, W)
    own, rem, need = sa.dist.split_own_remote(v, rank, n_pad, world)
    assert own.cols == n_pad and rem.cols == sum(len(need[q]) for q in range(world) if q != rank) * W
    assert np.array_equal(own.nzcount + rem.nzcount, v.nzcount) and len(own.mab) + len(rem.mab) == len(v.mab)
    assert sorted(np.concatenate([own.mab, rem.mab]).tolist()) == sorted(v.mab.tolist())
    B = sa.gen.dense_rhs(v.cols, N, seed=5).reshape(N, v.cols)
    bps = n_pad // W
    ids = np.concatenate([need[q] + q * bps for q in range(world) if q != rank])
    B_own = np.ascontiguousarray(B[:, rank * n_pad:(rank + 1) * n_pad]).reshape(-1)
    B_rem = np.ascontiguousarray(np.concatenate([B[:, j * W:(j + 1) * W] for j in ids], axis=1)).reshape(-1)
    C = O.vbr_multiply(own.rows, own.cols, W, own.row_part, own.nzcount, own.jab, own.mab, B_own, N)
    C = O.vbr_multiply(rem.rows, rem.cols, W, rem.row_part, rem.nzcount, rem.jab, rem.mab, B_rem, N, C_in=C)
    want = O.vbr_multiply(v.rows, v.cols, W, v.row_part, v.nzcount, v.jab, v.mab, B.reshape(-1), N)
    assert np.allclose(C, want, rtol=0, atol=1e-5 * np.abs(want).max())
    # tile layout round trip
    t = sa.dist.to_block_tiles(B_own, n_pad, N, W)
    assert np.array_equal(sa.dist.from_block_tiles(t, n_pad, N, W), B_own)
    assert np.array_equal(t[:W], B_own[:W]) and np.array_equal(t[W:2 * W], B_own[n_pad:n_pad + W])
    with pytest.raises(ValueError):
        sa.dist.split_own_remote(v, rank, n_pad + W, world)


# ---- strong scaling of a power-law matrix (bench.py --workload rmat --gpus N; BASELINE configs[4] in miniature) --------------------
def _rmat_worker(rank, world, port, out_dir):
    """what bench.py does per rank: identical seeded generation + reorder, block-row ranges by cost, this rank's slab (rows of the
    range, columns padded to world * shard_rows), B row-sharded, ONE all-gather over gloo; the slab's product by the oracle"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import sparta_amd as sa
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = sa.gen.rmat(10, 9000, seed=3, symmetrize=True, pattern_only=False)
        g = sa.BlockingEngine(blocking_algo=7, tau=0.4, col_block_size=W).GetGrouping(m)
        perm, part = sa.get_permutation(g), sa.get_partition(g)
        cost = np.add.reduceat(np.diff(m.rowptr)[perm].astype(np.float64) + 1.0, part[:-1])
        b0, b1 = sa.dist.partition_by_cost(cost, world)[rank]
        my_rows = perm[part[b0]:part[b1]]
        shard_rows = sa.dist.padded_shard_rows(-(-m.cols // world), W)
        slab = sa.dist.row_slab(m, my_rows, world * shard_rows)
        gl = np.repeat(np.arange(b1 - b0, dtype=np.int64), np.diff(part[b0:b1 + 1]))
        # B: the global cols x N matrix is seeded; rank r holds rows [r * shard_rows, (r + 1) * shard_rows) (zero past cols)
        Bglob = np.zeros((N, world * shard_rows), np.float32)
        Bglob[:, :m.cols] = sa.gen.dense_rhs(m.cols, N, seed=77).reshape(N, m.cols)
        shard = torch.from_numpy(np.ascontiguousarray(Bglob[:, rank * shard_rows:(rank + 1) * shard_rows]).reshape(-1))
        gathered = torch.empty(world * shard_rows * N, dtype=torch.float32)
        sa.dist.allgather_B(shard, gathered)
        Bfull = sa.dist.gathered_to_colmajor(gathered.numpy(), world, shard_rows, N)
        v = sa.VBR().fill_from_CSR_inplace(slab, gl, W)
        C = O.vbr_multiply(v.rows, v.cols, W, v.row_part, v.nzcount, v.jab, v.mab, Bfull, N)
        np.savez(os.path.join(out_dir, "rmat_rank%d.npz" % rank), C=C, rows=my_rows[sa.get_permutation(gl)], b0=b0, b1=b1, cost=cost[b0:b1].sum())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_rmat_strong_scaling_partition_world2(tmp_path):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rmat_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    sys.path.insert(0, ROOT)
    import sparta_amd as sa
    from oracle import oracle as O
    m = sa.gen.rmat(10, 9000, seed=3, symmetrize=True, pattern_only=False)
    B = sa.gen.dense_rhs(m.cols, N, seed=77)
    truth = O.csr_multiply(m.rows, m.rowptr, m.colidx, m.vals, B, m.cols, N).reshape(N, m.rows)
    res = [np.load(os.path.join(str(tmp_path), "rmat_rank%d.npz" % r)) for r in range(WORLD)]
    seen = np.concatenate([r["rows"] for r in res])
    assert np.array_equal(np.sort(seen), np.arange(m.rows)), "the ranks' row ranges must tile the matrix"
    assert int(res[0]["b0"]) == 0 and int(res[0]["b1"]) == int(res[1]["b0"])
    for r in res:
        C = r["C"].reshape(N, len(r["rows"]))
        assert np.array_equal(C, truth[:, r["rows"]])            # same products, same (ascending-column) order: bit-identical
    costs = [float(r["cost"]) for r in res]
    assert max(costs) / (sum(costs) / WORLD) < 1.3


def test_partition_by_cost_and_row_slab():
    import sparta_amd as sa
    assert sa.dist.partition_by_cost([1, 1, 1, 1, 10, 1, 1], 3) == [(0, 4), (4, 5), (5, 7)]        # the heavy item gets a rank of its own
    assert sa.dist.partition_by_cost([5.0], 4) == [(0, 0), (0, 0), (0, 1), (1, 1)] or sum(b - a for a, b in sa.dist.partition_by_cost([5.0], 4)) == 1
    rng = np.random.Generator(np.random.PCG64(2))
    c = rng.pareto(1.5, size=4000) + 0.1                                                           # power-law costs
    for world in (2, 4, 8):
        parts = sa.dist.partition_by_cost(c, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(c) and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        loads = np.array([c[a:b].sum() for a, b in parts])
        assert loads.max() <= c.sum() / world + c.max()
    m = sa.gen.rmat(9, 3000, seed=4, symmetrize=True, pattern_only=False)
    rows = rng.permutation(m.rows)[:100]
    s = sa.dist.row_slab(m, rows, m.cols + 64)
    assert s.rows == 100 and s.cols == m.cols + 64
    for k, r in enumerate(rows):
        assert np.array_equal(s.colidx[s.rowptr[k]:s.rowptr[k + 1]], m.colidx[m.rowptr[r]:m.rowptr[r + 1]])
        assert np.array_equal(s.vals[s.rowptr[k]:s.rowptr[k + 1]], m.vals[m.rowptr[r]:m.rowptr[r + 1]])


def _peer_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import sparta_amd as sa
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_pad = 48
        shard = torch.from_numpy(sa.gen.dense_rhs(n_pad, N, seed=170 + rank))
        a = torch.empty(world * n_pad * N, dtype=torch.float32)
        b = torch.zeros_like(a)
        sa.dist.allgather_B(shard, a)
        sa.dist.allgather_B_peer_copies(shard, b, rank, world)      # world - 1 sends + receives posted together
        pick = sa.dist.pick_allgather(shard, torch.empty_like(a), rank, world, reps=2)      # collective; every rank gets the same answer
        np.savez(os.path.join(out_dir, "peer%d.npz" % rank), a=a.numpy(), b=b.numpy(), mode=pick["mode"], equal=pick["equal"],
                 ag=pick["all_gather_ms"], pc=pick["peer_copies_ms"])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_peer_copies_equal_the_all_gather_and_every_rank_picks_the_same(tmp_path, world):
    """the alternative to the collective all-gather (world - 1 point-to-point copies per rank, one per xGMI link on the
    GPUs) leaves the same gathered B, and the plan-time pick between the two is one decision for the whole job"""
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_peer_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import sparta_amd as sa
    res = [np.load(os.path.join(str(tmp_path), "peer%d.npz" % r)) for r in range(world)]
    want = np.concatenate([sa.gen.dense_rhs(48, N, seed=170 + r) for r in range(world)])
    for r in range(world):
        assert np.array_equal(res[r]["a"], want) and np.array_equal(res[r]["b"], want)
        assert bool(res[r]["equal"]) and str(res[r]["mode"]) == str(res[0]["mode"]) and str(res[r]["mode"]) in ("all_gather", "peer_copies")
        assert float(res[r]["ag"]) == float(res[0]["ag"]) and float(res[r]["pc"]) == float(res[0]["pc"])      # the maxima over the ranks
