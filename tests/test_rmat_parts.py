"""CPU: the canonical R-MAT graph behind `bench.py --workload rmat-part` (BASELINE configs[4], and configs[3] slab-streamed) --
row model, cuts by expected cost, piece-wise generation (any row range without the rest), the canonical dense operand, the two-rank
flow over gloo (each rank generates only its part; one all-gather of B; product by the oracle), and bench.py starting its own ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALE, DENSITY, N, W = 13, 4e-3, 8, 16


def _graph():
    import sparta_amd as sa
    E = sa.gen.rmat_raw_edges_for_density(SCALE, DENSITY)
    return sa, E


def test_row_model_predicts_the_sampled_degrees_and_the_density():
    sa, E = _graph()
    m, st = sa.gen.rmat_rows(SCALE, E, 0, 1 << SCALE, device="cpu", return_stats=True)
    want = DENSITY * 4.0 ** SCALE
    assert abs(m.nztot() - want) / want < 0.01                       # the raw-edge count is calibrated for the stated density
    assert abs(st["raw_edges_drawn"] - E) <= (1 << sa.gen.rmat_piece_bits(SCALE))       # every piece rounds its expected share
    distinct, share = sa.gen.rmat_row_model(SCALE, E)
    pc = sa.gen._popcount_np(np.arange(1 << SCALE, dtype=np.uint64))
    deg = np.diff(m.rowptr)
    for k in range(6):
        got = deg[pc == k].mean()
        assert abs(got - distinct[k]) / distinct[k] < 0.05, (k, got, distinct[k])
    # R-MAT column marginal: P(column bit = 1) = b + d = 0.24 on average
    cpc = sa.gen._popcount_np(m.colidx.astype(np.uint64)).mean() / SCALE
    assert 0.20 < cpc < 0.27
    # rows ascending, no duplicates; values in (-1, 1), exact 24-bit grid
    assert all(np.all(np.diff(m.colidx[m.rowptr[i]:m.rowptr[i + 1]]) > 0) for i in range(0, m.rows, 97))
    assert np.abs(m.vals).max() < 1.0 and abs(float(m.vals.mean())) < 0.01


def test_any_row_range_is_generated_without_the_rest_of_the_graph():
    sa, E = _graph()
    full = sa.gen.rmat_rows(SCALE, E, 0, 1 << SCALE, device="cpu")
    rpp = 1 << (SCALE - sa.gen.rmat_piece_bits(SCALE))
    for r0, r1 in ((0, rpp), (2 * rpp, 5 * rpp), (7 * rpp, 8 * rpp), (3 * rpp, 3 * rpp)):
        part = sa.gen.rmat_rows(SCALE, E, r0, r1, device="cpu")
        a0, a1 = full.rowptr[r0], full.rowptr[r1]
        assert np.array_equal(part.rowptr, full.rowptr[r0:r1 + 1] - a0)
        assert np.array_equal(part.colidx, full.colidx[a0:a1]) and np.array_equal(part.vals, full.vals[a0:a1])
        assert part.cols == 1 << SCALE
    with pytest.raises(ValueError):
        sa.gen.rmat_rows(SCALE, E, 1, rpp, device="cpu")                 # not piece-aligned
    # another seed is another graph
    other = sa.gen.rmat_rows(SCALE, E, 0, rpp, seed=4, device="cpu")
    assert not np.array_equal(other.colidx[:100], full.colidx[:100])


def test_cuts_by_expected_cost_balance_the_real_costs():
    """the cost model behind the cuts (nonzeros of sparsely filled 64 x 64 blocks + 25 per well-filled block kept as an MFMA tile + 34 per row: gen.rmat_piece_table)
    evaluated on the REAL graph: the parts cut from the marginals alone carry equal shares of it"""
    sa, E = _graph()
    full = sa.gen.rmat_rows(SCALE, E, 0, 1 << SCALE, device="cpu")
    rows_of = np.repeat(np.arange(full.rows), np.diff(full.rowptr))
    blk = (rows_of // 64) * ((1 << SCALE) // 64) + full.colidx // 64
    ids, cnt = np.unique(blk, return_counts=True)
    br = ids // ((1 << SCALE) // 64)
    tb_cost, row_cost = sa.gen.rmat_cost_constants(256)                  # (what rmat_cuts prices a tile block / a row of C at by default)
    cost_br = np.bincount(br, weights=np.where(cnt >= 120, tb_cost, cnt.astype(np.float64)), minlength=full.rows // 64) + 64.0 * row_cost
    model = sa.gen.rmat_block_model(SCALE, E)
    # the model's totals against the graph's: nonzeros outside well-filled blocks, number of well-filled blocks
    from math import comb
    T = SCALE - 6
    pred_sparse = sum(comb(T, k) * model[k, 0] for k in range(T + 1))
    pred_tiles = sum(comb(T, k) * model[k, 1] for k in range(T + 1))
    assert abs(pred_sparse - cnt[cnt < 120].sum()) / pred_sparse < 0.05 and abs(pred_tiles - (cnt >= 120).sum()) / max(pred_tiles, 1.0) < 0.15
    for parts in (2, 4, 8):
        cuts = sa.gen.rmat_cuts(SCALE, E, parts)
        assert cuts[0][0] == 0 and cuts[-1][1] == 1 << SCALE and all(cuts[i][1] == cuts[i + 1][0] for i in range(parts - 1))
        real = np.array([cost_br[r0 // 64:r1 // 64].sum() for r0, r1 in cuts], np.float64)
        assert real.max() / real.mean() < 1.3, (parts, real)           # granularity of a piece at this small scale; ~1.03 at 2^23
        # the hub: the first part has far fewer rows than the last
        assert (cuts[0][1] - cuts[0][0]) < (cuts[-1][1] - cuts[-1][0])           # (rows have a price of their own since round 4: less skew than by nonzeros alone)


def test_dense_operand_rows_are_canonical():
    import torch
    sa, _ = _graph()
    B = sa.gen.dense_rhs_rows(0, 200, 5, seed=7, device="cpu")
    B2 = sa.gen.dense_rhs_rows(64, 200, 5, seed=7, device="cpu")
    assert torch.equal(B.view(5, 200)[:, 64:], B2.view(5, 136))
    assert float(B.abs().max()) <= 0.5 and abs(float(B.mean())) < 0.05
    Bh = sa.gen.dense_rhs_rows(0, 200, 5, seed=7, dtype=torch.float16, device="cpu")
    assert torch.equal(Bh, B.to(torch.float16))


# ---- two ranks over gloo: what bench_parts.run does per rank, with the oracle in the kernel's place ----------------------------------
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
        E = sa.gen.rmat_raw_edges_for_density(SCALE, DENSITY)
        r0, r1 = sa.gen.rmat_cuts(SCALE, E, world)[rank]                                  # no collective, no other rank's rows
        m = sa.gen.rmat_rows(SCALE, E, r0, r1, device="cpu")
        n = 1 << SCALE
        shard_rows = n // world
        shard = sa.gen.dense_rhs_rows(rank * shard_rows, (rank + 1) * shard_rows, N, seed=7, device="cpu")
        gathered = torch.empty(world * shard_rows * N, dtype=torch.float32)
        sa.dist.allgather_B(shard, gathered)                                               # the one exchange step
        g = sa.BlockingEngine(blocking_algo="fixed_size", row_block_size=64, col_block_size=W).GetGrouping(m)
        v = sa.VBR().fill_from_CSR_inplace(m, g, W, 64, False)
        Bfull = sa.dist.gathered_to_colmajor(gathered.numpy(), world, shard_rows, N)
        C = O.vbr_multiply(v.rows, v.cols, W, v.row_part, v.nzcount, v.jab, v.mab, Bfull, N)
        np.savez(os.path.join(out_dir, "part%d.npz" % rank), C=C, r0=r0, r1=r1, perm=sa.get_permutation(g), nnz=m.nztot())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_generate_only_their_parts_and_tile_the_graph(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    import sparta_amd as sa
    from oracle import oracle as O
    E = sa.gen.rmat_raw_edges_for_density(SCALE, DENSITY)
    full = sa.gen.rmat_rows(SCALE, E, 0, 1 << SCALE, device="cpu")
    B = sa.gen.dense_rhs_rows(0, 1 << SCALE, N, seed=7, device="cpu").numpy()
    truth = O.csr_multiply(full.rows, full.rowptr, full.colidx.astype(np.int64), full.vals, B, full.cols, N).reshape(N, full.rows)
    res = [np.load(os.path.join(str(tmp_path), "part%d.npz" % r)) for r in range(2)]
    assert int(res[0]["r0"]) == 0 and int(res[0]["r1"]) == int(res[1]["r0"]) and int(res[1]["r1"]) == 1 << SCALE
    assert int(res[0]["nnz"]) + int(res[1]["nnz"]) == full.nztot()
    for r in res:
        rows = np.arange(int(r["r0"]), int(r["r1"]))[r["perm"]]
        C = r["C"].reshape(N, len(rows))
        assert np.array_equal(C, truth[:, rows])                       # same products in the same (ascending-column) order: bit-identical


def test_bench_gpus_n_starts_its_own_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` without a launcher spawns torch.distributed.run as a CHILD (never an exec of a process that holds the GPU)
    and returns its exit code; in this container the ranks then stop at "needs a GPU" -- which is the evidence that they were started."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rmat-scale", "12"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert p.stderr.count("bench.py needs a GPU") >= 1 and "ChildFailedError" in p.stderr, p.stderr[-2000:]     # (torchrun may stop the second rank before it prints)
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("torch.distributed.run\", \"--nnodes=1\"") < main.index("    import torch\n"), "the ranks must be started before this process imports torch"
    assert "os.exec" not in src
