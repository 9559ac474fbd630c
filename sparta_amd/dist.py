"""Row-range partitioning of A across the GPUs of one node, B replicated by ONE all-gather (RCCL over xGMI).

The reference is single-GPU; this is the multi-GPU extension BASELINE.json's north_star asks
for.  Block-rows are independent (each writes a disjoint row range of C, vbr.cpp:355) and B is read-only, so:
  * A is split into contiguous block-row ranges balanced by executed work  sum h*w*nb  (not by row count: clustered /
    power-law matrices have very uneven block-rows);
  * every rank owns the rows of B that correspond to its column shard and one `all_gather_into_tensor` assembles the
    n_shards column-major slabs on every GPU -- exactly the layout sparta_vbs_spmm_gathered consumes, no repacking;
  * C stays row-partitioned: there is no collective on C.
One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm ("gloo" on CPU for the tests)."""
import numpy as np


def partition_block_rows(row_part, nzcount, block_col_size, world_size):
    """Contiguous block-row ranges [(b0, b1), ...] (one per rank, possibly empty at the end) balanced by the dense
    work each block-row executes, h * w * nb.  Greedy prefix split at the ideal cumulative targets."""
    row_part = np.asarray(row_part, np.int64)
    nzcount = np.asarray(nzcount, np.int64)
    h = np.diff(row_part)
    # +h so that empty block-rows (which still write zeros into C) carry a little weight
    work = h * int(block_col_size) * nzcount + h
    cum = np.concatenate([[0], np.cumsum(work)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        b = int(np.searchsorted(cum, target, side="left"))
        b = min(max(b, bounds[-1]), len(work))
        bounds.append(b)
    bounds.append(len(work))
    return [(bounds[i], bounds[i + 1]) for i in range(world_size)]


def padded_shard_rows(n_local, block_col_size):
    """rows per B shard: the local column count rounded up to a multiple of the column-block width"""
    w = int(block_col_size)
    return -(-int(n_local) // w) * w


def allgather_B(B_shard, B_gathered, group=None):
    """One all-gather of the ranks' (shard_rows x N, column-major) slabs of B into the gathered buffer
    (world * shard_rows * N elements) that sparta_vbs_spmm_gathered reads.  Works on any torch.distributed
    backend (nccl == RCCL on the GPUs, gloo on CPU tensors)."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(B_gathered, B_shard, group=group)
    return B_gathered


def gathered_to_colmajor(B_gathered, world_size, shard_rows, n_cols):
    """(host/numpy helper for tests) gathered slabs -> one column-major (world*shard_rows) x n_cols matrix, flat"""
    g = np.asarray(B_gathered).reshape(world_size, n_cols, shard_rows)      # slab s, column j, local row
    return np.ascontiguousarray(g.transpose(1, 0, 2)).reshape(-1)           # column j, slab s, local row
