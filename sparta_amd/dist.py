"""Row-range partitioning of A across the GPUs of one node, B replicated by ONE all-gather (RCCL over xGMI).

The reference is single-GPU; this is the multi-GPU extension BASELINE.json's north_star asks
for.  Block-rows are independent (each writes a disjoint row range of C, vbr.cpp:355) and B is read-only, so:
  * A is split into contiguous block-row ranges balanced by executed work  sum h*w*nb  (not by row count: clustered /
    power-law matrices have very uneven block-rows);
  * every rank owns the rows of B that correspond to its column shard and one `all_gather_into_tensor` assembles the
    n_shards column-major slabs on every GPU -- exactly the layout sparta_vbs_spmm_gathered consumes, no repacking;
  * C stays row-partitioned: there is no collective on C.
One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm ("gloo" on CPU for the tests).

Second exchange mode (RowBlockExchange): a slab of A usually touches only a few row-blocks of the other ranks' shards of B
(a FEM slab: one plane of halo on each side; a clustered graph: its neighbours).  Replicating all of B then moves
(world-1) x |shard| bytes per rank and step for nothing.  Here every rank tells the others once, at plan time, which
column-blocks its slab touches; per step ONE all-to-all (RCCL send/recv pairs on the point-to-point xGMI links) ships
exactly those row-blocks, while the product with the rank's OWN shard -- which needs no communication -- runs
concurrently; the product with the received blocks is added afterwards.  A slab that touches everything degenerates to
the all-gather's traffic; `needed_fraction` tells which mode pays."""
import numpy as np


def partition_block_rows(row_part, nzcount, block_col_size, world_size):
    """Contiguous block-row ranges [(b0, b1), ...] (one per rank, possibly empty at the end) balanced by the dense
    work each block-row executes, h * w * nb.  Greedy prefix split at the ideal cumulative targets."""
    row_part = np.asarray(row_part, np.int64)
    nzcount = np.asarray(nzcount, np.int64)
    h = np.diff(row_part)
    # +h so that empty block-rows (which still write zeros into C) carry a little weight
    work = h * int(block_col_size) * nzcount + h
    return partition_by_cost(work, world_size)


def partition_by_cost(cost, world_size):
    """Contiguous ranges [(b0, b1), ...] of the items (block-rows), one per rank, cut at the ideal cumulative targets of
    `cost` (any non-negative per-item cost: executed dense work for MFMA-carried matrices, nonzeros + rows for the
    sparse-row-carried power-law ones).  Ranks at the end may get an empty range when there are fewer items than ranks."""
    cost = np.asarray(cost, np.float64)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        b = int(np.searchsorted(cum, target, side="left"))
        if 0 < b <= len(cost) and target - cum[b - 1] < cum[min(b, len(cost))] - target:
            b -= 1                                                           # the cut nearer to the target (a heavy item goes to the emptier side)
        bounds.append(min(max(b, bounds[-1]), len(cost)))
    bounds.append(len(cost))
    return [(bounds[i], bounds[i + 1]) for i in range(world_size)]


def row_slab(cmat, rows, n_cols=None):
    """The CSR made of the given rows of `cmat` (original row ids, in the order given), columns unchanged; `n_cols` widens the
    column space (padding up to world * shard_rows).  One rank's slab of a row-range partition."""
    from .host import CSR
    rows = np.asarray(rows, np.int64)
    rp = np.asarray(cmat.rowptr, np.int64)
    cnt = rp[rows + 1] - rp[rows]
    rowptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    total = int(rowptr[-1])
    # position of every kept nonzero in the source arrays
    src = np.repeat(rp[rows] - rowptr[:-1], cnt) + np.arange(total, dtype=np.int64)
    colidx = np.asarray(cmat.colidx)[src]
    vals = None if cmat.vals is None else np.asarray(cmat.vals)[src]
    return CSR(len(rows), int(cmat.cols if n_cols is None else n_cols), rowptr, colidx, vals)


def gathered_rows(B_gathered, row_ids, world_size, shard_rows, n_cols, shard_ld=None):
    """(checks) rows `row_ids` of B out of the gathered device buffer: returns an n_cols x len(row_ids) float64 numpy array"""
    import torch
    ids = torch.as_tensor(np.asarray(row_ids, np.int64), device=B_gathered.device)
    shard_ld = shard_rows if shard_ld is None else shard_ld
    g = B_gathered.view(world_size, n_cols, shard_ld)                        # slab s, column j, local row (rows past shard_rows: padding)
    out = g[ids // shard_rows, :, ids % shard_rows]                          # len(ids) x n_cols
    return out.float().cpu().numpy().astype(np.float64).T


def padded_shard_ld(shard_rows, elem_bytes):
    """elements between the columns of a rank's slab of B: shard_rows, plus 64 when the columns would lie a multiple of 4 KB apart -- columns a large power of
    two apart fall onto the same cache sets and memory channels (a panel of B is 8..16 columns per load instruction): sparta_vbs_spmm_gathered_ld, DESIGN.md 12"""
    return int(shard_rows) + (64 if (int(shard_rows) * int(elem_bytes)) % 4096 == 0 else 0)


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


def allgather_B_peer_copies(B_shard, B_gathered, rank, world_size, group=None):
    """The same result as allgather_B by world_size - 1 point-to-point sends and receives posted together
    (`batch_isend_irecv`): every rank sends its slab straight to each peer and receives each peer's slab into its place in the
    gathered buffer.  On MI355X every pair of GPUs of a node has its own xGMI link, so the 7 copies of a rank run on 7 links at
    once, where a ring all-gather moves (world - 1) / world of the buffer over one link per direction;
    which one wins depends on slab size and RCCL's choice of algorithm -- `pick_allgather` measures both."""
    import torch.distributed as dist
    n = B_shard.numel()
    # validated on every rank BEFORE any op is posted: a rank that raised inside a half-posted batch would leave its peers hanging
    if B_gathered.dim() != 1 or B_shard.dim() != 1 or not B_gathered.is_contiguous() or not B_shard.is_contiguous() or B_gathered.numel() != world_size * n:
        raise ValueError("allgather_B_peer_copies: B_shard and B_gathered must be flat contiguous buffers with B_gathered.numel() == world_size * B_shard.numel() "
                         "(a padded shard stride is not supported here: use the collective)")
    B_gathered[rank * n:(rank + 1) * n].copy_(B_shard)
    ops = []
    for k in range(1, world_size):                            # rank r sends to r + k while it receives from r - k: no two posts of a pair cross
        dst, src = (rank + k) % world_size, (rank - k) % world_size
        ops.append(dist.P2POp(dist.isend, B_shard, dst, group=group))
        ops.append(dist.P2POp(dist.irecv, B_gathered[src * n:(src + 1) * n], src, group=group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    return B_gathered


def pick_allgather(B_shard, B_gathered, rank, world_size, reps=3, sync=None, group=None):
    """Plan-time choice between the collective all-gather and the peer copies: both are run (first for equality of the result, then
    `reps` timed repetitions each, wall clock between two `sync()` calls, maximum over the ranks); every rank returns the same
    {'mode': 'all_gather' | 'peer_copies', 'all_gather_ms', 'peer_copies_ms', 'equal'}.  Collective: call it on every rank.  B_gathered is
    overwritten (zeroed in between, and left filled by whichever variant ran last -- both leave the same gathered B)."""
    import time
    import torch
    import torch.distributed as dist
    sync = sync or (lambda: None)
    ref = allgather_B(B_shard, B_gathered, group=group).clone()
    B_gathered.zero_()
    allgather_B_peer_copies(B_shard, B_gathered, rank, world_size, group=group)
    sync()
    equal = torch.tensor([1.0 if torch.equal(ref, B_gathered) else 0.0], device=B_gathered.device)
    ms = []
    for fn in (lambda: allgather_B(B_shard, B_gathered, group=group),
               lambda: allgather_B_peer_copies(B_shard, B_gathered, rank, world_size, group=group)):
        fn(); sync()
        dist.barrier(group=group)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        ms.append((time.perf_counter() - t0) * 1e3 / reps)
    t = torch.tensor(ms + [float(equal.item())], dtype=torch.float64, device=B_gathered.device)
    dist.all_reduce(t[:2], op=dist.ReduceOp.MAX, group=group)
    eq = t[2:3].clone()
    dist.all_reduce(eq, op=dist.ReduceOp.MIN, group=group)
    ag, pc, ok = float(t[0]), float(t[1]), float(eq[0]) > 0.5
    return {"mode": "peer_copies" if ok and pc < 0.95 * ag else "all_gather", "all_gather_ms": round(ag, 4), "peer_copies_ms": round(pc, 4), "equal": ok}


def gathered_to_colmajor(B_gathered, world_size, shard_rows, n_cols):
    """(host/numpy helper for tests) gathered slabs -> one column-major (world*shard_rows) x n_cols matrix, flat"""
    g = np.asarray(B_gathered).reshape(world_size, n_cols, shard_rows)      # slab s, column j, local row
    return np.ascontiguousarray(g.transpose(1, 0, 2)).reshape(-1)           # column j, slab s, local row


# ---- sparsity-aware exchange: only the row-blocks of B a slab touches ---------------------------------------------

def to_block_tiles(B_colmajor, rows, n_cols, w):
    """(numpy helper) column-major rows x n_cols (ld = rows, rows % w == 0) -> row-block-tiled layout: block jb (rows
    [jb*w, (jb+1)*w) of B) is one contiguous w x n_cols column-major tile (ld = w) at element offset jb * w * n_cols.
    This is the layout RowBlockExchange keeps B in: a needed row-block is ONE contiguous chunk to send, and the SpMM kernel
    reads it in place through sparta_vbs_spmm_gathered(shard_rows = w, shard_stride = w * n_cols)."""
    B = np.asarray(B_colmajor).reshape(n_cols, rows // w, w)
    return np.ascontiguousarray(B.transpose(1, 0, 2)).reshape(-1)


def from_block_tiles(B_tiles, rows, n_cols, w):
    B = np.asarray(B_tiles).reshape(rows // w, n_cols, w)
    return np.ascontiguousarray(B.transpose(1, 0, 2)).reshape(-1)


def needed_blocks(jab, block_col_size, shard_rows, world_size):
    """Per source rank q: the sorted local ids (inside shard q) of the column-blocks that occur in `jab` (global ids in the
    padded numbering  owner * shard_rows / w + local)."""
    bps = int(shard_rows) // int(block_col_size)
    u = np.unique(np.asarray(jab, np.int64))
    if len(u) and (u[0] < 0 or u[-1] >= bps * world_size):
        raise ValueError("column-block id outside world_size * shard_rows / block_col_size")
    owner = u // bps
    return [(u[owner == q] - q * bps).astype(np.int32) for q in range(world_size)]


def split_own_remote(vb, rank, shard_rows, world_size):
    """Column split of a slab's VBS (same block-rows, same values, blocks in the same order):
      own    -- the blocks whose column-block lies in shard `rank`, ids local to the shard   (cols = shard_rows)
      remote -- the others, ids compacted in (source rank, block) order = the order the all-to-all delivers the tiles
                (cols = number of needed remote blocks * w; None when the slab touches no other shard)
    plus need[q] (needed_blocks).  own * B_own + remote * B_received == vb * B up to the order of the fp32 sums."""
    from .host import VBR
    w = int(vb.block_col_size)
    bps = int(shard_rows) // w
    if int(shard_rows) % w or vb.cols != world_size * shard_rows:
        raise ValueError("cols must be world_size * shard_rows and shard_rows a multiple of block_col_size")
    need = needed_blocks(vb.jab, w, shard_rows, world_size)
    jab = np.asarray(vb.jab, np.int64)
    nzc = np.asarray(vb.nzcount, np.int64)
    h = np.diff(np.asarray(vb.row_part, np.int64))
    ib = np.repeat(np.arange(len(nzc)), nzc)                   # block-row of every block
    size = h[ib] * w                                           # stored elements of every block
    off = np.concatenate([[0], np.cumsum(size)])
    is_own = (jab // bps) == rank
    remote_ids = np.concatenate([need[q].astype(np.int64) + q * bps for q in range(world_size) if q != rank]
                                + [np.zeros(0, np.int64)])

    def take(mask, new_ids, new_cols):
        sel = np.flatnonzero(mask)
        sz = size[sel]
        total = int(sz.sum())
        start = np.concatenate([[0], np.cumsum(sz)])[:-1]
        idx = np.repeat(off[sel] - start, sz) + np.arange(total)
        mab = np.asarray(vb.mab, np.float32)[idx] if total else np.zeros(0, np.float32)
        return VBR.from_arrays(vb.rows, new_cols, w, vb.row_part, np.bincount(ib[sel], minlength=len(nzc)), new_ids, mab)

    own = take(is_own, jab[is_own] - rank * bps, int(shard_rows))
    remote = None
    if len(remote_ids):
        remote = take(~is_own, np.searchsorted(remote_ids, jab[~is_own]), len(remote_ids) * w)
    return own, remote, need


class RowBlockExchange:
    """Per-step exchange of only the needed row-blocks of B + the two-part product (see the module docstring).

        ex = RowBlockExchange(vb_slab, rank, world, shard_rows, n_cols, device=local_rank)       # collective (plan exchange)
        ex.step(B_own_tiles, C)          # B_own_tiles: this rank's shard in the row-block-tiled layout (to_block_tiles)

    `pack` / `product` can be replaced (the CPU tests over gloo pass numpy-backed ones; the product path is the HIP
    library and has no CPU form).  `all_need[p][q]` (every rank's needed_blocks) skips the plan-time collective."""

    def __init__(self, vb, rank, world_size, shard_rows, n_cols, device=0, group=None, pack=None, product=None, all_need=None, dtype=0):
        import torch
        import torch.distributed as dist
        self.rank, self.world, self.w, self.N = int(rank), int(world_size), int(vb.block_col_size), int(n_cols)
        self.shard_rows, self.group, self.device = int(shard_rows), group, device
        self.own, self.remote, self.need = split_own_remote(vb, rank, shard_rows, world_size)
        tile = self.w * self.N
        if all_need is not None:                  # single-process drivers / tests: every rank's need lists, already known
            all_need = [[np.asarray(a).tolist() for a in per_rank] for per_rank in all_need]
        elif self.world > 1:
            all_need = [None] * self.world
            dist.all_gather_object(all_need, [a.tolist() for a in self.need], group=group)
        else:
            all_need = [[a.tolist() for a in self.need]]
        send = [np.asarray(all_need[p][self.rank], np.int32) if p != self.rank else np.zeros(0, np.int32) for p in range(self.world)]
        bps = self.shard_rows // self.w
        for a in send:
            if len(a) and (a.min() < 0 or a.max() >= bps):
                raise ValueError("a peer asked for a row-block outside this rank's shard")
        self.send_ids_host = np.concatenate(send + [np.zeros(0, np.int32)]).astype(np.int32)
        self.in_splits = [len(a) * tile for a in send]
        self.out_splits = [len(self.need[q]) * tile if q != self.rank else 0 for q in range(self.world)]
        self.n_send, self.n_recv = len(self.send_ids_host), sum(self.out_splits) // tile
        # every rank must agree on whether the collective is called at all
        self.any_exchange = any(len(all_need[p][q]) for p in range(self.world) for q in range(self.world) if p != q)
        # fraction of the all-gather's per-rank traffic this exchange still moves (max over ranks): 1.0 = nothing saved
        self.needed_fraction = (max(sum(len(all_need[p][q]) for q in range(self.world) if q != p) for p in range(self.world))
                                / float(max(1, (self.world - 1) * bps)))
        dev = torch.device("cpu") if device is None else torch.device("cuda", device)
        self.dtype = dtype                     # storage type of A and B on the device (sparta_amd.F32 / F16 / BF16); C is fp32
        tdt = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[int(dtype)]
        self.elem_bytes = 4 if int(dtype) == 0 else 2
        self.send_ids = torch.from_numpy(self.send_ids_host).to(dev)
        self.send_buf = torch.empty(max(1, self.n_send * tile), dtype=tdt, device=dev)
        self.recv_buf = torch.empty(max(1, self.n_recv * tile), dtype=tdt, device=dev)
        self._send_view, self._recv_view = self.send_buf[:self.n_send * tile], self.recv_buf[:self.n_recv * tile]
        self._pack = pack if pack is not None else self._pack_hip
        self._product = product if product is not None else self._product_hip
        self.d_own = self.d_rem = None
        self._side = None
        if product is None:
            self.d_own = self.own.to_device(device, dtype=dtype)
            self.d_rem = self.remote.to_device(device, dtype=dtype) if self.remote is not None else None

    # -- HIP implementations (the product) --
    def _pack_hip(self, B_tiles):
        import ctypes as C
        import torch
        from ._lib import lib, check
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.sparta_pack_blocks(C.c_void_p(B_tiles.data_ptr()), self.w * self.N * self.elem_bytes, C.c_void_p(self.send_ids.data_ptr()), self.n_send,
                                     C.c_void_p(self.send_buf.data_ptr()), C.c_void_p(st)))

    def _product_hip(self, which, B_tiles, C_out, accumulate):
        d = self.d_own if which == "own" else self.d_rem
        d.spmm_gathered(B_tiles, self.w, C_out, self.N, accumulate=accumulate)

    def _all_to_all(self):
        """the ONE collective of a step (asynchronous: returns the work handle)"""
        import torch.distributed as dist
        return dist.all_to_all_single(self._recv_view, self._send_view, self.out_splits, self.in_splits, group=self.group, async_op=True)

    def step(self, B_own_tiles, C_out, accumulate=False):
        """C (+)= A_slab * B with B distributed: pack -> all-to-all (asynchronous) || own-shard product -> remote product."""
        work = pack_done = None
        if self.any_exchange and self.device is not None:
            # pack + collective go to a side stream so that the own-shard product (which needs neither) starts at once
            import torch
            main = torch.cuda.current_stream(self.device)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
                self._pack_ev = torch.cuda.Event()
            self._side.wait_stream(main)                       # B_own_tiles is ready when `main` gets here
            with torch.cuda.stream(self._side):
                if self.n_send:
                    self._pack(B_own_tiles)
                    pack_done = self._pack_ev
                    pack_done.record(self._side)
                work = self._all_to_all()
        elif self.any_exchange:
            if self.n_send:
                self._pack(B_own_tiles)
            work = self._all_to_all()
        self._product("own", B_own_tiles, C_out, accumulate)
        if work is not None:
            work.wait()                                        # the current stream waits for the collective
            if self.remote is not None:
                self._product("remote", self.recv_buf, C_out, True)
        if pack_done is not None:
            torch.cuda.current_stream(self.device).wait_event(pack_done)    # the caller may overwrite B_own_tiles after step()

    def close(self):
        for d in (self.d_own, self.d_rem):
            if d is not None:
                d.close()
        self.d_own = self.d_rem = None
