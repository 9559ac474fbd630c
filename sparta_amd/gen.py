"""Seeded synthetic inputs shaped like BASELINE.json's configs (there is no network: SuiteSparse / OGB
matrices cannot be fetched, so each config has a generator that reproduces its size and structure).
All generators use numpy's PCG64 with an explicit seed and return a sparta_amd.CSR."""
import numpy as np


def _csr_from_coo(rows, cols, r, c, v=None):
    from .host import CSR
    key = r.astype(np.int64) * cols + c.astype(np.int64)
    key, idx = np.unique(key, return_index=True)          # sorts by (row, col) and drops duplicates
    r2 = (key // cols).astype(np.int64)
    c2 = (key % cols).astype(np.int32)
    rowptr = np.zeros(rows + 1, np.int64)
    np.add.at(rowptr, r2 + 1, 1)
    rowptr = np.cumsum(rowptr)
    vals = None if v is None else np.ascontiguousarray(v[idx], np.float32)
    return CSR(rows, cols, rowptr, c2, vals)


def uniform_random(n_rows, n_cols, nnz, seed=1234, pattern_only=False):
    """Config 1 family: exactly `nnz` distinct (row, col) positions uniformly at random, values U(-1, 1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    flat = rng.choice(n_rows * n_cols, size=int(nnz), replace=False)
    r, c = flat // n_cols, flat % n_cols
    v = None if pattern_only else rng.uniform(-1.0, 1.0, size=len(flat)).astype(np.float32)
    return _csr_from_coo(n_rows, n_cols, r, c, v)


def config1(seed=1234):
    """Synthetic 4096x4096 CSR at 1 % nnz (BASELINE.json configs[0])."""
    n = 4096
    return uniform_random(n, n, int(0.01 * n * n), seed)


def fem3d(nx, ny, nz, dof=3, seed=2, pattern_only=False, col_offset=0, total_cols=None):
    """Config 2 family ("cant"-like): stiffness-matrix pattern of a hexahedral FEM mesh of nx*ny*nz nodes with
    `dof` unknowns per node, 27-point node coupling, dense dof x dof blocks; rows ordered node-major (x fastest).
    SuiteSparse `cant` is 62 451 = 3 * 9*9*257 rows with 64.2 nnz/row; fem3d(9, 9, 257) gives the same row count,
    69.3 nnz/row and the same banded, 3-row-clustered structure.  col_offset/total_cols embed the pattern in a
    wider matrix (used for the weak-scaling multi-GPU workload)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nn = nx * ny * nz
    x, y, z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    x, y, z = x.ravel(), y.ravel(), z.ravel()
    node = x + nx * (y + ny * z)
    rr, cc = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) & (z + dz >= 0) & (z + dz < nz)
                a = node[ok]
                b = (x[ok] + dx) + nx * ((y[ok] + dy) + ny * (z[ok] + dz))
                for di in range(dof):
                    for dj in range(dof):
                        rr.append(a * dof + di)
                        cc.append(b * dof + dj)
    r = np.concatenate(rr)
    c = np.concatenate(cc) + col_offset
    n = nn * dof
    v = None if pattern_only else rng.uniform(-1.0, 1.0, size=len(r)).astype(np.float32)
    return _csr_from_coo(n, n if total_cols is None else total_cols, r, c, v)


def cant_like(seed=2, **kw):
    return fem3d(9, 9, 257, 3, seed, **kw)


def rmat(scale, n_edges, a=0.57, b=0.19, c=0.19, seed=3, symmetrize=False, pattern_only=True, row_range=None):
    """Configs 3-5 family: R-MAT power-law graph with 2**scale vertices; duplicates removed.
    row_range=(lo, hi) keeps only those rows (a rank's partition)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = 1 << scale
    r = np.zeros(n_edges, np.int64)
    col = np.zeros(n_edges, np.int64)
    ab, abc = a + b, a + b + c
    for lvl in range(scale):
        u = rng.random(n_edges)
        right = ((u >= a) & (u < ab)) | (u >= abc)       # quadrants b, d -> column bit set
        down = u >= ab                                    # quadrants c, d -> row bit set
        r |= down.astype(np.int64) << lvl
        col |= right.astype(np.int64) << lvl
    if symmetrize:
        r, col = np.concatenate([r, col]), np.concatenate([col, r])
    if row_range is not None:
        keep = (r >= row_range[0]) & (r < row_range[1])
        r, col = r[keep], col[keep]
    v = None if pattern_only else rng.uniform(-1.0, 1.0, size=len(r)).astype(np.float32)
    return _csr_from_coo(n, n, r, col, v)


def banded(n, half_bandwidth, density=1.0, seed=4, pattern_only=False):
    """n x n band matrix; each in-band entry kept with probability `density`."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rr, cc = [], []
    for d in range(-half_bandwidth, half_bandwidth + 1):
        i = np.arange(max(0, -d), min(n, n - d))
        keep = rng.random(len(i)) < density if density < 1.0 else np.ones(len(i), bool)
        rr.append(i[keep])
        cc.append(i[keep] + d)
    r, c = np.concatenate(rr), np.concatenate(cc)
    v = None if pattern_only else rng.uniform(-1.0, 1.0, size=len(r)).astype(np.float32)
    return _csr_from_coo(n, n, r, c, v)


def dense_rhs(n_rows, n_cols, seed=7, lo=-0.5, hi=0.5):
    """Dense operand B, column-major flat (ld = n_rows), U(lo, hi)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.uniform(lo, hi, size=n_rows * n_cols).astype(np.float32)


def fem3d_slab(nx, ny, nz_per_rank, rank, world, dof=3, pad_to=64, seed=2, pattern_only=False):
    """Row slab `rank` of the fem3d matrix of an nx x ny x (world*nz_per_rank) mesh, for row-partitioned multi-GPU
    runs.  Rows are the slab's own unknowns (local numbering); columns are GLOBAL in the padded numbering
    col = owner_rank * n_pad + local_index, n_pad = ceil(n_local / pad_to) * pad_to, so that every rank's shard of B
    has n_pad rows (a multiple of the column-block width: sparta_vbs_spmm_gathered) and the slab couples to its
    z-neighbours' shards through the 27-point stencil.  Returns (CSR, n_local, n_pad)."""
    rng = np.random.Generator(np.random.PCG64(seed + 1000 * rank))
    n_local = nx * ny * nz_per_rank * dof
    n_pad = -(-n_local // pad_to) * pad_to
    nz_tot = nz_per_rank * world
    x, y, z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz_per_rank), indexing="ij")
    x, y, z = x.ravel(), y.ravel(), z.ravel()
    node_local = x + nx * (y + ny * z)
    zg = z + rank * nz_per_rank
    rr, cc = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) & (zg + dz >= 0) & (zg + dz < nz_tot)
                a = node_local[ok]
                z2 = zg[ok] + dz
                owner = z2 // nz_per_rank
                b_local = (x[ok] + dx) + nx * ((y[ok] + dy) + ny * (z2 % nz_per_rank))
                for di in range(dof):
                    for dj in range(dof):
                        rr.append(a * dof + di)
                        cc.append(owner * n_pad + b_local * dof + dj)
    r, c = np.concatenate(rr), np.concatenate(cc)
    v = None if pattern_only else rng.uniform(-1.0, 1.0, size=len(r)).astype(np.float32)
    return _csr_from_coo(n_local, world * n_pad, r, c, v), n_local, n_pad


# ---- large power-law inputs, generated on the GPU (BASELINE configs[2]-[4]: 10^8 .. 10^10 nonzeros) -----------------------------
def rmat_device(scale, n_edges=None, target_nnz=None, a=0.57, b=0.19, c=0.19, seed=3, symmetrize=False, values="uniform", n=None,
                device=0, chunk=1 << 26, max_rounds=12, return_stats=False, row_slab=None):
    """R-MAT graph with 2**scale vertices (optionally only the leading n x n corner), sampled, sorted and de-duplicated on the
    GPU with torch (numpy needs minutes for 10^9 edges); the CSR comes back as host arrays (sparta_amd.CSR).

    n_edges     : raw edges to draw (before removing duplicates), or
    target_nnz  : keep drawing until at least this many DISTINCT entries exist (a stated density = target_nnz / n^2); the result
                  has slightly more (the last round's surplus is kept: dropping entries of a sorted list would bias the rows).
    symmetrize  : every edge (r, c) also gives (c, r).
    values      : "uniform" U(-1, 1) fp32, "ones", or None (pattern only).
    row_slab    : (k, parts), parts a power of two: only the rows [k * 2**scale / parts, (k + 1) * 2**scale / parts) of the graph, as a
                  (2**scale / parts) x 2**scale matrix -- the top log2(parts) bits of the row are fixed and the column bits of those levels are
                  drawn from the conditional distribution, so a slab of a graph too large for one GPU is sampled without the rest (the
                  R-MAT bits are independent across levels).  `target_nnz` / `n_edges` then refer to the slab.  No symmetrize, no corner.
    Same seed, same torch build, same GPU model -> same matrix; the seed of chunk k is seed * 1000003 + k."""
    import torch
    from .host import CSR
    if not torch.cuda.is_available():
        raise RuntimeError("rmat_device needs a GPU (use gen.rmat on the host for small cases)")
    dev = torch.device("cuda", device)
    full = 1 << scale
    n = full if n is None else int(n)
    ab, abc = a + b, a + b + c
    state = {"chunk": 0, "drawn": 0}
    slab_bits, slab_k, n_rows = 0, 0, n
    if row_slab is not None:
        slab_k, parts = int(row_slab[0]), int(row_slab[1])
        slab_bits = parts.bit_length() - 1
        if (1 << slab_bits) != parts or not (0 <= slab_k < parts) or symmetrize or n != full or slab_bits > scale:
            raise ValueError("row_slab = (k, parts): parts a power of two <= 2**scale, 0 <= k < parts, no symmetrize, no corner")
        n_rows = full >> slab_bits

    def draw(m):
        """m raw edges -> int64 keys r * n + c (edges outside the n x n corner rejected)"""
        out = []
        left = int(m)
        while left > 0:
            k = min(left, chunk)
            g = torch.Generator(device=dev).manual_seed(int(seed) * 1000003 + state["chunk"])
            state["chunk"] += 1
            r = torch.zeros(k, dtype=torch.int64, device=dev)
            col = torch.zeros(k, dtype=torch.int64, device=dev)
            for lvl in range(scale):
                u = torch.rand(k, generator=g, device=dev)
                if lvl >= scale - slab_bits:                       # a level whose row bit the slab fixes: the column bit given that row bit
                    down_bit = (slab_k >> (lvl - (scale - slab_bits))) & 1
                    p_right = (1.0 - abc) / (1.0 - ab) if down_bit else b / ab
                    col |= (u < p_right).to(torch.int64) << lvl
                    continue                                       # (the row bits of the slab are implicit: rows are numbered inside the slab)
                right = ((u >= a) & (u < ab)) | (u >= abc)        # quadrants b, d -> column bit set
                down = u >= ab                                     # quadrants c, d -> row bit set
                r |= down.to(torch.int64) << lvl
                col |= right.to(torch.int64) << lvl
            if n < full:
                keep = (r < n) & (col < n)
                r, col = r[keep], col[keep]
            key = r * n + col
            out.append(key)
            if symmetrize:
                out.append(col * n + r)
            left -= k
            state["drawn"] += k
        return torch.cat(out) if len(out) > 1 else out[0]

    if target_nnz is None:
        keys = torch.unique(draw(int(n_edges)))
    else:
        keys = torch.empty(0, dtype=torch.int64, device=dev)
        want = int(target_nnz)
        for _ in range(max_rounds):
            missing = want - keys.numel()
            if missing <= 0:
                break
            # duplicates: assume the next batch is at best as fresh as the last one was (first round: 1 / 0.7)
            m = int(missing * 1.35 / (2 if symmetrize else 1)) + 1024
            fresh = draw(m)
            keys = torch.unique(torch.cat([keys, fresh]))
            del fresh
    nnz = int(keys.numel())
    r = keys // n
    counts = torch.bincount(r, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr[1:])
    colidx = (keys - r * n).to(torch.int32)
    del keys, r, counts
    vals = None
    if values == "uniform":
        g = torch.Generator(device=dev).manual_seed(int(seed) * 7919 + 17)
        vals = torch.rand(nnz, generator=g, device=dev) * 2.0 - 1.0
    elif values == "ones":
        vals = torch.ones(nnz, dtype=torch.float32, device=dev)
    m = CSR(n_rows, n, rowptr.cpu().numpy(), colidx.cpu().numpy(), None if vals is None else vals.cpu().numpy())
    del rowptr, colidx, vals
    torch.cuda.empty_cache()
    if return_stats:
        return m, {"raw_edges_drawn": state["drawn"], "nnz": nnz, "density": nnz / float(n_rows) / float(n)}
    return m


def ogbn_products_like(seed=3, device=0, **kw):
    """BASELINE configs[2] stand-in (ogbn-products itself cannot be fetched): 2 449 029 vertices, ~61.86 M undirected edges
    symmetrised (~124 M nonzeros, values 1), degrees power-law: the leading 2 449 029^2 corner of an R-MAT 2^22 graph
    (SURVEY.md section 8(d))."""
    n = 2449029
    return rmat_device(22, target_nnz=2 * 61859140, seed=seed, symmetrize=True, values="ones", n=n, device=device, **kw)


# ---- canonical R-MAT: the graph is a pure function of (scale, raw_edges, seed); any row range can be sampled alone -------------
# BASELINE configs[3] / configs[4] at their stated sizes do not fit one generator pass (7e9 .. 5.5e10 nonzeros), and a row-partitioned
# job must not make every rank generate the whole graph.  Here the 2**scale rows are cut into canonical PIECES of PIECE_ROWS rows (an
# aligned dyadic range: the top bits of the row are fixed).  Piece p receives round(raw_edges * P(piece p)) raw edges -- its expected
# share, the R-MAT bits being independent across levels -- and every raw edge of a piece is a pure function of (seed, piece, index in the
# piece): a counter-based hash (splitmix64 finaliser) supplies 16 bits per level; at the levels the piece fixes only the column bit is
# drawn, from its conditional distribution given the row bit.  Duplicates are removed per piece; the value of an entry is a hash of
# (seed, row, column).  So rank r of an N-rank job and slab r of a one-GPU run produce bit-identical rows, whatever N is.
# SATURATED pieces (configs[3] at 1 % and 5 %: the hub rows are nearly full, a piece would draw 2 .. 40 raw edges per cell) are sampled
# cell by cell instead: cell (i, j) is present with probability 1 - exp(-raw_edges * p_ij) -- the Poisson limit of the same model, the one
# rmat_row_model counts with -- decided by a hash of (seed, i, j).  A piece is saturated when its raw edges reach half its cells: a rule of
# (scale, raw_edges) alone, so every rank agrees.
PIECE_ROWS_LOG2 = 10
_M64 = (1 << 64) - 1


def _s64(x):
    """python int -> the int64 with the same low 64 bits"""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


_GOLD, _MIX1, _MIX2 = _s64(0x9E3779B97F4A7C15), _s64(0xBF58476D1CE4E5B9), _s64(0x94D049BB133111EB)
_K_PIECE, _K_SEED, _K_VAL = _s64(0xD6E8FEB86659FD93), _s64(0xA0761D6478BD642F), _s64(0xE7037ED1A0B428DB)


def _mix64(z):
    """splitmix64 finaliser on an int64 torch tensor (wrapping arithmetic, logical shifts)"""
    z = z + _GOLD
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _MIX1
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _MIX2
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def rmat_piece_bits(scale):
    """number of row bits a canonical piece fixes (pieces = 2**that): pieces of 1024 rows from 2^20 rows up, of 64 rows below (tests,
    miniatures: the hub piece must stay a small share of the graph for the cuts to balance)"""
    return max(0, int(scale) - (PIECE_ROWS_LOG2 if int(scale) >= 20 else 6))


def rmat_row_model(scale, raw_edges, a=0.57, b=0.19, c=0.19):
    """Expected number of DISTINCT entries of a row as a function of its popcount k (k = 0 .. scale), for `raw_edges` raw R-MAT edges:
    sum over the columns j of 1 - exp(-raw_edges * p_ij); the columns are grouped by (ones at the row's zero levels, ones at its one
    levels).  Returns (distinct[k], raw_share[k]) -- everything a rank needs to cut the rows by cost WITHOUT seeing the graph."""
    from math import comb
    s = int(scale)
    d = 1.0 - a - b - c
    p1 = c + d                                            # P(row bit = 1)
    q0, q1 = b / (a + b), d / (c + d)                     # P(column bit = 1 | row bit = 0 / 1)
    distinct = np.zeros(s + 1)
    share = np.zeros(s + 1)
    for k in range(s + 1):
        prow = (p1 ** k) * ((1.0 - p1) ** (s - k))
        share[k] = prow
        u = np.arange(s - k + 1)
        v = np.arange(k + 1)
        mu = np.array([comb(s - k, int(x)) for x in u], np.float64)
        mv = np.array([comb(k, int(x)) for x in v], np.float64)
        qu = (q0 ** u) * ((1.0 - q0) ** (s - k - u))
        qv = (q1 ** v) * ((1.0 - q1) ** (k - v))
        lam = float(raw_edges) * prow * np.outer(qu, qv)
        distinct[k] = float((np.outer(mu, mv) * (-np.expm1(-lam))).sum())
    return distinct, share


def rmat_raw_edges_for_density(scale, density, a=0.57, b=0.19, c=0.19):
    """raw edge count whose expected number of distinct entries is density * 4**scale (bisection on the row model)"""
    from math import comb
    s = int(scale)
    want = float(density) * float(1 << s) * float(1 << s)
    mult = np.array([comb(s, k) for k in range(s + 1)], np.float64)
    lo, hi = want, want * 64.0 + 1024.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        got = float((mult * rmat_row_model(s, mid, a, b, c)[0]).sum())
        if got < want:
            lo = mid
        else:
            hi = mid
        if hi - lo <= max(1.0, 1e-9 * hi):
            break
    return int(round(hi))


def _popcount_np(x):
    x = np.asarray(x, np.uint64)
    out = np.zeros(x.shape, np.int64)
    while True:
        nz = x != 0
        if not nz.any():
            return out
        out += (x & np.uint64(1)).astype(np.int64)
        x = x >> np.uint64(1)


def rmat_block_model(scale, raw_edges, a=0.57, b=0.19, c=0.19, block=64, tile_min_nnz=120.0):
    """What the fixed block x block grid of the graph looks like, per popcount kr of a block-row's top (scale - log2 block) row bits:
    (expected nonzeros in blocks with fewer than tile_min_nnz expected nonzeros, expected number of the other -- well-filled -- blocks,
    expected nonzeros of the block-row).  The hybrid builder keeps the well-filled blocks as MFMA tiles and sends the nonzeros of the
    others through the sparse-row kernels (vbs_build.cpp: SPARTA_SPARSE_K_BLOCK = 60 per step, two steps per 64 x 64 16-bit block);
    a tile costs about as much as tile_min_nnz gathered nonzeros, so the two shares are what a part's product time is made of."""
    from math import comb, factorial
    s = int(scale)
    lb = int(block).bit_length() - 1
    T = s - lb
    d = 1.0 - a - b - c
    # the cells of one block: the low lb levels, grouped by how many levels fall in each quadrant
    comps = []
    for n00 in range(lb + 1):
        for n01 in range(lb + 1 - n00):
            for n10 in range(lb + 1 - n00 - n01):
                n11 = lb - n00 - n01 - n10
                mult = factorial(lb) / (factorial(n00) * factorial(n01) * factorial(n10) * factorial(n11))
                comps.append((mult, (a ** n00) * (b ** n01) * (c ** n10) * (d ** n11)))
    cm = np.array([x[0] for x in comps])
    cp = np.array([x[1] for x in comps])
    out = np.zeros((T + 1, 3))
    for kr in range(T + 1):
        u = np.arange(T - kr + 1)
        v = np.arange(kr + 1)
        mu = np.array([comb(T - kr, int(x)) for x in u], np.float64)
        mv = np.array([comb(kr, int(x)) for x in v], np.float64)
        F = float(raw_edges) * np.outer((a ** (T - kr - u)) * (b ** u), (c ** (kr - v)) * (d ** v))          # (u, v): the block's factor
        mult = np.outer(mu, mv)
        blk_nnz = (cm[None, None, :] * (-np.expm1(-F[:, :, None] * cp[None, None, :]))).sum(axis=2)
        tile = blk_nnz >= tile_min_nnz
        out[kr, 0] = float((mult * blk_nnz * ~tile).sum())
        out[kr, 1] = float((mult * tile).sum())
        out[kr, 2] = float((mult * blk_nnz).sum())
    return out


def rmat_cost_constants(n_cols=256):
    """(tile_block_cost, row_cost) of rmat_piece_table for a product with n_cols columns of B, in units of one nonzero on the sparse-row path.  Least squares over the part
    times of the round-4 records (profiles/r4, one MI355X, hub kernel + scalar gather + XCD streams): configs[4] (fp16, N = 256, 8 parts) 0.0425 ns per gathered nonzero,
    41.6 / 92.0 nonzero-equivalents per tile block / row of C (the eight part times within 4.8 %); configs[3] (bf16, N = 512, its 30 parts at 0.1 / 1 / 5 % jointly) 0.0805 ns,
    68.5 / 106.4 (within 4.2 %).  The parts those records ran on were cut with 25 / 34 (the first fit of the round): max / mean 1.16 (configs[4]), 1.11 / 1.19 / 1.30."""
    return (42.0, 92.0) if int(n_cols) <= 256 else (68.0, 106.0)


def rmat_piece_table(scale, raw_edges, a=0.57, b=0.19, c=0.19, tile_block_cost=42.0, row_cost=92.0):
    """Per canonical piece: (raw edges it receives, expected cost of its product).  Cost unit: one nonzero on the sparse-row path (one
    gathered row of B); a well-filled 64 x 64 block kept as an MFMA tile costs `tile_block_cost` of them, a row of C `row_cost` (least
    squares over part times on one MI355X: rmat_cost_constants holds the current fits, the defaults here are those of configs[4].  First fit of round 4: 0.054 ns per gathered nonzero,
    1.35 ns per tile, 1.85 ns per row = 25 / 34; round 3: 0.102 ns, 5.6 ns per tile, rows ~ 0: the parts cut with those
    coefficients ran 34.3-42.3 ms, max / mean 1.12, once the tiles had become four times cheaper);
    the model's own counts match the builder's: part 0 predicted 432.86 M sparse nonzeros + 1.4208 M tiles, built 432.82 M + 1.4212 M).  Pure arithmetic on the R-MAT marginals: every
    rank computes the same table without seeing the graph."""
    s = int(scale)
    f = rmat_piece_bits(s)
    p1 = 1.0 - a - b
    pc = _popcount_np(np.arange(1 << f, dtype=np.uint64))
    share = (p1 ** pc) * ((1.0 - p1) ** (f - pc))
    raw = np.floor(float(raw_edges) * share + 0.5).astype(np.int64)
    from math import comb
    low = s - f                                           # row bits inside a piece
    lb = 6                                                # a 64-row block-row fixes all but the low 6 row bits
    if low < lb:
        raise ValueError("pieces must hold whole 64-row block-rows")
    bm = rmat_block_model(s, raw_edges, a, b, c)
    per_blockrow = bm[:, 0] + tile_block_cost * bm[:, 1] + row_cost * 64.0         # by popcount of the block-row's top bits
    mid = low - lb                                        # row bits of a block-row's id that vary inside the piece
    mid_mult = np.array([comb(mid, k) for k in range(mid + 1)], np.float64)
    per_pc = np.array([float((mid_mult * per_blockrow[t:t + mid + 1]).sum()) for t in range(f + 1)])
    cost = per_pc[pc]
    return raw, cost


def _popcount_t(x):
    """popcount of the low 32 bits of an int64 torch tensor (SWAR)"""
    x = x - ((x >> 1) & 0x55555555)
    x = (x & 0x33333333) + ((x >> 2) & 0x33333333)
    x = (x + (x >> 4)) & 0x0F0F0F0F
    return ((x * 0x01010101) >> 24) & 0xFF


def _rmat_piece_cellwise(torch, dev, s, raw_edges, row0, rpp, seed_term, a, b, c, rows_per_chunk=None):
    """A saturated piece, cell by cell: returns (counts per row, column ids int32) of rows [row0, row0 + rpp)."""
    import math
    n = 1 << s
    d = 1.0 - a - b - c
    la, lb, lc, ld = (math.log(x) for x in (a, b, c, d))
    cols = torch.arange(n, dtype=torch.int64, device=dev)
    pc_c = _popcount_t(cols).to(torch.float32)
    rows_per_chunk = rows_per_chunk or max(1, (1 << 26) // n)
    counts, out = [], []
    for q0 in range(0, rpp, rows_per_chunk):
        r = torch.arange(row0 + q0, row0 + min(rpp, q0 + rows_per_chunk), dtype=torch.int64, device=dev)
        n11 = _popcount_t(r[:, None] & cols[None, :]).to(torch.float32)
        pc_r = _popcount_t(r).to(torch.float32)[:, None]
        n10 = pc_r - n11                                                   # row bit 1, column bit 0: quadrant c
        n01 = pc_c[None, :] - n11                                          # row bit 0, column bit 1: quadrant b
        n00 = float(s) - pc_r - pc_c[None, :] + n11
        lam = torch.exp(n00 * la + n01 * lb + n10 * lc + n11 * ld + math.log(float(raw_edges)))
        p = -torch.expm1(-lam)
        del n11, n10, n01, n00, lam
        hv = _mix64((r[:, None] * n + cols[None, :]) * _K_PIECE + seed_term)
        u = (((hv >> 40) & 0xFFFFFF).to(torch.float32) + 0.5) * (1.0 / 16777216.0)
        present = u < p
        del hv, u, p
        counts.append(present.sum(dim=1))
        out.append(present.nonzero()[:, 1].to(torch.int32))               # row-major order: rows ascending, columns ascending inside a row
        del present
    return torch.cat(counts), torch.cat(out)


def rmat_piece_is_cellwise(scale, raw_of_piece):
    """the canonical rule: a piece whose raw edges reach half its cells is sampled cell by cell"""
    s = int(scale)
    rpp = 1 << (s - rmat_piece_bits(s))
    return np.asarray(raw_of_piece, np.float64) * 2.0 >= float(rpp) * float(1 << s)


def rmat_cuts(scale, raw_edges, parts, a=0.57, b=0.19, c=0.19, n_cols=256):
    """`parts` contiguous row ranges [(row0, row1), ...] of equal expected cost (sparse nonzeros + tile blocks + rows, priced for a product with n_cols columns of B:
    rmat_cost_constants), cut at piece boundaries."""
    from .dist import partition_by_cost
    tb, rc = rmat_cost_constants(n_cols)
    _, cost = rmat_piece_table(scale, raw_edges, a, b, c, tile_block_cost=tb, row_cost=rc)
    rows_per_piece = 1 << (int(scale) - rmat_piece_bits(scale))
    return [(p0 * rows_per_piece, p1 * rows_per_piece) for p0, p1 in partition_by_cost(cost, parts)]


def rmat_rows(scale, raw_edges, row0, row1, seed=3, a=0.57, b=0.19, c=0.19, values="uniform", device=0, chunk=1 << 26, group=1 << 28,
              return_stats=False):
    """Rows [row0, row1) (piece-aligned) of THE R-MAT graph (scale, raw_edges, seed) as a (row1 - row0) x 2**scale CSR, generated with
    torch on `device` (an int = that GPU, or "cpu" for tests) without the rest of the graph."""
    import torch
    from .host import CSR
    s = int(scale)
    f = rmat_piece_bits(s)
    low = s - f
    rpp = 1 << low
    if row0 % rpp or row1 % rpp or not (0 <= row0 <= row1 <= (1 << s)):
        raise ValueError("row range must be aligned to pieces of %d rows" % rpp)
    dev = torch.device("cpu") if device == "cpu" else torch.device("cuda", int(device))
    n = 1 << s
    p0, p1 = row0 // rpp, row1 // rpp
    raw_all, _ = rmat_piece_table(s, raw_edges, a, b, c)
    raw = raw_all[p0:p1]
    A16, AB16, ABC16 = int(round(a * 65536)), int(round((a + b) * 65536)), int(round((a + b + c) * 65536))
    d = 1.0 - a - b - c
    Q0, Q1 = int(round(b / (a + b) * 65536)), int(round(d / (c + d) * 65536))
    seed_term = _s64(int(seed) * (_K_SEED & _M64))
    counts_parts, col_parts = [], []
    drawn = 0
    cellwise = rmat_piece_is_cellwise(s, raw)
    n_cellwise = int(cellwise.sum())
    # groups of whole pieces (duplicates never cross a piece: pieces are disjoint in rows), each at most `group` raw edges if possible
    g0 = 0
    npieces = p1 - p0
    while g0 < npieces:
        if cellwise[g0]:
            cnt_, col_ = _rmat_piece_cellwise(torch, dev, s, raw_edges, (p0 + g0) * rpp, rpp, seed_term, a, b, c)
            counts_parts.append(cnt_)
            col_parts.append(col_)
            g0 += 1
            continue
        g1, tot = g0, 0
        while g1 < npieces and not cellwise[g1] and (g1 == g0 or tot + int(raw[g1]) <= group):
            tot += int(raw[g1])
            g1 += 1
        cum = torch.from_numpy(np.concatenate([[0], np.cumsum(raw[g0:g1])]).astype(np.int64)).to(dev)
        keys = []
        e0 = 0
        while e0 < tot:
            e1 = min(tot, e0 + chunk)
            gi = torch.arange(e0, e1, dtype=torch.int64, device=dev)
            pl = torch.searchsorted(cum, gi, right=True) - 1            # piece inside the group
            idx = gi - cum[pl]
            piece = pl + (p0 + g0)
            del gi
            x = piece * _K_PIECE + idx + seed_term
            r = torch.zeros_like(idx)
            col = torch.zeros_like(idx)
            h = None
            for lvl in range(s):
                if lvl % 4 == 0:
                    h = _mix64(x + _s64((lvl // 4 + 1) * (_GOLD & _M64)))
                u = (h >> (16 * (lvl % 4))) & 0xFFFF
                if lvl >= low:                                            # a level the piece fixes: the column bit given the row bit
                    down = (piece >> (lvl - low)) & 1
                    thr = torch.where(down == 1, Q1, Q0)
                    col |= (u < thr).to(torch.int64) << lvl
                else:
                    right = ((u >= A16) & (u < AB16)) | (u >= ABC16)
                    down = u >= AB16
                    r |= down.to(torch.int64) << lvl
                    col |= right.to(torch.int64) << lvl
            keys.append((pl * rpp + r) * n + col)                          # row inside the group
            del x, r, col, h, u, pl, idx, piece
            e0 = e1
        drawn += tot
        if keys:
            k = torch.unique(torch.cat(keys) if len(keys) > 1 else keys[0])
        else:
            k = torch.empty(0, dtype=torch.int64, device=dev)
        del keys
        rr = k // n
        counts_parts.append(torch.bincount(rr, minlength=(g1 - g0) * rpp))
        col_parts.append((k - rr * n).to(torch.int32))
        del k, rr
        g0 = g1
    n_rows = row1 - row0
    counts = torch.cat(counts_parts) if counts_parts else torch.zeros(0, dtype=torch.int64, device=dev)
    colidx = torch.cat(col_parts) if col_parts else torch.zeros(0, dtype=torch.int32, device=dev)
    del counts_parts, col_parts
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr[1:])
    nnz = int(colidx.numel())
    vals = None
    if values is not None:
        vals_parts = []
        rows_of = torch.repeat_interleave(torch.arange(n_rows, dtype=torch.int64, device=dev), counts) + row0
        for e0 in range(0, nnz, chunk):
            e1 = min(nnz, e0 + chunk)
            hv = _mix64((rows_of[e0:e1] * n + colidx[e0:e1].to(torch.int64)) * _K_VAL + seed_term)
            if values == "uniform":
                vals_parts.append((((hv >> 40) & 0xFFFFFF).to(torch.float32) + 0.5) * (2.0 / 16777216.0) - 1.0)       # 24 bits: exact in fp32
            else:
                vals_parts.append(torch.ones(e1 - e0, dtype=torch.float32, device=dev))
        vals = torch.cat(vals_parts) if vals_parts else torch.zeros(0, dtype=torch.float32, device=dev)
        del rows_of, vals_parts
    m = CSR(n_rows, n, rowptr.cpu().numpy(), colidx.cpu().numpy(), None if vals is None else vals.cpu().numpy())
    del rowptr, colidx, vals, counts
    if dev.type == "cuda":
        torch.cuda.empty_cache()
    if return_stats:
        return m, {"raw_edges_drawn": int(drawn), "nnz": nnz, "rows": [int(row0), int(row1)], "pieces": [int(p0), int(p1)], "cellwise_pieces": n_cellwise}
    return m


def dense_rhs_rows(row0, row1, n_cols, seed=7, dtype=None, device=0, chunk_rows=None):
    """Rows [row0, row1) of THE dense operand B (seed): B[j, c] = a hash of (seed, j, c) mapped to U(-0.5, 0.5) on a 2**-16 grid (exact in
    fp16 / bf16 only up to their rounding; the same for every rank count).  Returned column-major (ld = row1 - row0), flat, on `device`."""
    import torch
    dev = torch.device("cpu") if device == "cpu" else torch.device("cuda", int(device))
    dtype = torch.float32 if dtype is None else dtype
    nr = int(row1) - int(row0)
    out = torch.empty(nr * int(n_cols), dtype=dtype, device=dev)
    ov = out.view(int(n_cols), nr)
    seed_term = _s64(int(seed) * (_K_SEED & _M64))
    j = torch.arange(int(row0), int(row1), dtype=torch.int64, device=dev)
    for cix in range(int(n_cols)):
        hv = _mix64((j * 65536 + cix) * _K_VAL + seed_term)
        ov[cix] = ((((hv >> 40) & 0xFFFF).to(torch.float32) + 0.5) * (1.0 / 65536.0) - 0.5).to(dtype)
    return out
