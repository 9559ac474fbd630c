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
