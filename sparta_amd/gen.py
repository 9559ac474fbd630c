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
