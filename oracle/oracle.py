"""ctypes binding of oracle/liboracle.so -- the plain-C restatement of the reference's hot path
(sparta_oracle.c; every function there cites the reference file:line it follows).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, os.path.join(_HERE, "liboracle.so")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        L = C.CDLL(_PATH)
        lp, fp = C.POINTER(C.c_long), C.POINTER(C.c_float)
        dist_args = [lp, C.c_long, C.c_long, lp, C.c_long, C.c_long, C.c_long]
        L.oracle_hamming_distance_group.restype = C.c_float
        L.oracle_hamming_distance_group.argtypes = dist_args
        L.oracle_jaccard_distance_group.restype = C.c_float
        L.oracle_jaccard_distance_group.argtypes = dist_args
        L.oracle_merge_rows.restype = C.c_long
        L.oracle_merge_rows.argtypes = [lp, C.c_long, lp, C.c_long, lp]
        L.oracle_get_permutation.argtypes = [lp, C.c_long, lp]
        L.oracle_get_partition.restype = C.c_long
        L.oracle_get_partition.argtypes = [lp, C.c_long, lp]
        L.oracle_get_fixed_size_grouping.argtypes = [lp, C.c_long, C.c_long, lp]
        L.oracle_get_grouping.argtypes = [C.c_long, lp, lp, C.c_int, C.c_int, C.c_float, C.c_long, C.c_long, C.c_int, C.c_int,
                                          C.c_int, lp, lp]
        L.oracle_vbr_fill_inplace.argtypes = [C.c_long, C.c_long, lp, lp, fp, lp, C.c_long, C.c_long, C.c_int, lp, lp, lp, lp, fp]
        L.oracle_vbr_multiply.argtypes = [C.c_long, C.c_long, C.c_long, C.c_long, lp, lp, lp, fp, fp, C.c_int, fp]
        L.oracle_vbr_multiply_range.argtypes = [C.c_long, C.c_long, C.c_long, lp, lp, lp, fp, C.c_long, C.c_long, fp, C.c_int, fp]
        L.oracle_csr_multiply.argtypes = [C.c_long, lp, lp, fp, fp, C.c_long, C.c_long, fp]
        L.oracle_collect_blocking_info.argtypes = [C.c_long, C.c_long, lp, lp, lp, C.c_long, lp, fp]
        _lib = L
    return _lib


def _l(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def distance(sim_measure, row_a, group_a, row_b, group_b, block_size):
    a, b = _l(row_a), _l(row_b)
    f = lib().oracle_jaccard_distance_group if (sim_measure & 1) else lib().oracle_hamming_distance_group
    return float(f(_lp(a), len(a), group_a, _lp(b), len(b), group_b, block_size))


def merge_rows(row_a, row_b):
    a, b = _l(row_a), _l(row_b)
    out = np.zeros(len(a) + len(b) + 1, np.int64)
    n = lib().oracle_merge_rows(_lp(a), len(a), _lp(b), len(b), _lp(out))
    return out[:n].copy()


def get_permutation(grouping):
    g = _l(grouping)
    out = np.zeros(len(g), np.int64)
    lib().oracle_get_permutation(_lp(g), len(g), _lp(out))
    return out


def get_partition(grouping):
    g = _l(grouping)
    out = np.zeros(len(g) + 2, np.int64)
    n = lib().oracle_get_partition(_lp(g), len(g), _lp(out))
    return out[:n].copy()


def get_fixed_size_grouping(grouping, row_block_size):
    g = _l(grouping)
    out = np.zeros(len(g), np.int64)
    lib().oracle_get_fixed_size_grouping(_lp(g), len(g), row_block_size, _lp(out))
    return out


def get_grouping(rows, rowptr, colidx, blocking_algo=3, sim_measure=1, tau=0.5, col_block_size=1, row_block_size=1,
                 use_groups=False, use_pattern=True, force_fixed_size=False, structured_m=2, structured_n=4):
    rp, ci = _l(rowptr), _l(colidx)
    g = np.zeros(rows, np.int64)
    cnt = np.zeros(2, np.int64)
    L = lib()
    L.oracle_get_grouping_mn.argtypes = [C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_int, C.c_int, C.c_float, C.c_long, C.c_long,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    rc = L.oracle_get_grouping_mn(rows, _lp(rp), _lp(ci), blocking_algo, sim_measure, tau, col_block_size, row_block_size,
                                  int(use_groups), int(use_pattern), int(force_fixed_size), int(structured_m), int(structured_n), _lp(g), _lp(cnt))
    if rc != 0:
        raise NotImplementedError("oracle does not restate blocking_algo %d" % blocking_algo)
    return g, dict(comparison_counter=int(cnt[0]), merge_counter=int(cnt[1]))


class OracleVBR:
    """The five arrays of the reference's VBR, built by the oracle's fill_from_CSR_inplace restatement."""

    def __init__(self, rows, cols, rowptr, colidx, vals, grouping, col_block_size, row_block_size=0, force_fixed_size=False):
        rp, ci, g = _l(rowptr), _l(colidx), _l(grouping)
        v = None if vals is None else np.ascontiguousarray(vals, np.float32)
        dims = np.zeros(6, np.int64)
        L = lib()
        L.oracle_vbr_fill_inplace(rows, cols, _lp(rp), _lp(ci), _fp(v), _lp(g), col_block_size, row_block_size,
                                  int(force_fixed_size), _lp(dims), None, None, None, None)
        self.rows, self.cols, self.block_rows, self.block_cols, self.nztot, self.nblocks = map(int, dims)
        self.block_col_size = int(col_block_size)
        self.row_part = np.zeros(self.block_rows + 1, np.int64)
        self.nzcount = np.zeros(self.block_rows, np.int64)
        self.jab = np.zeros(max(self.nblocks, 1), np.int64)
        self.mab = np.zeros(max(self.nztot, 1), np.float32)
        L.oracle_vbr_fill_inplace(rows, cols, _lp(rp), _lp(ci), _fp(v), _lp(g), col_block_size, row_block_size,
                                  int(force_fixed_size), _lp(dims), _lp(self.row_part), _lp(self.nzcount), _lp(self.jab),
                                  _fp(self.mab))
        self.jab = self.jab[:self.nblocks]
        self.mab = self.mab[:self.nztot]


def vbr_multiply(rows, cols, block_col_size, row_part, nzcount, jab, mab, B, n_cols, C_in=None, block_row_range=None):
    """C (+)= A*B by the oracle's VBR::multiply restatement. Returns C (column-major flat, ld = rows)."""
    rp, nz, jb = _l(row_part), _l(nzcount), _l(jab)
    m = np.ascontiguousarray(mab, np.float32)
    Bc = np.ascontiguousarray(B, np.float32)
    Cm = np.zeros(rows * n_cols, np.float32) if C_in is None else np.ascontiguousarray(C_in, np.float32).copy()
    b0, b1 = (0, len(nz)) if block_row_range is None else block_row_range
    lib().oracle_vbr_multiply_range(rows, cols, block_col_size, _lp(rp), _lp(nz), _lp(jb), _fp(m), b0, b1, _fp(Bc), n_cols, _fp(Cm))
    return Cm


def csr_multiply(rows, rowptr, colidx, vals, B, ldb, n_cols):
    rp, ci = _l(rowptr), _l(colidx)
    v = None if vals is None else np.ascontiguousarray(vals, np.float32)
    Bc = np.ascontiguousarray(B, np.float32)
    Cm = np.zeros(rows * n_cols, np.float32)
    lib().oracle_csr_multiply(rows, _lp(rp), _lp(ci), _fp(v), _fp(Bc), ldb, n_cols, _fp(Cm))
    return Cm


def collect_blocking_info(rows, cols, rowptr, colidx, grouping, col_block_size):
    rp, ci, g = _l(rowptr), _l(colidx), _l(grouping)
    info = np.zeros(3, np.int64)
    avg = C.c_float(0)
    lib().oracle_collect_blocking_info(rows, cols, _lp(rp), _lp(ci), _lp(g), col_block_size, _lp(info), C.byref(avg))
    return dict(VBR_nzcount=int(info[0]), VBR_nzblocks_count=int(info[1]), VBR_longest_row=int(info[2]),
                VBR_average_height=float(avg.value))


def usable_cpus():
    """the CPUs this process may actually use: its affinity mask, capped by the cgroup's CPU quota (a container that sees 256 hardware threads behind a quota of 16 CPUs runs
    16 threads' worth of work however many it starts)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def vbr_multiply_mt(rows, cols, block_col_size, row_part, nzcount, jab, mab, B, n_cols, block_row_range=None, n_threads=None, C_out=None):
    """The same restatement of VBR::multiply on several host threads: block-rows are independent (each writes its own rows of C,
    vbr.cpp:355), so contiguous ranges of them -- balanced by executed multiply-adds -- run concurrently (ctypes releases the
    GIL).  TEST / BASELINE INFRASTRUCTURE: the "all host cores" figure next to the reference-faithful single-thread one.
    Returns (C, seconds of the threaded region)."""
    import os
    import time
    from concurrent.futures import ThreadPoolExecutor
    rp, nz, jb = _l(row_part), _l(nzcount), _l(jab)
    m = np.ascontiguousarray(mab, np.float32)
    Bc = np.ascontiguousarray(B, np.float32)
    Cm = np.zeros(rows * n_cols, np.float32) if C_out is None else C_out
    b0, b1 = (0, len(nz)) if block_row_range is None else block_row_range
    if n_threads is None:
        n_threads = usable_cpus()
    n_threads = max(1, min(int(n_threads), b1 - b0))
    work = (np.diff(rp)[b0:b1] * nz[b0:b1]).astype(np.float64) + 1.0
    cum = np.concatenate([[0.0], np.cumsum(work)])
    cuts = [b0 + int(np.searchsorted(cum, cum[-1] * t / n_threads, side="left")) for t in range(n_threads)] + [b1]
    fn = lib().oracle_vbr_multiply_range
    args = (rows, cols, block_col_size, _lp(rp), _lp(nz), _lp(jb), _fp(m))

    def run(t):
        if cuts[t] < cuts[t + 1]:
            fn(*args, cuts[t], cuts[t + 1], _fp(Bc), n_cols, _fp(Cm))

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=n_threads) as ex:
        list(ex.map(run, range(n_threads)))
    return Cm, time.perf_counter() - t0, n_threads
