"""Host-side mirror of the reference's CSR / BlockingEngine / VBR interface (same names, argument
meaning and defaults), implemented by the host C++ in libsparta_amd.so."""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import lib, check

# BlockingType (include/definitions.h:17) -- values of the `-a` flag
BLOCKING_ALGOS = dict(iterative=0, iterative_structured=1, fixed_size=2, iterative_clocked=3, iterative_queue=4,
                      iterative_max_size=5, scramble=6, minhash=7)   # 7: extension, not in the reference

_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


def _p64(a):
    return a.ctypes.data_as(_i64p)


def _p32(a):
    return a.ctypes.data_as(_i32p)


def _pf(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


class CSR:
    """Sparse input matrix.  The reference keeps one heap array per row (`nzcount[i]`, `ja[i][]`, `ma[i][]`,
    include/matrices.h:22-28); here the same content is flat: rowptr[rows+1] (int64), colidx (int32, ascending
    within a row), vals (float32 or None == `pattern_only`)."""

    def __init__(self, rows, cols, rowptr, colidx, vals=None):
        self.rows, self.cols = int(rows), int(cols)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        self.vals = None if vals is None else np.ascontiguousarray(vals, dtype=np.float32)
        if self.rowptr.shape != (self.rows + 1,):
            raise ValueError("rowptr must have rows+1 entries")
        if self.vals is not None and self.vals.shape != self.colidx.shape:
            raise ValueError("vals and colidx differ in length")

    @property
    def pattern_only(self):
        return self.vals is None

    @property
    def nzcount(self):
        return np.diff(self.rowptr)

    def nztot(self):
        return int(self.rowptr[-1])

    def ja(self, i):
        return self.colidx[self.rowptr[i]:self.rowptr[i + 1]]

    def ma(self, i):
        return None if self.vals is None else self.vals[self.rowptr[i]:self.rowptr[i + 1]]

    @classmethod
    def from_scipy(cls, m, pattern_only=False):
        m = m.tocsr()
        m.sort_indices()
        return cls(m.shape[0], m.shape[1], m.indptr, m.indices, None if pattern_only else m.data)

    def to_scipy(self):
        import scipy.sparse as sp
        v = np.ones(len(self.colidx), np.float32) if self.vals is None else self.vals
        return sp.csr_matrix((v, self.colidx, self.rowptr), shape=(self.rows, self.cols))

    # ---- files and row permutations (the callers either side of the path) ------------------------------------------
    @classmethod
    def read_from_edgelist(cls, path, delimiter=" ", pattern_only=False, mat_fmt=_lib.FMT_EL, symmetrize=False,
                           mode=_lib.IO_COMPAT):
        """void CSR::read_from_edgelist(infile, delimiter, pattern_only, mat_fmt, symmetrize) (src/general/csr.cpp:183-365).
        mode IO_COMPAT reproduces the reference's readers (first data line of an .el dropped, .mtx pattern-only with one
        line skipped after the size line); IO_STRICT reads the formats as documented."""
        h = _lib.CsrHost()
        check(lib.sparta_csr_read(str(path).encode(), delimiter.encode(), int(bool(pattern_only)), int(mat_fmt),
                                  int(bool(symmetrize)), int(mode), C.byref(h)))
        try:
            rowptr = np.ctypeslib.as_array(h.rowptr, shape=(h.rows + 1,)).copy()
            colidx = np.ctypeslib.as_array(h.colidx, shape=(max(h.nnz, 1),))[:h.nnz].copy()
            vals = None if h.pattern_only else np.ctypeslib.as_array(h.vals, shape=(max(h.nnz, 1),))[:h.nnz].copy()
            return cls(h.rows, h.cols, rowptr, colidx, vals)
        finally:
            lib.sparta_csr_host_free(C.byref(h))

    def save_to_edgelist(self, path, delimiter=" ", mat_fmt=_lib.FMT_EL):
        """void CSR::save_to_edgelist(outfile, delimiter, pattern_only, mat_fmt) (csr.cpp:169-179)."""
        check(lib.sparta_csr_write_edgelist(str(path).encode(), self.rows, _p64(self.rowptr), _p32(self.colidx), delimiter.encode(),
                                            int(mat_fmt)))

    def permute_rows(self, permutation):
        """void CSR::permute_rows(permutation) (csr.cpp:67-76, utilities.h:95-107): new row k = old row permutation[k]."""
        p = np.ascontiguousarray(permutation, np.int64)
        if p.shape != (self.rows,):
            raise ValueError("CSR.permute_rows argument must have the same length as rows")
        cnt = self.nzcount[p]
        rowptr = np.zeros(self.rows + 1, np.int64)
        np.cumsum(cnt, out=rowptr[1:])
        src = np.repeat(self.rowptr[:-1][p] - rowptr[:-1], cnt) + np.arange(int(rowptr[-1]), dtype=np.int64)
        self.colidx = np.ascontiguousarray(self.colidx[src])
        if self.vals is not None:
            self.vals = np.ascontiguousarray(self.vals[src])
        self.rowptr = rowptr

    def reorder(self, grouping):
        """void CSR::reorder(grouping) (csr.cpp:101-109): rows of one group become adjacent (get_permutation order)."""
        if len(grouping) != self.rows:
            raise ValueError("CSR.reorder argument must have the same length as rows")
        self.permute_rows(get_permutation(grouping))

    def reorder_by_degree(self, descending=True):
        """void CSR::reorder_by_degree(descending) (csr.cpp:123-155; flag -r 1 / -1)."""
        perm = np.empty(self.rows, np.int64)
        check(lib.sparta_degree_permutation(self.rows, _p64(self.rowptr), int(bool(descending)), _p64(perm)))
        self.permute_rows(perm)


class BlockingEngine:
    """Row-clustering reorder engine; field names and defaults of the reference's BlockingEngine
    (include/blocking.h:9-56)."""

    def __init__(self, tau=0.5, col_block_size=1, row_block_size=1, use_groups=False, use_pattern=True,
                 force_fixed_size=False, blocking_algo=3, sim_measure=1, structured_m=2, structured_n=4,
                 minhash_bands=0, minhash_rows=0, minhash_max_eval=0, minhash_max_rows=0):
        self.tau = tau
        self.col_block_size = col_block_size
        self.row_block_size = row_block_size
        self.use_groups = use_groups
        self.use_pattern = use_pattern
        self.force_fixed_size = force_fixed_size
        self.blocking_algo = BLOCKING_ALGOS.get(blocking_algo, blocking_algo)
        self.sim_measure = sim_measure          # SetComparator(choice): 0 Hamming, 1 Jaccard (blocking.cpp:699-717)
        self.structured_m, self.structured_n = structured_m, structured_n      # blocking_algo 1 (include/blocking.h:20-21)
        # blocking_algo 7 (extension: LSH-bucketed clustering); 0 = library defaults
        self.minhash_bands, self.minhash_rows, self.minhash_max_eval, self.minhash_max_rows = minhash_bands, minhash_rows, minhash_max_eval, minhash_max_rows
        self.comparison_counter = 0
        self.merge_counter = 0
        self.timer_total = 0.0
        self.timer_comparisons = 0.0
        self.timer_merges = 0.0
        self.average_row_distance = 0.0
        self.average_merge_tau = 0.0
        self.multiplication_timer_avg = 0.0
        self.multiplication_timer_std = 0.0
        self.VBR_nzcount = 0
        self.VBR_nzblocks_count = 0
        self.VBR_average_height = 0.0
        self.VBR_longest_row = 0
        self.grouping_result = None

    def SetComparator(self, choice):
        self.sim_measure = choice

    def _cfg(self):
        c = _lib.ReorderCfg()
        c.blocking_algo = int(self.blocking_algo)
        c.sim_measure = int(self.sim_measure)
        c.tau = float(self.tau)
        c.use_groups = int(bool(self.use_groups))
        c.col_block_size = int(self.col_block_size)
        c.row_block_size = int(self.row_block_size)
        c.use_pattern = int(bool(self.use_pattern))
        c.force_fixed_size = int(bool(self.force_fixed_size))
        c.structured_m, c.structured_n = int(self.structured_m), int(self.structured_n)
        c.minhash_bands, c.minhash_rows = int(self.minhash_bands), int(self.minhash_rows)
        c.minhash_max_eval, c.minhash_max_rows = int(self.minhash_max_eval), int(self.minhash_max_rows)
        return c

    def GetGrouping(self, cmat):
        """std::vector<intT> BlockingEngine::GetGrouping(const CSR&)  (blocking.cpp:633-676)"""
        g = np.empty(cmat.rows, np.int64)
        st = _lib.ReorderStats()
        cfg = self._cfg()
        check(lib.sparta_reorder(cmat.rows, cmat.cols, _p64(cmat.rowptr), _p32(cmat.colidx), C.byref(cfg), _p64(g), C.byref(st)))
        self.comparison_counter = st.comparison_counter
        self.merge_counter = st.merge_counter
        self.average_row_distance = st.average_row_distance
        self.average_merge_tau = st.average_merge_tau
        self.timer_total = st.timer_total
        self.timer_comparisons = st.timer_comparisons
        self.timer_merges = st.timer_merges
        self.grouping_result = g
        return g

    def CollectBlockingInfo(self, cmat):
        """BlockingEngine::CollectBlockingInfo (blocking.cpp:576-631)"""
        info = np.zeros(3, np.int64)
        avg = C.c_float(0)
        g = np.ascontiguousarray(self.grouping_result, np.int64)
        check(lib.sparta_blocking_info(cmat.rows, cmat.cols, _p64(cmat.rowptr), _p32(cmat.colidx), _p64(g),
                                       int(self.col_block_size), _p64(info), C.byref(avg)))
        self.VBR_nzcount, self.VBR_nzblocks_count = int(info[0]), int(info[1])
        self.VBR_longest_row = max(self.VBR_longest_row, int(info[2]))
        self.VBR_average_height = avg.value


class VBR:
    """Variable-block-sparse matrix: the fields of the reference's struct VBR (include/matrices.h:93-104)."""

    def __init__(self):
        self.rows = self.cols = self.block_rows = self.block_cols = 0
        self.block_col_size = 0
        self.nztot = 0
        self.nzcount = self.jab = self.row_part = self.mab = None
        self._dev = self._dev_t = None

    def _drop_device_images(self):
        """the cached device handles (of A for multiply, of A^T for multiply_BA) describe the OLD arrays: close and forget both"""
        for name in ("_dev", "_dev_t"):
            d = getattr(self, name, None)
            if d is not None:
                d.close()
            setattr(self, name, None)

    def fill_from_CSR_inplace(self, cmat, grouping, col_block_size, row_block_size=0, force_fixed_size=False):
        """VBR::fill_from_CSR_inplace (include/matrices.h:118, vbr.cpp:135-237)"""
        g = np.ascontiguousarray(grouping, np.int64)
        if g.shape != (cmat.rows,):
            raise ValueError("grouping must have one entry per row")
        h = _lib.VbsHost()
        check(lib.sparta_vbs_build(cmat.rows, cmat.cols, _p64(cmat.rowptr), _p32(cmat.colidx), _pf(cmat.vals), _p64(g),
                                   int(col_block_size), int(row_block_size), int(bool(force_fixed_size)), C.byref(h)))
        try:
            self.rows, self.cols = h.rows, h.cols
            self.block_rows, self.block_cols = h.block_rows, h.block_cols
            self.block_col_size, self.nztot = h.block_col_size, h.nztot
            self.row_part = np.ctypeslib.as_array(h.row_part, (h.block_rows + 1,)).copy()
            self.nzcount = np.ctypeslib.as_array(h.nzcount, (h.block_rows,)).copy()
            self.jab = np.ctypeslib.as_array(h.jab, (max(h.nblocks, 1),))[:h.nblocks].copy()
            self.mab = np.ctypeslib.as_array(h.mab, (max(h.nztot, 1),))[:h.nztot].copy()
        finally:
            lib.sparta_vbs_host_free(C.byref(h))
        self._drop_device_images()
        return self

    def _take(self, h):
        self.rows, self.cols = h.rows, h.cols
        self.block_rows, self.block_cols = h.block_rows, h.block_cols
        self.block_col_size, self.nztot = h.block_col_size, h.nztot
        self.row_part = np.ctypeslib.as_array(h.row_part, (h.block_rows + 1,)).copy()
        self.nzcount = np.ctypeslib.as_array(h.nzcount, (max(h.block_rows, 1),))[:h.block_rows].copy()
        self.jab = np.ctypeslib.as_array(h.jab, (max(h.nblocks, 1),))[:h.nblocks].copy()
        self.mab = np.ctypeslib.as_array(h.mab, (max(h.nztot, 1),))[:h.nztot].copy()

    def fill_from_CSR(self, cmat, row_partition, block_size):
        """VBR::fill_from_CSR (include/matrices.h:117, vbr.cpp:239-321): rows stay in their order, block-rows from a row partition"""
        rp = np.ascontiguousarray(row_partition, np.int64)
        h = _lib.VbsHost()
        check(lib.sparta_vbs_build_partition(cmat.rows, cmat.cols, _p64(cmat.rowptr), _p32(cmat.colidx), _pf(cmat.vals), _p64(rp), len(rp),
                                             int(block_size), C.byref(h)))
        try:
            self._take(h)
        finally:
            lib.sparta_vbs_host_free(C.byref(h))
        self._drop_device_images()
        return self

    def get_block_start(self, row_block_idx):
        """VBR::get_block_start (vbr.cpp:33-49): element offset into mab of block-row `row_block_idx`; out of range -> 0 (the
        reference prints "[RANGE ERROR]" and returns mab itself)"""
        if row_block_idx < 0 or row_block_idx >= self.block_rows:
            return 0
        h = np.diff(self.row_part[:row_block_idx + 1])
        return int((h * self.nzcount[:row_block_idx]).sum() * self.block_col_size)

    def partition_check(self, candidate_part):
        """VBR::partition_check (vbr.cpp:108-118): 0 valid, 1 empty, 2 last entry != rows, 3 decreasing"""
        p = np.ascontiguousarray(candidate_part, np.int64)
        return int(lib.sparta_vbs_partition_check(_p64(p), len(p), int(self.rows)))

    def fill_from_CSR_inplace_fixed(self, cmat, row_block_size, col_block_size, force_fixed_size=False):
        """the fixed-grid overload (vbr.cpp:121-132): grouping[i] = i / row_block_size"""
        g = np.arange(cmat.rows, dtype=np.int64) // int(row_block_size)
        return self.fill_from_CSR_inplace(cmat, g, col_block_size, row_block_size, force_fixed_size)

    @classmethod
    def from_arrays(cls, rows, cols, block_col_size, row_part, nzcount, jab, mab):
        v = cls()
        v.rows, v.cols, v.block_col_size = int(rows), int(cols), int(block_col_size)
        v.row_part = np.ascontiguousarray(row_part, np.int64)
        v.nzcount = np.ascontiguousarray(nzcount, np.int64)
        v.jab = np.ascontiguousarray(jab, np.int64)
        v.mab = np.ascontiguousarray(mab, np.float32)
        v.block_rows = len(v.nzcount)
        v.block_cols = (v.cols - 1) // v.block_col_size + 1
        v.nztot = len(v.mab)
        v._drop_device_images()
        return v

    def _host_struct(self):
        h = _lib.VbsHost()
        h.rows, h.cols, h.block_rows, h.block_cols = self.rows, self.cols, self.block_rows, self.block_cols
        h.block_col_size, h.nztot, h.nblocks = self.block_col_size, len(self.mab), len(self.jab)
        keep = (np.ascontiguousarray(self.row_part, np.int64), np.ascontiguousarray(self.nzcount, np.int64),
                np.ascontiguousarray(self.jab, np.int64), np.ascontiguousarray(self.mab, np.float32))
        h.row_part, h.nzcount, h.jab, h.mab = _p64(keep[0]), _p64(keep[1]), _p64(keep[2]), _pf(keep[3])
        return h, keep

    def to_blocked_ell(self):
        """prepare_cusparse_BLOCKEDELLPACK (src/cuda/cuda_utilities.cpp:1656-1710): -> (ell_blocksize, ellColInd[rows/bs][ell_cols] with
        -1 padding, ellValues[rows][ell_cols * bs]); needs fixed square blocks (row_block_size == col_block_size, padded sizes)."""
        h, keep = self._host_struct()
        n = C.c_int64(0)
        check(lib.sparta_vbs_to_blocked_ell(C.byref(h), C.byref(n), None, None))
        bs = int(self.block_col_size)
        ind = np.zeros((self.rows // bs, n.value), np.int64)
        val = np.zeros((self.rows, n.value * bs), np.float32)
        check(lib.sparta_vbs_to_blocked_ell(C.byref(h), C.byref(n), _p64(ind), _pf(val)))
        return bs, ind, val

    def save(self, path):
        """Binary container (sparta_vbs_save): the five arrays + checksum in one file -- the reorder is paid once."""
        h = _lib.VbsHost()
        h.rows, h.cols, h.block_rows, h.block_cols = self.rows, self.cols, self.block_rows, self.block_cols
        h.block_col_size, h.nztot, h.nblocks = self.block_col_size, len(self.mab), len(self.jab)
        rp, nz = np.ascontiguousarray(self.row_part, np.int64), np.ascontiguousarray(self.nzcount, np.int64)
        jab, mab = np.ascontiguousarray(self.jab, np.int64), np.ascontiguousarray(self.mab, np.float32)
        h.row_part, h.nzcount, h.jab, h.mab = _p64(rp), _p64(nz), _p64(jab), _pf(mab)
        check(lib.sparta_vbs_save(str(path).encode(), C.byref(h)))

    @classmethod
    def load(cls, path):
        h = _lib.VbsHost()
        check(lib.sparta_vbs_load(str(path).encode(), C.byref(h)))
        try:
            v = cls.from_arrays(h.rows, h.cols, h.block_col_size, np.ctypeslib.as_array(h.row_part, (h.block_rows + 1,)).copy(),
                                np.ctypeslib.as_array(h.nzcount, (max(h.block_rows, 1),))[:h.block_rows].copy(),
                                np.ctypeslib.as_array(h.jab, (max(h.nblocks, 1),))[:h.nblocks].copy(),
                                np.ctypeslib.as_array(h.mab, (max(h.nztot, 1),))[:h.nztot].copy())
            v.block_cols = h.block_cols
            return v
        finally:
            lib.sparta_vbs_host_free(C.byref(h))

    def to_device(self, device=0, dtype=_lib.F32, block_row_range=None):
        from .device import DeviceVBS
        return DeviceVBS(self, device=device, dtype=dtype, block_row_range=block_row_range)

    def multiply_BA(self, B, B_rows, C_out, device=0):
        """C += B * A (dense x VBS), host buffers, column-major: B is B_rows x rows, C is B_rows x cols.  The reference's
        cublas_blockmat_multiplyBA (include/cuda_utilities.h:40) has this call shape but does not compute this product (DESIGN.md
        section 8); this does.  Returns the kernel time in ms."""
        from .device import DeviceVBS
        if getattr(self, "_dev_t", None) is None or self._dev_t.device != device:
            self._dev_t = DeviceVBS.transposed_of(self, device=device)
        return self._dev_t.spmm_BA_host(B, B_rows, C_out, accumulate=True)

    def multiply(self, B, B_cols, C_out, device=0, algo=_lib.SPMM_MFMA):
        """void VBR::multiply(DataT* B, int B_cols, DataT_C* C)  (include/matrices.h:121): C += A*B with host
        buffers, column-major, ld(B) = cols, ld(C) = rows -- executed on the GPU.  Returns the kernel time in ms."""
        if self._dev is None or self._dev.device != device:
            self._dev = self.to_device(device)
        return self._dev.spmm_host(B, B_cols, C_out, accumulate=True, algo=algo)


def get_permutation(grouping):
    g = np.ascontiguousarray(grouping, np.int64)
    out = np.empty(len(g), np.int64)
    check(lib.sparta_get_permutation(_p64(g), len(g), _p64(out)))
    return out


def get_partition(grouping):
    g = np.ascontiguousarray(grouping, np.int64)
    out = np.empty(len(g) + 1, np.int64)
    n = C.c_int64(0)
    check(lib.sparta_get_partition(_p64(g), len(g), _p64(out), C.byref(n)))
    return out[:n.value].copy()


def get_fixed_size_grouping(grouping, row_block_size):
    g = np.ascontiguousarray(grouping, np.int64)
    out = np.empty(len(g), np.int64)
    check(lib.sparta_get_fixed_size_grouping(_p64(g), len(g), int(row_block_size), _p64(out)))
    return out


def row_distance(sim_measure, row_a, group_a, row_b, group_b, block_size):
    a = np.ascontiguousarray(row_a, np.int64)
    b = np.ascontiguousarray(row_b, np.int64)
    d = C.c_float(0)
    check(lib.sparta_row_distance(int(sim_measure), _p64(a), len(a), int(group_a), _p64(b), len(b), int(group_b),
                                  int(block_size), C.byref(d)))
    return d.value


def merge_rows(row_a, row_b):
    a = np.ascontiguousarray(row_a, np.int64)
    b = np.ascontiguousarray(row_b, np.int64)
    out = np.empty(len(a) + len(b) + 1, np.int64)
    n = C.c_int64(0)
    check(lib.sparta_merge_rows(_p64(a), len(a), _p64(b), len(b), _p64(out), C.byref(n)))
    return out[:n.value].copy()


def save_grouping(path, grouping):
    """The `<outfile>.g` file of save_blocking_data (src/general/utilities.cpp:239-243): one group id per line."""
    g = np.ascontiguousarray(grouping, np.int64)
    check(lib.sparta_grouping_write(str(path).encode(), _p64(g), len(g)))


def read_grouping_file(path, rows=None):
    """read_grouping_file + the leading-count rule of test/general/Matrix_Analysis.cpp:10-32,77-78."""
    cap = 1 << 16
    while True:
        out = np.empty(cap, np.int64)
        n = C.c_int64(0)
        rc = lib.sparta_grouping_read(str(path).encode(), -1 if rows is None else int(rows), _p64(out), cap, C.byref(n))
        if rc != 0 and n.value > cap:
            cap = int(n.value) + 1
            continue
        check(rc)
        return out[:n.value].copy()


CSV_COLUMNS = ("matrix", "rows", "cols", "nonzeros", "symmetrize", "blocking_algo", "tau", "row_block_size", "col_block_size",
               "use_pattern", "sim_use_groups", "sim_measure", "reorder", "exp_name", "b_cols", "warmup", "exp_repetitions",
               "multiplication_algo", "n_streams", "time_to_block", "time_to_merge", "time_to_compare", "VBR_nzcount",
               "VBR_nzblocks_count", "VBR_average_height", "VBR_longest_row", "merge_counter", "comparison_counter",
               "average_merge_tau", "average_row_distance", "avg_time_multiply", "std_time_multiply")


def blocking_csv_row(**fields):
    """(header, values) of the reference's 32-column statistics row (src/general/utilities.cpp:175-236).  Keyword names are
    the CSV column names (CSV_COLUMNS); missing ones default to 0 / ""."""
    f = _lib.CsvFields()
    names = {"VBR_nzcount": "vbr_nzcount", "VBR_nzblocks_count": "vbr_nzblocks_count", "VBR_average_height": "vbr_average_height",
             "VBR_longest_row": "vbr_longest_row"}
    unknown = set(fields) - set(CSV_COLUMNS)
    if unknown:
        raise TypeError("unknown CSV columns: %s" % sorted(unknown))
    keep = []
    for k in CSV_COLUMNS:
        v = fields.get(k, "" if k in ("matrix", "exp_name") else 0)
        if k in ("matrix", "exp_name"):
            v = str(v).encode()
            keep.append(v)
        setattr(f, names.get(k, k), v)
    hb, vb = C.create_string_buffer(2048), C.create_string_buffer(4096)
    check(lib.sparta_blocking_csv_row(C.byref(f), hb, len(hb), vb, len(vb)))
    return hb.value.decode(), vb.value.decode()


def save_blocking_data(outfile, engine, cmat, grouping, save_blocking=True, **fields):
    """save_blocking_data(outfile, cLine, bEngine, cmat, save_blocking, blocking_outfile) (utilities.cpp:175-245): appends the
    header + value line to `outfile` and writes the grouping to `outfile + ".g"`.  Blocking statistics come from
    `engine` (CollectBlockingInfo + the counters of the last GetGrouping); the command-line fields are keyword arguments."""
    if engine.grouping_result is None:
        engine.grouping_result = np.ascontiguousarray(grouping, np.int64)
    engine.CollectBlockingInfo(cmat)
    row = dict(rows=cmat.rows, cols=cmat.cols, nonzeros=cmat.nztot(), blocking_algo=int(engine.blocking_algo), tau=float(engine.tau),
               row_block_size=int(engine.row_block_size), col_block_size=int(engine.col_block_size),
               use_pattern=int(bool(engine.use_pattern)), sim_use_groups=int(bool(engine.use_groups)),
               sim_measure=int(engine.sim_measure),
               time_to_block=float(getattr(engine, "timer_total", 0.0)), time_to_merge=float(getattr(engine, "timer_merges", 0.0)),
               time_to_compare=float(getattr(engine, "timer_comparisons", 0.0)),
               VBR_nzcount=int(engine.VBR_nzcount), VBR_nzblocks_count=int(engine.VBR_nzblocks_count),
               VBR_average_height=float(engine.VBR_average_height), VBR_longest_row=int(engine.VBR_longest_row),
               merge_counter=int(getattr(engine, "merge_counter", 0)), comparison_counter=int(getattr(engine, "comparison_counter", 0)),
               average_merge_tau=float(getattr(engine, "average_merge_tau", 0.0)),
               average_row_distance=float(getattr(engine, "average_row_distance", 0.0)),
               avg_time_multiply=float(getattr(engine, "multiplication_timer_avg", 0.0)),
               std_time_multiply=float(getattr(engine, "multiplication_timer_std", 0.0)))
    row.update(fields)
    header, values = blocking_csv_row(**row)
    with open(outfile, "a") as f:
        f.write(header + "\n" + values + "\n")
    if save_blocking:
        save_grouping(str(outfile) + ".g", grouping)
    return header, values
