"""ctypes binding of oracle/_ref/libsparta_ref.so -- the REAL reference (HicrestLaboratory/SPARTA,
src/general/*.cpp compiled unmodified by oracle/Makefile) behind our own C-ABI window
oracle/ref_shim.cpp.

TEST INFRASTRUCTURE ONLY: imported by tests/, tests/golden/make_golden.py, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  Never imported by the product package `sparta_amd`.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libsparta_ref.so")

# BlockingType enum of the reference (include/definitions.h:17)
ALGO = dict(iterative=0, iterative_structured=1, fixed_size=2, iterative_clocked=3,
            iterative_queue=4, iterative_max_size=5, scramble=6)


def available():
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libsparta_ref.so missing: run `make -C oracle ref` where /root/reference exists")
        L = C.CDLL(_PATH)
        vp, lp, fp = C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_float)
        L.ref_csr_create.restype = vp
        L.ref_csr_create.argtypes = [C.c_long, C.c_long, lp, lp, fp]
        L.ref_csr_read.restype = vp
        L.ref_csr_read.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.ref_csr_destroy.argtypes = [vp]
        for n in ("ref_csr_rows", "ref_csr_cols", "ref_csr_nnz"):
            getattr(L, n).restype = C.c_long
            getattr(L, n).argtypes = [vp]
        L.ref_csr_pattern_only.argtypes = [vp]
        L.ref_csr_export.argtypes = [vp, lp, lp, fp]
        L.ref_csr_multiply.argtypes = [vp, fp, C.c_long, fp]
        L.ref_set_structured.argtypes = [C.c_int, C.c_int]
        L.ref_csr_reorder.argtypes = [vp, lp, C.c_long]
        L.ref_csr_reorder_by_degree.argtypes = [vp, C.c_int]
        L.ref_csr_save_to_edgelist.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        L.ref_save_blocking_data.argtypes = [vp, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_float, fp, fp, C.c_char_p, C.c_long,
                                             C.c_char_p, C.c_long]
        L.ref_get_grouping.argtypes = [vp, C.c_int, C.c_float, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                       lp, lp, fp, lp, fp]
        L.ref_get_permutation.argtypes = [lp, C.c_long, lp]
        L.ref_get_partition.restype = C.c_long
        L.ref_get_partition.argtypes = [lp, C.c_long, lp]
        L.ref_get_fixed_size_grouping.argtypes = [lp, C.c_long, C.c_long, lp]
        L.ref_merge_rows.restype = C.c_long
        L.ref_merge_rows.argtypes = [lp, C.c_long, lp, C.c_long, lp]
        L.ref_distance.restype = C.c_float
        L.ref_distance.argtypes = [C.c_int, lp, C.c_long, C.c_long, lp, C.c_long, C.c_long, C.c_long]
        L.ref_vbr_create.restype = vp
        L.ref_vbr_create.argtypes = [vp, lp, C.c_long, C.c_long, C.c_long, C.c_int]
        L.ref_vbr_destroy.argtypes = [vp]
        L.ref_vbr_create_partition.restype = vp
        L.ref_vbr_create_partition.argtypes = [vp, lp, C.c_long, C.c_long]
        L.ref_vbr_block_start.restype = C.c_long
        L.ref_vbr_block_start.argtypes = [vp, C.c_long]
        L.ref_vbr_partition_check.argtypes = [vp, lp, C.c_long]
        L.ref_vbr_dims.argtypes = [vp, lp]
        L.ref_vbr_export.argtypes = [vp, lp, lp, lp, fp]
        L.ref_vbr_multiply.argtypes = [vp, fp, C.c_int, fp]
        _lib = L
    return _lib


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class RefCSR:
    """Owns a reference `CSR` (include/matrices.h:10-91)."""

    def __init__(self, rows=None, cols=None, rowptr=None, colidx=None, vals=None, handle=None):
        L = lib()
        if handle is not None:
            self.h = handle
        else:
            rp, ci = _i64(rowptr), _i64(colidx)
            v = None if vals is None else np.ascontiguousarray(vals, dtype=np.float32)
            self.h = L.ref_csr_create(rows, cols, _lp(rp), _lp(ci), _fp(v))
        self.rows = L.ref_csr_rows(self.h)
        self.cols = L.ref_csr_cols(self.h)
        self.nnz = L.ref_csr_nnz(self.h)
        self.pattern_only = bool(L.ref_csr_pattern_only(self.h))

    @classmethod
    def read(cls, path, delim=" ", pattern_only=False, mat_fmt=0, symmetrize=False):
        h = lib().ref_csr_read(path.encode(), delim.encode(), int(pattern_only), int(mat_fmt), int(symmetrize))
        if not h:
            raise ValueError("reference reader failed on %s" % path)
        return cls(handle=h)

    def export(self):
        rp = np.zeros(self.rows + 1, np.int64)
        ci = np.zeros(self.nnz, np.int64)
        v = np.zeros(self.nnz, np.float32)
        lib().ref_csr_export(self.h, _lp(rp), _lp(ci), _fp(v))
        return rp, ci, v

    def multiply(self, B, n_cols):
        """reference CSR::multiply; B column-major with ld = rows (csr.cpp:61 -> square A only)."""
        B = np.ascontiguousarray(B, np.float32)
        Cm = np.zeros(self.rows * n_cols, np.float32)
        lib().ref_csr_multiply(self.h, _fp(B), n_cols, _fp(Cm))
        return Cm

    def grouping(self, algo=3, tau=0.5, col_block_size=1, row_block_size=1, use_groups=False,
                 use_pattern=True, force_fixed_size=False, sim_measure=1, with_info=False, structured_m=2, structured_n=4):
        lib().ref_set_structured(int(structured_m), int(structured_n))
        g = np.zeros(self.rows, np.int64)
        st = np.zeros(2, np.int64)
        fst = np.zeros(2, np.float32)
        info = np.zeros(3, np.int64)
        finfo = np.zeros(1, np.float32)
        lib().ref_get_grouping(self.h, algo, tau, col_block_size, row_block_size, int(use_groups), int(use_pattern),
                               int(force_fixed_size), sim_measure, _lp(g), _lp(st), _fp(fst),
                               _lp(info) if with_info else None, _fp(finfo) if with_info else None)
        stats = dict(comparison_counter=int(st[0]), merge_counter=int(st[1]),
                     average_row_distance=float(fst[0]), average_merge_tau=float(fst[1]))
        if with_info:
            stats.update(VBR_nzcount=int(info[0]), VBR_nzblocks_count=int(info[1]), VBR_longest_row=int(info[2]),
                         VBR_average_height=float(finfo[0]))
        return g, stats

    def reorder(self, grouping):
        g = _i64(grouping)
        lib().ref_csr_reorder(self.h, _lp(g), len(g))

    def reorder_by_degree(self, descending=True):
        lib().ref_csr_reorder_by_degree(self.h, int(bool(descending)))

    def save_to_edgelist(self, path, delim=" ", pattern_only=False, mat_fmt=0):
        if lib().ref_csr_save_to_edgelist(self.h, str(path).encode(), delim.encode(), int(pattern_only), int(mat_fmt)) != 0:
            raise OSError("cannot write %s" % path)

    CSV_INTS = ("symmetrize", "blocking_algo", "row_block_size", "col_block_size", "use_pattern", "sim_use_groups", "sim_measure",
                "reorder", "b_cols", "warmup", "exp_repetitions", "multiplication_algo", "n_streams", "force_fixed_size")

    def save_blocking_data(self, filename="m.el", exp_name="", tau=0.1, timers=None, mult=None, **ints):
        """the reference's save_blocking_data (utilities.cpp:175-245) after a GetGrouping with these settings:
        returns (csv text = header line + value line, grouping-file text)"""
        defaults = dict(symmetrize=0, blocking_algo=3, row_block_size=3, col_block_size=3, use_pattern=1, sim_use_groups=0,
                        sim_measure=1, reorder=0, b_cols=1024, warmup=1, exp_repetitions=5, multiplication_algo=0, n_streams=4,
                        force_fixed_size=0)
        defaults.update(ints)
        arr = (C.c_int * len(self.CSV_INTS))(*[int(defaults[k]) for k in self.CSV_INTS])
        t = None if timers is None else np.ascontiguousarray(timers, np.float32)
        m = None if mult is None else np.ascontiguousarray(mult, np.float32)
        csv = C.create_string_buffer(1 << 14)
        g = C.create_string_buffer(32 * self.rows + 64)
        rc = lib().ref_save_blocking_data(self.h, filename.encode(), exp_name.encode(), arr, float(tau), _fp(t), _fp(m), csv, len(csv), g, len(g))
        if rc != 0:
            raise RuntimeError("ref_save_blocking_data failed")
        return csv.value.decode(), g.value.decode()

    def __del__(self):
        try:
            if self.h:
                lib().ref_csr_destroy(self.h)
                self.h = None
        except Exception:
            pass


class RefVBR:
    """Owns a reference `VBR` (include/matrices.h:93-122) built by fill_from_CSR_inplace."""

    def __init__(self, csr, grouping, col_block_size, row_block_size=0, force_fixed_size=False, row_partition=None):
        if row_partition is not None:          # VBR::fill_from_CSR (vbr.cpp:239-321): no permutation, block-rows from a partition
            rp = _i64(row_partition)
            self.h = lib().ref_vbr_create_partition(csr.h, _lp(rp), len(rp), col_block_size)
        else:
            g = _i64(grouping)
            self.h = lib().ref_vbr_create(csr.h, _lp(g), len(g), col_block_size, row_block_size, int(force_fixed_size))
        d = np.zeros(7, np.int64)
        lib().ref_vbr_dims(self.h, _lp(d))
        (self.rows, self.cols, self.block_rows, self.block_cols, self.block_col_size, self.nztot, self.nblocks) = map(int, d)

    def export(self, with_mab=True):
        rp = np.zeros(self.block_rows + 1, np.int64)
        nz = np.zeros(self.block_rows, np.int64)
        jab = np.zeros(self.nblocks, np.int64)
        mab = np.zeros(self.nztot, np.float32) if with_mab else None
        lib().ref_vbr_export(self.h, _lp(rp), _lp(nz), _lp(jab), _fp(mab))
        return rp, nz, jab, mab

    def block_start(self, row_block_idx):
        """VBR::get_block_start (vbr.cpp:33-49) as an element offset into mab"""
        return int(lib().ref_vbr_block_start(self.h, int(row_block_idx)))

    def partition_check(self, part):
        """VBR::partition_check (vbr.cpp:108-118)"""
        p = _i64(part)
        return int(lib().ref_vbr_partition_check(self.h, _lp(p), len(p)))

    def multiply(self, B, n_cols, Cin=None):
        """reference VBR::multiply: C += A*B, B col-major ld=cols, C col-major ld=rows (vbr.cpp:323-372)."""
        # VBR::multiply reads B past its end when cols % w != 0 (vbr.cpp:351,362: the zero-padded last block column is multiplied with
        # whatever follows the last column of B in memory -- 0 * garbage, NaN if the garbage is inf / NaN).  The wrapper hands it a copy
        # followed by zeros so that the read is defined (0 * 0): the reference's result on DEFINED inputs, which is what parity means.
        Bp = np.zeros(np.asarray(B).size + 4096, np.float32)
        Bp[:np.asarray(B).size] = np.asarray(B, np.float32).reshape(-1)
        B = Bp
        Cm = np.zeros(self.rows * n_cols, np.float32) if Cin is None else np.ascontiguousarray(Cin, np.float32).copy()
        lib().ref_vbr_multiply(self.h, _fp(B), n_cols, _fp(Cm))
        return Cm

    def __del__(self):
        try:
            if self.h:
                lib().ref_vbr_destroy(self.h)
                self.h = None
        except Exception:
            pass


def get_permutation(grouping):
    g = _i64(grouping)
    out = np.zeros(len(g), np.int64)
    lib().ref_get_permutation(_lp(g), len(g), _lp(out))
    return out


def get_partition(grouping):
    g = _i64(grouping)
    out = np.zeros(len(g) + 2, np.int64)
    n = lib().ref_get_partition(_lp(g), len(g), _lp(out))
    return out[:n].copy()


def get_fixed_size_grouping(grouping, row_block_size):
    g = _i64(grouping)
    out = np.zeros(len(g), np.int64)
    lib().ref_get_fixed_size_grouping(_lp(g), len(g), row_block_size, _lp(out))
    return out


def merge_rows(A, B):
    A, B = _i64(A), _i64(B)
    out = np.zeros(len(A) + len(B) + 1, np.int64)
    n = lib().ref_merge_rows(_lp(A), len(A), _lp(B), len(B), _lp(out))
    return out[:n].copy()


def distance(which, A, gA, B, gB, block_size):
    """which: 0 Hamming, 1 Jaccard, 2/3 the 'OPENMP' twins (blocking.cpp:720-994)."""
    A, B = _i64(A), _i64(B)
    return float(lib().ref_distance(which, _lp(A), len(A), gA, _lp(B), len(B), gB, block_size))
