"""Device side: VBS image on one MI355X and the fused SpMM (sparta_vbs_create / sparta_vbs_spmm)."""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import lib, check

_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)


def device_count():
    return int(lib.sparta_device_count())


class DeviceVBS:
    """Opaque device image of a VBS matrix + its tile plan (sparta_vbs_t)."""

    def __init__(self, vbmat, device=0, dtype=_lib.F32, block_row_range=None):
        self.device = int(device)
        self.dtype = dtype
        self.h = C.c_void_p(None)
        if vbmat is None:                      # from_csr fills the handle
            return
        rp = np.ascontiguousarray(vbmat.row_part, np.int64)
        nz = np.ascontiguousarray(vbmat.nzcount, np.int64)
        jab = np.ascontiguousarray(vbmat.jab, np.int64)
        mab = np.ascontiguousarray(vbmat.mab, np.float32)
        b0, b1 = (0, vbmat.block_rows) if block_row_range is None else block_row_range
        check(lib.sparta_vbs_create_range(C.byref(self.h), vbmat.rows, vbmat.cols, vbmat.block_rows, vbmat.block_col_size,
                                          rp.ctypes.data_as(_i64p), nz.ctypes.data_as(_i64p), jab.ctypes.data_as(_i64p),
                                          mab.ctypes.data_as(_f32p), int(b0), int(b1), int(dtype), self.device))
        info = self.info()
        self.rows, self.cols = info["rows"], info["cols"]

    @classmethod
    def from_csr(cls, cmat, grouping, col_block_size, row_block_size=0, force_fixed_size=False, device=0, dtype=_lib.F32):
        """sparta_vbs_create_from_csr: build + upload in one step without expanding the nearly empty block-rows into dense blocks
        (what makes 10^8-nonzero power-law matrices fit); same product as VBR().fill_from_CSR_inplace(...).to_device()."""
        self = cls(None, device=device, dtype=dtype)
        g = np.ascontiguousarray(grouping, np.int64)
        if g.shape != (cmat.rows,):
            raise ValueError("grouping must have one entry per row")
        rp = np.ascontiguousarray(cmat.rowptr, np.int64)
        ci = np.ascontiguousarray(cmat.colidx, np.int32)
        vals = None if cmat.vals is None else np.ascontiguousarray(cmat.vals, np.float32)
        check(lib.sparta_vbs_create_from_csr(C.byref(self.h), cmat.rows, cmat.cols, rp.ctypes.data_as(_i64p), ci.ctypes.data_as(C.POINTER(C.c_int32)),
                                             None if vals is None else vals.ctypes.data_as(_f32p), g.ctypes.data_as(_i64p), int(col_block_size),
                                             int(row_block_size), int(bool(force_fixed_size)), int(dtype), self.device))
        info = self.info()
        self.rows, self.cols = info["rows"], info["cols"]
        return self

    @staticmethod
    def plan_stats(cmat, grouping, col_block_size, row_block_size=0, force_fixed_size=False, dtype=_lib.F32):
        """sparta_vbs_plan_stats: what from_csr WOULD build for this grouping (no GPU, nothing built): tiles, their area and MFMA
        steps, and what is left to the sparse-row kernels."""
        g = np.ascontiguousarray(grouping, np.int64)
        if g.shape != (cmat.rows,):
            raise ValueError("grouping must have one entry per row")
        rp = np.ascontiguousarray(cmat.rowptr, np.int64)
        ci = np.ascontiguousarray(cmat.colidx, np.int32)
        vals = None if cmat.vals is None else np.ascontiguousarray(cmat.vals, np.float32)
        st = np.zeros(8, np.int64)
        check(lib.sparta_vbs_plan_stats(cmat.rows, cmat.cols, rp.ctypes.data_as(_i64p), ci.ctypes.data_as(C.POINTER(C.c_int32)),
                                        None if vals is None else vals.ctypes.data_as(_f32p), g.ctypes.data_as(_i64p), int(col_block_size),
                                        int(row_block_size), int(bool(force_fixed_size)), int(dtype), st.ctypes.data_as(_i64p)))
        keys = ["tile_blocks", "tile_area", "mfma_steps", "sparse_nnz", "sparse_rows", "block_rows", "rows", "union_steps"]
        return {k: int(st[i]) for i, k in enumerate(keys)}

    @staticmethod
    def predict_cost(cmat, grouping, col_block_size, row_block_size=0, force_fixed_size=False, dtype=_lib.F32, n_cols=128):
        """Predicted time of one product of the handle from_csr would build (plan_stats x measured rates on one MI355X): the dense tiles at
        the executed rate of the stream kernels on power-law hubs (16-bit: 220 TFLOP/s = 13.6 ms for 5.8e9 stored elements x 256 columns;
        fp32: 100), the sparse rows at the gather rate (one n_cols-wide row of B per nonzero: 5.9 TB/s when every gather goes to HBM, 9.5 TB/s
        where the library takes the long rows column window by column window -- from 262144 columns and 4 M nonzeros, vbs_capi.cpp)
        + their rows of C.  Good to ~10 % on R-MAT parts (profiles/r3): enough to choose between two blockings of one matrix, which is all it is for."""
        st = DeviceVBS.plan_stats(cmat, grouping, col_block_size, row_block_size, force_fixed_size, dtype)
        esz = 4.0 if dtype == _lib.F32 else 2.0
        t_tiles = 2.0 * st["tile_area"] * n_cols / ((100e12 if dtype == _lib.F32 else 220e12))
        rate = 9.5e12 if (cmat.cols >= 4 * 65536 and st["sparse_nnz"] >= (4 << 20)) else 5.9e12
        t_sparse = st["sparse_nnz"] * (n_cols * esz + 8.0) / rate + st["sparse_rows"] * n_cols * 4.0 / 5.9e12
        st["ms_tiles"], st["ms_sparse"] = t_tiles * 1e3, t_sparse * 1e3
        st["ms"] = (t_tiles + t_sparse) * 1e3
        return st

    @classmethod
    def transposed_of(cls, vbmat, device=0, dtype=_lib.F32):
        """sparta_vbs_create_transposed: the handle of A^T, for the B * A product (spmm_BA)"""
        self = cls(None, device=device, dtype=dtype)
        rp = np.ascontiguousarray(vbmat.row_part, np.int64)
        nz = np.ascontiguousarray(vbmat.nzcount, np.int64)
        jab = np.ascontiguousarray(vbmat.jab, np.int64)
        mab = np.ascontiguousarray(vbmat.mab, np.float32)
        check(lib.sparta_vbs_create_transposed(C.byref(self.h), vbmat.rows, vbmat.cols, vbmat.block_rows, vbmat.block_col_size,
                                               rp.ctypes.data_as(_i64p), nz.ctypes.data_as(_i64p), jab.ctypes.data_as(_i64p),
                                               mab.ctypes.data_as(_f32p), int(dtype), self.device))
        info = self.info()
        self.rows, self.cols = info["rows"], info["cols"]          # rows = cols(A), cols = rows(A)
        return self

    def spmm_BA_host(self, B, M, C_out, accumulate=True):
        """C (+)= B * A on the handle of A^T, host buffers: B is M x rows(A), C is M x cols(A), both column-major (ld = M).  Returns kernel ms."""
        B = np.ascontiguousarray(B, np.float32).reshape(-1)
        if not (isinstance(C_out, np.ndarray) and C_out.dtype == np.float32 and C_out.flags.c_contiguous):
            raise ValueError("C must be a contiguous float32 numpy array (it is written in place)")
        if B.size < M * self.cols or C_out.size < M * self.rows:
            raise ValueError("B or C too small")
        dt = C.c_float(0)
        check(lib.sparta_vbs_spmm_ba(self.h, B.ctypes.data_as(C.c_void_p), int(M), int(M), C_out.ctypes.data_as(C.c_void_p), int(M),
                                     int(bool(accumulate)), _lib.PTR_HOST, None, C.byref(dt)))
        return dt.value

    def info(self):
        a = np.zeros(16, np.int64)
        check(lib.sparta_vbs_info(self.h, a.ctypes.data_as(_i64p)))
        keys = ["rows", "cols", "block_rows", "block_col_size", "nblocks", "nztot", "tiles16", "tiles32", "tiles64",
                "sparse_rows", "a_bytes", "exec_area", "stream_steps", "stream_workers", "split_tiles", "last_path"]
        return {k: int(a[i]) for i, k in enumerate(keys)}

    def sparse_info(self):
        """the part of the matrix on the sparse-row path: rows, nonzeros, rows handled by one wave, hub rows"""
        a = np.zeros(4, np.int64)
        check(lib.sparta_vbs_sparse_info(self.h, a.ctypes.data_as(_i64p)))
        return {"rows": int(a[0]), "nnz": int(a[1]), "short_rows": int(a[2]), "hub_rows": int(a[3])}

    def colres_info(self):
        """the resident-column image of a small fp32 handle (k_colres.hip): see sparta_vbs_colres_info; "nc" = columns per workgroup of the last product on that path"""
        a = np.zeros(12, np.int64)
        check(lib.sparta_vbs_colres_info(self.h, a.ctypes.data_as(_i64p)))
        return {"slices": int(a[0]), "entries": int(a[1]), "long_rows": int(a[2]), "plane": int(a[3]), "lmax": int(a[4]), "nc": int(a[5]), "nnz": int(a[6]), "unit": int(a[7]),
                "parts": int(a[8]), "ranges": int(a[9]), "small_parts": int(a[10]), "used_small": int(a[11])}

    def union_info(self):
        """the column-compacted ("union-pattern") tiles of a handle made from a CSR (k_union.hip): see sparta_vbs_union_info"""
        a = np.zeros(12, np.int64)
        check(lib.sparta_vbs_union_info(self.h, a.ctypes.data_as(_i64p)))
        keys = ["tiles32", "tiles64", "steps32", "steps64", "area", "list_entries", "nnz", "workers", "rows", "tail_nnz", "exec_area", "row_tile"]
        return {k: int(a[i]) for i, k in enumerate(keys)}

    def hub_info(self):
        """the hub part of a 16-bit plan (group tiles of long 64-row tiles for the GEMM-shaped kernel): see sparta_vbs_hub_info"""
        a = np.zeros(10, np.int64)
        check(lib.sparta_vbs_hub_info(self.h, a.ctypes.data_as(_i64p)))
        keys = ["steps", "tiles", "groups", "tiles_per_group", "stored_area", "union_area", "workers", "chunks", "segments"]
        return {k: int(a[i]) for i, k in enumerate(keys)}

    def spmm_host(self, B, n_cols, C_out, accumulate=True, algo=_lib.SPMM_MFMA, b_layout=_lib.COL_MAJOR,
                  c_layout=_lib.COL_MAJOR):
        """Host buffers in, host buffers out (the reference back-ends' contract). Returns kernel ms."""
        B = np.ascontiguousarray(B, np.float32).reshape(-1)
        if not (isinstance(C_out, np.ndarray) and C_out.dtype == np.float32 and C_out.flags.c_contiguous):
            raise ValueError("C must be a contiguous float32 numpy array (it is written in place)")
        ldb = self.cols if b_layout == _lib.COL_MAJOR else n_cols
        ldc = self.rows if c_layout == _lib.COL_MAJOR else n_cols
        if B.size < self.cols * n_cols or C_out.size < self.rows * n_cols:
            raise ValueError("B or C too small")
        dt = C.c_float(0)
        check(lib.sparta_vbs_spmm(self.h, B.ctypes.data_as(C.c_void_p), ldb, b_layout, int(n_cols),
                                  C_out.ctypes.data_as(C.c_void_p), ldc, c_layout, int(bool(accumulate)), _lib.PTR_HOST, None,
                                  int(algo), C.byref(dt)))
        return dt.value

    def spmm(self, B, C_out, n_cols, accumulate=False, algo=_lib.SPMM_MFMA, b_layout=_lib.COL_MAJOR, c_layout=_lib.COL_MAJOR,
             ldb=None, ldc=None, timed=False, stream=None):
        """Device tensors (torch, on this device): stream-ordered on torch's current stream, no copies.
        B: cols x n_cols, C: rows x n_cols in the given layouts. Returns kernel ms if timed else None.
        Handles created with dtype F16 / BF16 take B in that type (column-major, even ldb); C is always float32."""
        import torch
        want_b = {_lib.F32: torch.float32, _lib.F16: torch.float16, _lib.BF16: torch.bfloat16}[self.dtype]
        if not (B.is_cuda and C_out.is_cuda and B.dtype == want_b and C_out.dtype == torch.float32):
            raise ValueError("B must be a %s tensor and C a float32 tensor, both on the GPU" % want_b)
        if B.device.index != self.device or C_out.device.index != self.device:
            raise ValueError("B and C must live on device %d" % self.device)
        ldb = (self.cols if b_layout == _lib.COL_MAJOR else n_cols) if ldb is None else ldb
        ldc = (self.rows if c_layout == _lib.COL_MAJOR else n_cols) if ldc is None else ldc
        need_b = ldb * (n_cols if b_layout == _lib.COL_MAJOR else self.cols)
        need_c = ldc * (n_cols if c_layout == _lib.COL_MAJOR else self.rows)
        if B.numel() < need_b or C_out.numel() < need_c or not B.is_contiguous() or not C_out.is_contiguous():
            raise ValueError("B or C too small / not contiguous for the stated leading dimensions")
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        dt = C.c_float(0)
        check(lib.sparta_vbs_spmm(self.h, C.c_void_p(B.data_ptr()), int(ldb), b_layout, int(n_cols), C.c_void_p(C_out.data_ptr()),
                                  int(ldc), c_layout, int(bool(accumulate)), _lib.PTR_DEVICE, C.c_void_p(st), int(algo),
                                  C.byref(dt) if timed else None))
        return dt.value if timed else None

    def spmm_gathered(self, B_gathered, shard_rows, C_out, n_cols, accumulate=False, algo=_lib.SPMM_MFMA,
                      c_layout=_lib.COL_MAJOR, shard_stride=None, timed=False, stream=None, shard_ld=None):
        """Multi-GPU entry: B_gathered is the all-gather result (n_shards column-major slabs of shard_rows x n_cols; shard_ld: elements between the columns of a
        slab, default shard_rows -- pad it when shard_rows is a multiple of a large power of two: sparta_vbs_spmm_gathered_ld)."""
        import torch
        want_b = {_lib.F32: torch.float32, _lib.F16: torch.float16, _lib.BF16: torch.bfloat16}[self.dtype]
        if not (B_gathered.is_cuda and C_out.is_cuda and B_gathered.dtype == want_b and C_out.dtype == torch.float32):
            raise ValueError("B_gathered must be a %s tensor and C a float32 tensor, both on the GPU" % want_b)
        if B_gathered.device.index != self.device or C_out.device.index != self.device:
            raise ValueError("B_gathered and C must live on device %d" % self.device)
        shard_ld = shard_rows if shard_ld is None else int(shard_ld)
        shard_stride = shard_ld * n_cols if shard_stride is None else shard_stride
        if B_gathered.numel() < (self.cols // shard_rows - 1) * shard_stride + shard_ld * (n_cols - 1) + shard_rows:
            raise ValueError("B_gathered too small")
        ldc = self.rows if c_layout == _lib.COL_MAJOR else n_cols
        need_c = ldc * (n_cols if c_layout == _lib.COL_MAJOR else self.rows)
        if C_out.numel() < need_c or not B_gathered.is_contiguous() or not C_out.is_contiguous():
            raise ValueError("C too small / B_gathered or C not contiguous")
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        dt = C.c_float(0)
        check(lib.sparta_vbs_spmm_gathered_ld(self.h, C.c_void_p(B_gathered.data_ptr()), int(shard_rows), int(shard_ld), int(shard_stride), int(n_cols),
                                              C.c_void_p(C_out.data_ptr()), int(ldc), c_layout, int(bool(accumulate)), C.c_void_p(st),
                                              int(algo), C.byref(dt) if timed else None))
        return dt.value if timed else None

    def prepare_b(self, B, n_cols, ldb=None, shard_rows=0, shard_stride=None, stream=None):
        """sparta_vbs_prepare_b: a B that stays the same over many products, prepared once (its row-major copy for the sparse-row kernels is
        made now instead of on every product).  Returns a PreparedB; B itself is referenced, not copied: keep it alive and unchanged."""
        import torch
        want_b = {_lib.F32: torch.float32, _lib.F16: torch.float16, _lib.BF16: torch.bfloat16}[self.dtype]
        if not (B.is_cuda and B.dtype == want_b and B.device.index == self.device and B.is_contiguous()):
            raise ValueError("B must be a contiguous %s tensor on device %d" % (want_b, self.device))
        if shard_rows:
            ldb = shard_rows if ldb is None else int(ldb)                      # gathered B: ldb = shard_ld, the column stride inside a slab (padded slabs: > shard_rows)
            shard_stride = ldb * n_cols if shard_stride is None else int(shard_stride)
            need = (self.cols // shard_rows - 1) * shard_stride + ldb * (n_cols - 1) + shard_rows
        else:
            ldb = self.cols if ldb is None else int(ldb)
            shard_stride, need = 0, ldb * n_cols
        if B.numel() < need:
            raise ValueError("B too small for the stated layout")
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        h = C.c_void_p(None)
        check(lib.sparta_vbs_prepare_b(self.h, C.c_void_p(B.data_ptr()), int(ldb), int(shard_rows), int(shard_stride), int(n_cols), C.c_void_p(st), C.byref(h)))
        return PreparedB(self, h, B, int(n_cols))

    def spmm_prepared(self, Bp, C_out, accumulate=False, c_layout=_lib.COL_MAJOR, ldc=None, timed=False, stream=None):
        """sparta_vbs_spmm_prepared: C (+)= A * B for a PreparedB of this handle"""
        import torch
        if Bp.owner is not self or not Bp.h:
            raise ValueError("this B was prepared for another handle (or already closed)")
        if not (C_out.is_cuda and C_out.dtype == torch.float32 and C_out.device.index == self.device and C_out.is_contiguous()):
            raise ValueError("C must be a contiguous float32 tensor on device %d" % self.device)
        ldc = (self.rows if c_layout == _lib.COL_MAJOR else Bp.n_cols) if ldc is None else ldc
        if C_out.numel() < ldc * (Bp.n_cols if c_layout == _lib.COL_MAJOR else self.rows):
            raise ValueError("C too small")
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        dt = C.c_float(0)
        check(lib.sparta_vbs_spmm_prepared(self.h, Bp.h, C.c_void_p(C_out.data_ptr()), int(ldc), c_layout, int(bool(accumulate)), C.c_void_p(st),
                                           C.byref(dt) if timed else None))
        return dt.value if timed else None

    def set_class_timing(self, enable=True):
        check(lib.sparta_vbs_set_class_timing(self.h, int(bool(enable))))

    def class_times(self):
        """ms of the last spmm per kernel (needs set_class_timing(True)).  Stream path (info()['last_path'] == 1):
        {'stream': main persistent kernel, 'fixup': split-tile fix-up}; generic path: per tile class {16, 32, 64}; 'sparse':
        the sparse-row kernels (+ the transposes of B / C they need), 0 when the handle has no such rows."""
        a = np.zeros(4, np.float32)
        check(lib.sparta_vbs_class_times(self.h, a.ctypes.data_as(_f32p)))
        if self.info()["last_path"] == 1:
            return {"stream": float(a[0]), "fixup": float(a[1]), "union": float(a[2]), "sparse": float(a[3])}      # 'union': the column-compacted tiles (k_union.hip; + the transpose of B when they ask for it first)
        if self.union_info()["area"] > 0 and self.info()["tiles64"] == 0:                     # (slot 2 is the column-compacted tiles' when no 64-row class ran)
            return {"class16": float(a[0]), "class32": float(a[1]), "union": float(a[2]), "sparse": float(a[3])}
        return {"class16": float(a[0]), "class32": float(a[1]), "class64": float(a[2]), "sparse": float(a[3])}

    def clock_mhz(self):
        """Shader clock (MHz) the product kernels of the last timed spmm ran at, keyed like class_times()
        (needs set_class_timing(True); 0.0 where no probe ran, e.g. the fix-up kernel)."""
        a = np.zeros(4, np.float64)
        check(lib.sparta_vbs_clock_mhz(self.h, a.ctypes.data_as(C.POINTER(C.c_double))))
        if self.info()["last_path"] == 1:
            return {"stream": float(a[0])}
        return {"class16": float(a[0]), "class32": float(a[1]), "class64": float(a[2])}

    def close(self):
        if self.h:
            lib.sparta_vbs_destroy(self.h)
            self.h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PreparedB:
    """sparta_b_t: a constant dense operand prepared for one handle (DeviceVBS.prepare_b)"""

    def __init__(self, owner, h, B, n_cols):
        self.owner, self.h, self.B, self.n_cols = owner, h, B, n_cols          # (keeps B alive)

    def close(self):
        if self.h:
            lib.sparta_b_destroy(self.h)
            self.h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vbs_multiply(vbmat, B, B_cols, C_out, n_streams=None, device=0):
    """Drop-in for the reference's GPU back-ends `f(const VBR&, DataT* B, int B_cols, DataT_C* C, float& dt[, n_streams])`
    (include/cuda_utilities.h:38-44): host B/C, C += A*B, returns dt in ms (kernel only). `n_streams` is accepted and
    ignored (there is one fused launch per tile class, not one GEMM per block)."""
    return vbmat.multiply(B, B_cols, C_out, device=device)
