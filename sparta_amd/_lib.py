"""ctypes binding of libsparta_amd.so (the C-ABI declared in include/sparta_amd.h).

The shared library is built in-tree by `make -C sparta_amd/csrc` (see __graft_entry__.build()).
There is NO Python/CPU fallback: if the library is missing, importing this module raises.
"""
import ctypes as C
import os

try:
    # PyTorch-ROCm wheels bundle their own libamdhip64; a process must end up with ONE HIP runtime.  Loading torch
    # first makes libsparta_amd.so (NEEDED libamdhip64.so.7) bind to the runtime torch uses, so torch tensors,
    # torch streams and our kernels share one device context.  Without torch the system ROCm runtime is used.
    import torch  # noqa: F401
except ImportError:  # pragma: no cover - torch is optional for the pure C-ABI user
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPARTA_AMD_LIB") or os.path.join(_HERE, "libsparta_amd.so")   # SPARTA_AMD_LIB: developer builds (make timeline)

# status codes (include/sparta_amd.h)
OK, ERR_INVALID, ERR_ALLOC, ERR_HIP, ERR_UNSUPPORTED, ERR_IO, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5, -6
F32, F16, BF16 = 0, 1, 2
COL_MAJOR, ROW_MAJOR = 0, 1
PTR_HOST, PTR_DEVICE = 0, 1
SPMM_MFMA, SPMM_EXACT = 0, 1
FMT_EL, FMT_MTX = 0, 1
IO_COMPAT, IO_STRICT = 0, 1

# every symbol include/sparta_amd.h declares (tests check the library exports all of them)
SYMBOLS = [
    "sparta_reorder_cfg_default", "sparta_reorder", "sparta_get_permutation", "sparta_get_partition",
    "sparta_get_fixed_size_grouping", "sparta_row_distance", "sparta_merge_rows", "sparta_vbs_build",
    "sparta_vbs_host_free", "sparta_blocking_info", "sparta_vbs_create", "sparta_vbs_create_range", "sparta_vbs_create_from_csr",
    "sparta_vbs_spmm", "sparta_vbs_spmm_gathered", "sparta_vbs_spmm_gathered_ld", "sparta_pack_blocks", "sparta_vbs_create_transposed", "sparta_vbs_spmm_ba", "sparta_vbs_set_class_timing",
    "sparta_vbs_class_times", "sparta_vbs_clock_mhz", "sparta_vbs_destroy", "sparta_vbs_info", "sparta_vbs_sparse_info", "sparta_vbs_colres_info", "sparta_colres_host_check", "sparta_vbs_union_info", "sparta_union_host_check", "sparta_vbs_hub_info", "sparta_device_count", "sparta_last_error",
    "sparta_version",
    "sparta_csr_read", "sparta_csr_read_buffer", "sparta_csr_host_free", "sparta_csr_write_edgelist", "sparta_grouping_write", "sparta_grouping_read",
    "sparta_blocking_csv_row", "sparta_degree_permutation", "sparta_vbs_save", "sparta_vbs_load", "sparta_vbs_to_blocked_ell",
    "sparta_vbs_build_partition", "sparta_vbs_partition_check", "sparta_vbs_plan_stats",
    "sparta_vbs_prepare_b", "sparta_vbs_spmm_prepared", "sparta_b_destroy",
]


class CsrHost(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("nnz", C.c_int64), ("pattern_only", C.c_int32),
                ("rowptr", C.POINTER(C.c_int64)), ("colidx", C.POINTER(C.c_int32)), ("vals", C.POINTER(C.c_float))]


class CsvFields(C.Structure):
    _fields_ = [("matrix", C.c_char_p), ("rows", C.c_int64), ("cols", C.c_int64), ("nonzeros", C.c_int64),
                ("symmetrize", C.c_int32), ("blocking_algo", C.c_int32), ("tau", C.c_float),
                ("row_block_size", C.c_int32), ("col_block_size", C.c_int32), ("use_pattern", C.c_int32),
                ("sim_use_groups", C.c_int32), ("sim_measure", C.c_int32), ("reorder", C.c_int32),
                ("exp_name", C.c_char_p), ("b_cols", C.c_int32), ("warmup", C.c_int32), ("exp_repetitions", C.c_int32),
                ("multiplication_algo", C.c_int32), ("n_streams", C.c_int32),
                ("time_to_block", C.c_float), ("time_to_merge", C.c_float), ("time_to_compare", C.c_float),
                ("vbr_nzcount", C.c_int64), ("vbr_nzblocks_count", C.c_int64), ("vbr_average_height", C.c_float),
                ("vbr_longest_row", C.c_int64), ("merge_counter", C.c_int64), ("comparison_counter", C.c_int64),
                ("average_merge_tau", C.c_float), ("average_row_distance", C.c_float),
                ("avg_time_multiply", C.c_float), ("std_time_multiply", C.c_float)]


class ReorderCfg(C.Structure):
    _fields_ = [("blocking_algo", C.c_int32), ("sim_measure", C.c_int32), ("tau", C.c_float), ("use_groups", C.c_int32),
                ("col_block_size", C.c_int64), ("row_block_size", C.c_int64), ("use_pattern", C.c_int32),
                ("force_fixed_size", C.c_int32), ("structured_m", C.c_int32), ("structured_n", C.c_int32),
                ("minhash_bands", C.c_int32), ("minhash_rows", C.c_int32), ("minhash_max_eval", C.c_int32), ("minhash_max_rows", C.c_int32)]


class ReorderStats(C.Structure):
    _fields_ = [("comparison_counter", C.c_int64), ("merge_counter", C.c_int64), ("average_row_distance", C.c_float),
                ("average_merge_tau", C.c_float), ("timer_total", C.c_float), ("timer_comparisons", C.c_float),
                ("timer_merges", C.c_float), ("reserved", C.c_int32)]


class VbsHost(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("block_rows", C.c_int64), ("block_cols", C.c_int64),
                ("block_col_size", C.c_int64), ("nztot", C.c_int64), ("nblocks", C.c_int64),
                ("row_part", C.POINTER(C.c_int64)), ("nzcount", C.POINTER(C.c_int64)), ("jab", C.POINTER(C.c_int64)),
                ("mab", C.POINTER(C.c_float))]


class SpartaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sparta_amd error %d: %s" % (code, msg))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("sparta_amd: %s is missing -- build it with `make -C sparta_amd/csrc` "
                          "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    i64p, i32p, f32p, vp = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_void_p
    L.sparta_reorder_cfg_default.argtypes = [C.POINTER(ReorderCfg)]
    L.sparta_reorder_cfg_default.restype = None
    L.sparta_reorder.argtypes = [C.c_int64, C.c_int64, i64p, i32p, C.POINTER(ReorderCfg), i64p, C.POINTER(ReorderStats)]
    L.sparta_get_permutation.argtypes = [i64p, C.c_int64, i64p]
    L.sparta_get_partition.argtypes = [i64p, C.c_int64, i64p, i64p]
    L.sparta_get_fixed_size_grouping.argtypes = [i64p, C.c_int64, C.c_int64, i64p]
    L.sparta_row_distance.argtypes = [C.c_int32, i64p, C.c_int64, C.c_int64, i64p, C.c_int64, C.c_int64, C.c_int64, f32p]
    L.sparta_merge_rows.argtypes = [i64p, C.c_int64, i64p, C.c_int64, i64p, i64p]
    L.sparta_vbs_build.argtypes = [C.c_int64, C.c_int64, i64p, i32p, f32p, i64p, C.c_int64, C.c_int64, C.c_int32,
                                   C.POINTER(VbsHost)]
    L.sparta_vbs_host_free.argtypes = [C.POINTER(VbsHost)]
    L.sparta_vbs_build_partition.argtypes = [C.c_int64, C.c_int64, i64p, i32p, f32p, i64p, C.c_int64, C.c_int64, C.POINTER(VbsHost)]
    L.sparta_vbs_partition_check.argtypes = [i64p, C.c_int64, C.c_int64]
    L.sparta_vbs_save.argtypes = [C.c_char_p, C.POINTER(VbsHost)]
    L.sparta_vbs_load.argtypes = [C.c_char_p, C.POINTER(VbsHost)]
    L.sparta_vbs_to_blocked_ell.argtypes = [C.POINTER(VbsHost), i64p, i64p, f32p]
    L.sparta_vbs_host_free.restype = None
    L.sparta_blocking_info.argtypes = [C.c_int64, C.c_int64, i64p, i32p, i64p, C.c_int64, i64p, f32p]
    L.sparta_vbs_create.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64, C.c_int64, i64p, i64p, i64p, f32p,
                                    C.c_int32, C.c_int32]
    L.sparta_vbs_create_range.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64, C.c_int64, i64p, i64p, i64p, f32p,
                                          C.c_int64, C.c_int64, C.c_int32, C.c_int32]
    L.sparta_vbs_create_from_csr.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, i64p, i32p, f32p, i64p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
    L.sparta_vbs_plan_stats.argtypes = [C.c_int64, C.c_int64, i64p, i32p, f32p, i64p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, i64p]
    L.sparta_vbs_prepare_b.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, vp, C.POINTER(vp)]
    L.sparta_vbs_spmm_prepared.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp, f32p]
    L.sparta_b_destroy.argtypes = [vp]
    L.sparta_vbs_create_transposed.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64, C.c_int64, i64p, i64p, i64p, f32p, C.c_int32, C.c_int32]
    L.sparta_vbs_spmm_ba.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, C.c_int64, C.c_int32, C.c_int32, vp, f32p]
    L.sparta_vbs_spmm.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp,
                                  C.c_int32, f32p]
    L.sparta_vbs_spmm_gathered.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int32, vp, C.c_int64, C.c_int32, C.c_int32, vp, C.c_int32, f32p]
    L.sparta_vbs_spmm_gathered_ld.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, vp, C.c_int64, C.c_int32, C.c_int32, vp, C.c_int32, f32p]
    L.sparta_pack_blocks.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, vp]
    L.sparta_vbs_set_class_timing.argtypes = [vp, C.c_int32]
    L.sparta_vbs_class_times.argtypes = [vp, f32p]
    L.sparta_vbs_clock_mhz.argtypes = [vp, C.POINTER(C.c_double)]
    L.sparta_vbs_destroy.argtypes = [vp]
    L.sparta_csr_read.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CsrHost)]
    L.sparta_csr_read_buffer.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CsrHost)]
    L.sparta_csr_host_free.argtypes = [C.POINTER(CsrHost)]
    L.sparta_csr_host_free.restype = None
    L.sparta_csr_write_edgelist.argtypes = [C.c_char_p, C.c_int64, i64p, i32p, C.c_char_p, C.c_int32]
    L.sparta_grouping_write.argtypes = [C.c_char_p, i64p, C.c_int64]
    L.sparta_grouping_read.argtypes = [C.c_char_p, C.c_int64, i64p, C.c_int64, C.POINTER(C.c_int64)]
    L.sparta_blocking_csv_row.argtypes = [C.POINTER(CsvFields), C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
    L.sparta_degree_permutation.argtypes = [C.c_int64, i64p, C.c_int32, i64p]
    L.sparta_vbs_info.argtypes = [vp, i64p]
    L.sparta_vbs_sparse_info.argtypes = [vp, i64p]
    L.sparta_vbs_hub_info.argtypes = [vp, i64p]
    L.sparta_vbs_colres_info.argtypes = [vp, i64p]
    L.sparta_colres_host_check.argtypes = [C.c_int64, C.c_int64, i64p, i32p, f32p, i64p, f32p, f32p, i64p]
    L.sparta_vbs_union_info.argtypes = [vp, i64p]
    L.sparta_union_host_check.argtypes = [C.c_int64, C.c_int64, i64p, i32p, f32p, i64p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, f32p, C.POINTER(C.c_double), i64p]
    L.sparta_device_count.argtypes = []
    L.sparta_last_error.restype = C.c_char_p
    L.sparta_version.restype = C.c_char_p
    return L


lib = _load()


def check(rc):
    if rc != OK:
        raise SpartaError(rc, lib.sparta_last_error().decode(errors="replace"))
