"""sparta_amd -- MI355X-native block-sparse SpMM path with SPARTA's reorder-and-multiply API.

Host-side mirror of the reference's operator interface for the ONE hot path
(Jaccard row-clustering reorder -> VBS build -> VBS A x dense B):

    reference (C++)                                   here
    ----------------------------------------------    ------------------------------------------
    struct CSR            include/matrices.h:10       sparta_amd.CSR
    class  BlockingEngine include/blocking.h:9        sparta_amd.BlockingEngine  (.GetGrouping, .CollectBlockingInfo)
    struct VBR            include/matrices.h:93       sparta_amd.VBR             (.fill_from_CSR_inplace, .multiply)
    cublas_*/cutlas_* multiply back-ends
                          include/cuda_utilities.h:38 sparta_amd.vbs_multiply / DeviceVBS.spmm

All arithmetic happens in libsparta_amd.so (host C++ + HIP kernels for gfx950) through the C-ABI in
include/sparta_amd.h; this package is a thin ctypes layer and holds no algorithmic fallback.
"""
from ._lib import (SpartaError, LIB_PATH, F32, F16, BF16, COL_MAJOR, ROW_MAJOR, SPMM_MFMA, SPMM_EXACT,  # noqa: F401
                   FMT_EL, FMT_MTX, IO_COMPAT, IO_STRICT)
from .host import (CSR, BlockingEngine, VBR, get_permutation, get_partition, get_fixed_size_grouping,  # noqa: F401
                   row_distance, merge_rows, BLOCKING_ALGOS, save_grouping, read_grouping_file, blocking_csv_row,
                   save_blocking_data, CSV_COLUMNS)
from .device import DeviceVBS, vbs_multiply, device_count  # noqa: F401
from . import gen, dist  # noqa: F401

# revision tag of the device kernels: PMC-derived numbers kept under profiles/ (HBM traffic per launch) carry it, and bench.py
# only quotes them for the revision they were measured on
KERNEL_REV = "r5i"

# names of the reference's GPU back-ends this path replaces (include/cuda_utilities.h:38-44,
# include/cutlass_bellpack_lib.h:19-25): all map onto the single fused kernel family.
cublas_fixed_blocks_multiply = vbs_multiply
cublas_blockmat_batched = vbs_multiply
cutlas_fixed_blocks_multiply = vbs_multiply
cutlas_blockmat_batched = vbs_multiply
