/*
 * sparta_amd.h -- C-ABI of the MI355X-native block-sparse SpMM path
 * (Jaccard row-clustering reorder -> VBS build -> VBS A x dense B on gfx950).
 *
 * This is the drop-in boundary for the ONE hot path of HicrestLaboratory/SPARTA.  Every entry
 * point below cites the reference interface it replaces (paths relative to the reference tree).
 * Plain pointers and sizes only; no C++/torch types; no exceptions cross the boundary.
 * All functions return SPARTA_OK (0) or a negative status; sparta_last_error() returns a
 * thread-local message for the last failure on the calling thread.
 *
 * Conventions kept from the reference:
 *   - `intT` is 64-bit (include/definitions.h:4) -> every VBS index array is int64_t here.
 *   - A VBS block is column-major h x w, blocks of a block-row are consecutive, block-rows are
 *     consecutive (include/matrices.h:95-104).
 *   - B is column-major cols x N with ld = cols, C is column-major rows x N with ld = rows, the
 *     rows of C are in REORDERED order, and C is accumulated into (src/general/vbr.cpp:323-372).
 *     Both layouts and the accumulate flag are explicit arguments here.
 *   - `dt` is the device time of the multiply only, in milliseconds, excluding host<->device
 *     copies (src/cuda/cuda_utilities.cpp:828,872-875).
 */
#ifndef SPARTA_AMD_H
#define SPARTA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define SPARTA_OK               0
#define SPARTA_ERR_INVALID     -1   /* bad argument (null pointer, negative size, unsorted row, ...) */
#define SPARTA_ERR_ALLOC       -2   /* host or device allocation failed */
#define SPARTA_ERR_HIP         -3   /* a HIP runtime call failed (message has the HIP error string) */
#define SPARTA_ERR_UNSUPPORTED -4   /* valid request this build does not implement */
#define SPARTA_ERR_IO          -5   /* file could not be read / parsed */
#define SPARTA_ERR_NO_DEVICE   -6   /* no gfx950 device visible: the product path has NO CPU fallback */

/* ---- enums -------------------------------------------------------------------------------- */
/* storage/compute type of the VBS values on the device; accumulation is always fp32 */
#define SPARTA_F32  0
#define SPARTA_F16  1
#define SPARTA_BF16 2

#define SPARTA_COL_MAJOR 0          /* the reference's layout for B and C */
#define SPARTA_ROW_MAJOR 1

#define SPARTA_PTR_HOST   0         /* B and C are host buffers: copied in/out, dt excludes the copies */
#define SPARTA_PTR_DEVICE 1         /* B and C are device buffers: stream-ordered, nothing is copied */

/* BlockingType of the reference, same numeric values (include/definitions.h:17; flag -a) */
#define SPARTA_BLOCKING_ITERATIVE            0
#define SPARTA_BLOCKING_ITERATIVE_STRUCTURED 1   /* m:n structured variant (IterativeBlockingPatternMN, blocking.cpp:19-87) */
#define SPARTA_BLOCKING_FIXED_SIZE           2
#define SPARTA_BLOCKING_ITERATIVE_CLOCKED    3   /* the reference's default (include/input.h:27) */
#define SPARTA_BLOCKING_ITERATIVE_QUEUE      4
#define SPARTA_BLOCKING_ITERATIVE_MAX_SIZE   5   /* dispatches to IterativeBlockingKeeper (blocking.cpp:655) */
#define SPARTA_BLOCKING_SCRAMBLE             6
#define SPARTA_BLOCKING_MINHASH              7   /* EXTENSION (not in the reference): LSH-bucketed clustering with the same merge rule, for inputs
                                                   the quadratic scans cannot handle (approximate: see sparta_amd/csrc/reorder.cpp) */

/* similarity measure, flag -m (include/input.h:29, src/general/blocking.cpp:699-717) */
#define SPARTA_SIM_HAMMING 0
#define SPARTA_SIM_JACCARD 1

/* kernel selection for sparta_vbs_spmm (`algo` argument).  Within SPARTA_SPMM_MFMA the library picks between its
 * product paths by measuring them once per (n_cols, layouts) on the handle; env SPARTA_PATH=stream|class|generic forces one. */
#define SPARTA_SPMM_MFMA  0   /* hand-written MFMA kernels (the product path) */
#define SPARTA_SPMM_EXACT 1   /* fp32 only: unfused mul+add in the reference's summation order,
                                 bit-identical to VBR::multiply on finite inputs (slow; for parity) */

/* ---- reorder (host-side C++) -------------------------------------------------------------- */
/* replaces the configuration fields of class BlockingEngine (include/blocking.h:12-23) as filled
 * from the command line by BlockingEngine(CLineReader&) (src/general/blocking.cpp:678-688). */
typedef struct sparta_reorder_cfg {
    int32_t blocking_algo;     /* SPARTA_BLOCKING_*            (-a, default 3)    */
    int32_t sim_measure;       /* SPARTA_SIM_*                 (-m, default 1)    */
    float   tau;               /* merge threshold, dist <= tau (-t, default 0.1)  */
    int32_t use_groups;        /* weight distances by cluster size (-g, default 0)*/
    int64_t col_block_size;    /* w                            (-b, default 3)    */
    int64_t row_block_size;    /* max / fixed block-row height (-B, default 3)    */
    int32_t use_pattern;       /* merge rows into the pattern  (-p, default 1)    */
    int32_t force_fixed_size;  /* re-chunk into equal heights  (-F, default 0)    */
    int32_t structured_m;      /* blocking_algo 1 only: at most m hits per column ... (include/blocking.h:20, default 2) */
    int32_t structured_n;      /* ... inside every run of n merged rows               (include/blocking.h:21, default 4) */
    int32_t minhash_bands;     /* blocking_algo 7 only: LSH bands (0 = 16) ...                                              */
    int32_t minhash_rows;      /* ... minhash values per band (0 = from tau), ...                                            */
    int32_t minhash_max_eval;  /* ... exact comparisons per seed at most (0 = 512), ...                                       */
    int32_t minhash_max_rows;  /* ... a cluster never grows beyond this many rows (0 = unlimited); only a seed's own identical copies can exceed it */
} sparta_reorder_cfg;

/* replaces the measuring fields of BlockingEngine (include/blocking.h:28-42) */
typedef struct sparta_reorder_stats {
    int64_t comparison_counter;
    int64_t merge_counter;
    float   average_row_distance;
    float   average_merge_tau;
    float   timer_total;        /* microseconds, as in the reference (blocking.cpp:236-238) */
    float   timer_comparisons;
    float   timer_merges;
    int32_t reserved;
} sparta_reorder_stats;

/* fills *cfg with the reference's command-line defaults (include/input.h:15-42) */
void sparta_reorder_cfg_default(sparta_reorder_cfg* cfg);

/* replaces std::vector<intT> BlockingEngine::GetGrouping(const CSR&)  (include/blocking.h:47,
 * src/general/blocking.cpp:633-676).  The CSR is flat: rowptr[rows+1], colidx ascending within each
 * row.  grouping_out[rows] receives one group id per row (the reference's convention: the id is the
 * seed row; algo 5 numbers incomplete clusters seed+rows).  stats may be NULL. */
int sparta_reorder(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx,
                   const sparta_reorder_cfg* cfg, int64_t* grouping_out, sparta_reorder_stats* stats);

/* replaces get_permutation / get_partition / get_fixed_size_grouping (src/general/utilities.cpp:8-54).
 * perm_out[new_row] = old_row.  part_out needs room for n+1 entries; *n_part_out receives the count
 * (block_rows + 1). */
int sparta_get_permutation(const int64_t* grouping, int64_t n, int64_t* perm_out);
int sparta_get_partition(const int64_t* grouping, int64_t n, int64_t* part_out, int64_t* n_part_out);
int sparta_get_fixed_size_grouping(const int64_t* grouping, int64_t n, int64_t row_block_size, int64_t* grouping_out);

/* replaces HammingDistanceGroup / JaccardDistanceGroup (src/general/blocking.cpp:859-994): distance
 * between a cluster pattern row_a (column ids, weight group_a) and a row row_b (weight group_b) on
 * column blocks of width block_size. */
int sparta_row_distance(int32_t sim_measure, const int64_t* row_a, int64_t size_a, int64_t group_a,
                        const int64_t* row_b, int64_t size_b, int64_t group_b, int64_t block_size, float* dist_out);

/* replaces merge_rows (src/general/utilities.cpp:145-173), including its lossy behaviour.
 * out needs room for size_a + size_b entries. */
int sparta_merge_rows(const int64_t* row_a, int64_t size_a, const int64_t* row_b, int64_t size_b,
                      int64_t* out, int64_t* size_out);

/* ---- on-disk formats either side of the path (host-side C++, sparta_amd/csrc/io.cpp) ------- */
#define SPARTA_FMT_EL  0            /* MatrixFormat::el  (include/definitions.h:15; flag -R 0) */
#define SPARTA_FMT_MTX 1            /* MatrixFormat::mtx (flag -R 1) */
#define SPARTA_IO_COMPAT 0          /* what the reference's readers do, quirks included */
#define SPARTA_IO_STRICT 1          /* the formats as documented: no dropped line, MatrixMarket banner/values/symmetry honoured */

/* a CSR in flat arrays, owned by the library (release with sparta_csr_host_free) */
typedef struct sparta_csr_host {
    int64_t rows, cols, nnz;
    int32_t pattern_only;         /* 1: vals == NULL, every stored entry counts as 1 (include/matrices.h:28) */
    int64_t* rowptr;              /* [rows + 1] */
    int32_t* colidx;              /* [nnz], in file order inside a row (the reference never sorts a row) */
    float*   vals;                /* [nnz] or NULL */
} sparta_csr_host;

/* replaces CSR::read_from_edgelist(infile, delimiter, pattern_only, mat_fmt, symmetrize)
 * (include/matrices.h:63, src/general/csr.cpp:183-365).  SPARTA_IO_COMPAT: `.el` -- leading '#'/'%' lines skipped, THE
 * FIRST REMAINING LINE IS DISCARDED (csr.cpp:213), "row<delim>col[<delim>value]" per line, row ids non-decreasing, rows =
 * last row id + 1, cols = largest column id + 1, symmetrize mirrors an upper-triangular pattern-only input; `.mtx` -- always
 * pattern-only, one line after the size line is skipped, symmetry ignored (csr.cpp:309-365).  Inputs on which the
 * reference throws or runs into undefined behaviour return SPARTA_ERR_IO.  SPARTA_IO_STRICT reads the documented formats. */
int sparta_csr_read(const char* path, const char* delimiter, int32_t pattern_only, int32_t mat_fmt, int32_t symmetrize, int32_t mode,
                    sparta_csr_host* out);
/* the same readers on a text already in memory (what is left in the std::ifstream the reference's CSR constructor is handed,
 * include/matrices.h:58-63) */
int sparta_csr_read_buffer(const char* text, int64_t len, const char* delimiter, int32_t pattern_only, int32_t mat_fmt, int32_t symmetrize,
                           int32_t mode, sparta_csr_host* out);
void sparta_csr_host_free(sparta_csr_host* m);

/* replaces CSR::save_to_edgelist (src/general/csr.cpp:169-179): "i<delim>j" per entry (el) / "j<delim>i" (mtx flavour) */
int sparta_csr_write_edgelist(const char* path, int64_t rows, const int64_t* rowptr, const int32_t* colidx, const char* delimiter,
                              int32_t mat_fmt);

/* the grouping file `<outfile>.g` that save_blocking_data writes (src/general/utilities.cpp:239-243): one group id per line */
int sparta_grouping_write(const char* path, const int64_t* grouping, int64_t n);
/* replaces read_grouping_file + the count-line rule of test/general/Matrix_Analysis.cpp:10-32,77-78: lines that do not start
 * with a number are skipped; a file with expected_rows + 1 numbers has a leading count, which is dropped.  expected_rows < 0:
 * no check.  *n_out = number of ids found (also on SPARTA_ERR_IO when it does not match expected_rows). */
int sparta_grouping_read(const char* path, int64_t expected_rows, int64_t* out, int64_t capacity, int64_t* n_out);

/* the reference's 32-column statistics row (src/general/utilities.cpp:175-236): one header line and one value line, every field
 * followed by ','; floats print as std::to_string(float) ("%f").  Field names are the reference's CSV column names. */
typedef struct sparta_csv_fields {
    const char* matrix;           /* CLineReader::filename_ */
    int64_t rows, cols, nonzeros;
    int32_t symmetrize, blocking_algo;
    float   tau;
    int32_t row_block_size, col_block_size, use_pattern, sim_use_groups, sim_measure, reorder;
    const char* exp_name;
    int32_t b_cols, warmup, exp_repetitions, multiplication_algo, n_streams;
    float   time_to_block, time_to_merge, time_to_compare;          /* microseconds (sparta_reorder_stats) */
    int64_t vbr_nzcount, vbr_nzblocks_count;                        /* sparta_blocking_info */
    float   vbr_average_height;
    int64_t vbr_longest_row, merge_counter, comparison_counter;
    float   average_merge_tau, average_row_distance;
    float   avg_time_multiply, std_time_multiply;                   /* milliseconds */
} sparta_csv_fields;
int sparta_blocking_csv_row(const sparta_csv_fields* f, char* header_out, int64_t header_cap, char* values_out, int64_t values_cap);

/* the row permutation of CSR::reorder_by_degree (src/general/csr.cpp:123-155; flag -r -1 / 1): perm_out[k] = old index of
 * the row that moves to position k (apply with CSR::permute_rows semantics, utilities.h:95-107) */
int sparta_degree_permutation(int64_t rows, const int64_t* rowptr, int32_t descending, int64_t* perm_out);

/* ---- VBS build (host-side C++) ------------------------------------------------------------- */
/* the five arrays + scalars of struct VBR (include/matrices.h:93-104), owned by the library */
typedef struct sparta_vbs_host {
    int64_t rows, cols;           /* padded up to block multiples when force_fixed_size */
    int64_t block_rows, block_cols;
    int64_t block_col_size;
    int64_t nztot;                /* number of stored values = sum h*w over nonzero blocks */
    int64_t nblocks;              /* number of nonzero blocks = sum nzcount */
    int64_t* row_part;            /* [block_rows + 1] */
    int64_t* nzcount;             /* [block_rows]     */
    int64_t* jab;                 /* [nblocks]        */
    float*   mab;                 /* [nztot]          */
} sparta_vbs_host;

/* replaces VBR::fill_from_CSR_inplace(cmat, grouping, col_block_size, row_block_size, force_fixed_size)
 * (include/matrices.h:118, src/general/vbr.cpp:135-237).  vals == NULL means pattern-only (every
 * stored nonzero is 1, vbr.cpp:217).  The caller releases *out with sparta_vbs_host_free. */
int sparta_vbs_build(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                     const int64_t* grouping, int64_t col_block_size, int64_t row_block_size, int32_t force_fixed_size,
                     sparta_vbs_host* out);
void sparta_vbs_host_free(sparta_vbs_host* v);

/* replaces VBR::fill_from_CSR(cmat, row_partition, block_size) (include/matrices.h:117, src/general/vbr.cpp:239-321): the rows
 * keep their order, block-row ib = rows [row_partition[ib], row_partition[ib+1]); repeated entries give block-rows of height 0
 * with nzcount 0, as in the reference.  Where the reference prints "PARTITION CHECK ERROR" and continues (vbr.cpp:253-257) this
 * returns SPARTA_ERR_INVALID. */
int sparta_vbs_build_partition(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                               const int64_t* row_partition, int64_t n_part, int64_t block_size, sparta_vbs_host* out);
/* replaces VBR::partition_check(candidate_part) (src/general/vbr.cpp:108-118): 0 = valid, 1 = empty, 2 = last entry != rows,
 * 3 = decreasing.  (A status of the partition, not a SPARTA_* code.) */
int sparta_vbs_partition_check(const int64_t* part, int64_t n_part, int64_t rows);

/* Binary VBS container ("SPARTAVB", little-endian, 96-byte header + the four arrays, FNV-1a checksum; layout in
 * sparta_amd/csrc/io.cpp): the reorder + build cost is paid once.  sparta_vbs_load verifies magic, sizes, checksum and the
 * structural invariants and fills *out (release with sparta_vbs_host_free).  No reference counterpart. */
int sparta_vbs_save(const char* path, const sparta_vbs_host* v);
int sparta_vbs_load(const char* path, sparta_vbs_host* out);

/* Blocked-ELL view of a fixed-square-block VBS: replaces prepare_cusparse_BLOCKEDELLPACK(VBR*, int* ell_blocksize, int* ellValue_cols,
 * int* ellColInd_rows, int* ellColInd_cols, int* num_blocks, intT** ellColInd, DataT_C** ellValues)
 * (src/cuda/cuda_utilities.cpp:1656-1710) -- the arrays the reference hands to cusparseCreateBlockedEll.  ell_blocksize =
 * block_col_size; *ell_cols_out = most blocks in a block-row; ell_col_ind: (rows / bs) x ell_cols, -1 = padding block;
 * ell_values: rows x (ell_cols * bs), row-major.  Call with ell_col_ind = ell_values = NULL to get ell_cols first.
 * SPARTA_ERR_INVALID where the reference exits (rows or cols not a multiple of the block size) and where its silent
 * assumption fails (a block-row that is not bs rows tall). */
int sparta_vbs_to_blocked_ell(const sparta_vbs_host* v, int64_t* ell_cols_out, int64_t* ell_col_ind, float* ell_values);

/* replaces BlockingEngine::CollectBlockingInfo (src/general/blocking.cpp:576-631): statistics of
 * the VBS a grouping would give, without building it. info_out: [VBR_nzcount, VBR_nzblocks_count,
 * VBR_longest_row]; avg_height_out: VBR_average_height. */
int sparta_blocking_info(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx,
                         const int64_t* grouping, int64_t col_block_size, int64_t* info_out, float* avg_height_out);

/* ---- device: VBS A x dense B (hand-written HIP, gfx950) ---------------------------------------- */
typedef struct sparta_vbs sparta_vbs_t;   /* opaque; owns the device image of A and the launch plan */

/* Uploads a VBS matrix (the reference's arrays, host pointers) to `device` once and builds the tile
 * plan.  Replaces the per-call cudaMalloc + H2D of A that every reference back-end performs
 * (src/cuda/cuda_utilities.cpp:779-789).  dtype SPARTA_F16 / SPARTA_BF16: the values are rounded (nearest even) and
 * re-laid-out for the 16-bit MFMA here, once (needs block_col_size % 32 == 0); products are accumulated in fp32. */
int sparta_vbs_create(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t block_col_size,
                      const int64_t* row_part, const int64_t* nzcount, const int64_t* jab, const float* mab,
                      int32_t dtype, int32_t device);

/* Same, restricted to block-rows [block_row_begin, block_row_end): the row-range partition used for
 * multi-GPU runs.  C of the resulting handle has row_part[end]-row_part[begin] rows (local numbering). */
int sparta_vbs_create_range(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t block_col_size,
                            const int64_t* row_part, const int64_t* nzcount, const int64_t* jab, const float* mab,
                            int64_t block_row_begin, int64_t block_row_end, int32_t dtype, int32_t device);

/* Device handle straight from the CSR and a grouping, for matrices whose VBS image is mostly zeros (clustered power-law graphs:
 * 98 % of the stored area): does what sparta_vbs_build + sparta_vbs_create do, except that the block-rows the sparse-row
 * kernels will take anyway are never expanded into dense blocks -- neither on the host nor on the device.  Same product as a
 * handle made the two-step way (same decisions, same kernels, same data).  Columns must be strictly ascending within a row.
 * SPARTA_SPMM_EXACT is not available on such a handle.  No reference counterpart (the reference always expands:
 * src/general/vbr.cpp:205-228). */
int sparta_vbs_create_from_csr(sparta_vbs_t** out, int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                               const int64_t* grouping, int64_t col_block_size, int64_t row_block_size, int32_t force_fixed_size,
                               int32_t dtype, int32_t device);

/* What sparta_vbs_create_from_csr WOULD build for this grouping, without building or uploading anything (no GPU needed): the same
 * per-block decisions (well-filled blocks -> dense MFMA tiles, the nonzeros of the others -> rows of (column, value) for the sparse-row
 * kernels).  stats[8] = {tile blocks, stored elements of the tiles, MFMA steps of the tiles, sparse nonzeros, sparse rows, block-rows,
 * rows, 0}.  Lets a caller compare two blockings of one matrix -- the clustering of sparta_reorder against the fixed grid of the
 * reference's `-a 2 -F 1` arm (src/scripts/run_multiplication_experiments_fixed_cluster.sh:14-16) -- before paying for either handle. */
int sparta_vbs_plan_stats(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals, const int64_t* grouping,
                          int64_t col_block_size, int64_t row_block_size, int32_t force_fixed_size, int32_t dtype, int64_t* stats);

/* C (+)= A * B.  Replaces VBR::multiply(B, B_cols, C) (include/matrices.h:121) and the GPU back-ends
 *   cublas_fixed_blocks_multiply / cublas_blockmat_batched / cutlas_* (const VBR&, DataT* B, int B_cols,
 *   DataT_C* C, float& dt[, int n_streams])            (include/cuda_utilities.h:38-44,
 *                                                        include/cutlass_bellpack_lib.h:19-25).
 * B: `cols` x n_cols, C: `rows` x n_cols (rows of this handle).  fp32 handles: fp32 B and C, any layout, any n_cols.
 * 16-bit handles: C is fp32; with SPARTA_PTR_DEVICE B is in the handle's 16-bit type, column-major, ldb even; any n_cols (the
 * reference's `-c` is arbitrary, include/input.h:15-42): whole 128-column slabs go through the kernels as they are, the last
 * n_cols % 128 columns through a zero-padded slab in per-handle scratch (128 columns of B and of C); with SPARTA_PTR_HOST B is fp32
 * like in the reference and is rounded on the device.  Tolerance of the 16-bit path: exact products of the ROUNDED inputs, fp32 accumulation -- the same bound
 * as the fp32 MFMA path relative to the reference's multiply run on the rounded inputs.  accumulate = 1 is the reference's
 * semantics (C += A*B); 0 overwrites C (every row of C is written).  ptr_space HOST: buffers are
 * copied to/from the device around the kernel and *dt_ms (may be NULL) covers the kernel only;
 * DEVICE: launched on `stream` (a hipStream_t, NULL = default stream); if dt_ms != NULL the call
 * records events and synchronises on them, otherwise it returns without synchronising.
 * Leading dimensions: the stream kernels address B and C with 32-bit byte offsets inside a column slab.  fp32 handles take the 64-bit
 * per-class kernels beyond ldb, ldc ~ 4.2 M elements (column-major; correct, slower); 16-bit handles run up to ldb < 34 M (16 M for SPARTA_H16_WIDE plans), ldc < 17 M elements
 * and return SPARTA_ERR_UNSUPPORTED beyond (a gathered B has the slab height as its leading dimension).
 * A handle carries per-handle scratch (split-tile workspace, layout copies of B, step lists of a gathered B): it must not run on two
 * streams at once.  The FIRST call of a shape (n_cols, layouts, shard_rows) on a handle may allocate that scratch, and on fp32 handles
 * times its two product paths once; every later call of the shape is kernel launches only and can be captured into a hipGraph.  A call
 * that would have to allocate or time while its stream is being captured returns SPARTA_ERR_UNSUPPORTED (and leaves the capture
 * intact): run it once outside the capture first. */
int sparta_vbs_spmm(sparta_vbs_t* A, const void* B, int64_t ldb, int32_t b_layout, int32_t n_cols,
                    void* C, int64_t ldc, int32_t c_layout, int32_t accumulate,
                    int32_t ptr_space, void* stream, int32_t algo, float* dt_ms);

/* Multi-GPU entry point: B_gathered is what ONE ncclAllGather (RCCL) of the ranks' row shards of B leaves on
 * every GPU: n_shards consecutive column-major slabs of shard_rows x n_cols (ld = shard_rows), slab s holding
 * rows [s*shard_rows, (s+1)*shard_rows) of B, consecutive slabs shard_stride elements apart.  shard_rows must
 * be a multiple of block_col_size (so that no B panel straddles two slabs) and cols == n_shards * shard_rows.
 * Device pointers only; 16-bit handles take the slabs in their 16-bit type (shard_rows and shard_stride even).
 * No reference counterpart (the reference is single-GPU: SURVEY.md section 2.1). */
int sparta_vbs_spmm_gathered(sparta_vbs_t* A, const void* B_gathered, int64_t shard_rows, int64_t shard_stride, int32_t n_cols,
                             void* C, int64_t ldc, int32_t c_layout, int32_t accumulate, void* stream, int32_t algo, float* dt_ms);
/* the same with the columns of a slab `shard_ld` >= shard_rows elements apart (sparta_vbs_spmm_gathered: shard_ld = shard_rows; 16-bit handles: shard_ld even).
 * Why: a leading dimension that is a multiple of a large power of two -- 2^20 rows of a 16-bit B: columns exactly 2 MB apart -- maps the columns of a panel of B
 * onto the same cache sets and memory channels: the hub kernel of a 16-bit handle measured 0.80 PFLOP/s with such a B and 1.05 with 64 elements of padding per
 * column (MI355X).  Pad the ranks' shard buffers before the all-gather (bench_parts.py does); for sparta_vbs_spmm the same advice holds for
 * ldb.  No reference counterpart (single-GPU). */
int sparta_vbs_spmm_gathered_ld(sparta_vbs_t* A, const void* B_gathered, int64_t shard_rows, int64_t shard_ld, int64_t shard_stride, int32_t n_cols,
                                void* C, int64_t ldc, int32_t c_layout, int32_t accumulate, void* stream, int32_t algo, float* dt_ms);

/* Multi-GPU, sparsity-aware exchange: gathers chunks of `block_bytes` bytes (a multiple of 16; one chunk = one
 * block_col_size x n_cols tile of B in the row-block-tiled layout) from `src` into consecutive chunks of `dst`:
 * dst chunk i = src chunk ids_dev[i].  All pointers are device pointers (16-byte aligned); stream-ordered.  A rank uses it
 * to assemble the send buffer of the ONE all-to-all that ships to every other rank exactly the row-blocks of B that rank's
 * slab of A touches (sparta_amd/dist.py: RowBlockExchange); the receive buffer is then read in place by
 * sparta_vbs_spmm_gathered(shard_rows = block_col_size, shard_stride = block_col_size * n_cols).
 * No reference counterpart (single-GPU). */
int sparta_pack_blocks(const void* src, int64_t block_bytes, const int32_t* ids_dev, int64_t n_blocks, void* dst, void* stream);

/* B * A (dense x VBS).  The reference's cublas_blockmat_multiplyBA(const VBR&, DataT* B, int B_rows, DataT_C* C, float& dt, int n_streams)
 * (include/cuda_utilities.h:40, src/cuda/cuda_utilities.cpp:553-721) is NOT a B * A (it offsets B by block_col_size * ib elements, uses
 * the first block-row's height for every block and sizes C as B_rows x A_rows: DESIGN.md section 8), so there is nothing to be identical
 * to; these two entries compute the product itself.  sparta_vbs_create_transposed uploads A^T (same VBS arrays as sparta_vbs_create);
 * sparta_vbs_spmm_ba: C (M x cols, column-major, ldc) (+)= B (M x rows, column-major, ldb) * A, rows of A in the VBS's (reordered) order.
 * fp32 tolerance as for sparta_vbs_spmm. */
int sparta_vbs_create_transposed(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t block_col_size,
                                 const int64_t* row_part, const int64_t* nzcount, const int64_t* jab, const float* mab,
                                 int32_t dtype, int32_t device);
int sparta_vbs_spmm_ba(sparta_vbs_t* At, const void* B, int64_t ldb, int32_t M, void* C, int64_t ldc, int32_t accumulate,
                       int32_t ptr_space, void* stream, float* dt_ms);

/* A dense operand that does NOT change between products, prepared once.  The reference's drivers multiply the same B `-x` times
 * (test/cuda/cuda_multiply.cpp:250-269); the sparse-row kernels of a handle read B row-major, so a column-major (the reference's layout)
 * or gathered B is otherwise transposed into per-handle scratch on EVERY product (11 % of a power-law product).
 * sparta_vbs_prepare_b makes that copy once, on `stream`, into memory the returned object owns; B itself is not copied and must stay
 * valid and unchanged while the object is used.  shard_rows = 0: B is column-major cols x n_cols with leading dimension ldb; shard_rows >
 * 0: the gathered layout of sparta_vbs_spmm_gathered_ld with ldb = the column stride inside a slab (shard_ld; 0 = unpadded = shard_rows).  Device pointers only.  sparta_vbs_spmm_prepared is sparta_vbs_spmm /
 * sparta_vbs_spmm_gathered on that B (SPARTA_SPMM_MFMA; the exception state of a failed call is a plain status code: the handle is usable
 * again).  The implicit per-call transpose of sparta_vbs_spmm stays the default: nothing changes for a caller that never prepares. */
typedef struct sparta_b sparta_b_t;
int sparta_vbs_prepare_b(sparta_vbs_t* A, const void* B, int64_t ldb, int64_t shard_rows, int64_t shard_stride, int32_t n_cols, void* stream,
                         sparta_b_t** out);
int sparta_vbs_spmm_prepared(sparta_vbs_t* A, const sparta_b_t* Bp, void* C, int64_t ldc, int32_t c_layout, int32_t accumulate, void* stream,
                             float* dt_ms);
int sparta_b_destroy(sparta_b_t* Bp);

/* Per-tile-class device timing for roofline reports: when enabled, sparta_vbs_spmm brackets each class
 * launch with HIP events on the launch stream; sparta_vbs_class_times waits for them and writes the last
 * call's milliseconds into ms_out[4]: stream path -> {stream kernel, fix-up kernel, 0, 0};
 * per-class / generic path -> {<=16-row class, <=32-row class, <=64-row class, sparse-row kernels}.
 * (Block-rows whose blocks are nearly empty are multiplied as sparse rows, HBM-bound, instead of as MFMA tiles: see
 * sparta_amd/csrc/vbs_spmm.hip "sparse-row path"; env SPARTA_SPARSE_K=0 keeps everything on the MFMA kernels.) */
int sparta_vbs_set_class_timing(sparta_vbs_t* A, int32_t enable);
int sparta_vbs_class_times(sparta_vbs_t* A, float* ms_out);

/* Shader clock of the last timed call, per launch slot as in sparta_vbs_class_times (mhz_out[4], 0 where no probe ran).
 * With class timing enabled, workgroup 0 of each product kernel reads s_memtime (shader-clock cycles) and s_memrealtime
 * (constant 100 MHz) at entry and exit; the ratio is the clock the MFMA pipes really ran at.  Under a dense fp32 MFMA
 * load on random data MI355X settles well below its 2.4 GHz peak clock (power limit), which scales the attainable
 * fraction of the 157.3 TFLOP/s fp32 matrix peak: bench.py reports it next to the roofline.  No reference counterpart. */
int sparta_vbs_clock_mhz(sparta_vbs_t* A, double* mhz_out);

int sparta_vbs_destroy(sparta_vbs_t* A);

/* plan / roofline introspection. info_out (int64[16]):
 *  [0] rows [1] cols [2] block_rows [3] block_col_size [4] nblocks [5] nztot (area)
 *  [6] tiles of <=16 rows [7] <=32 rows [8] <=64 rows [9] 0 [10] device bytes of A
 *  [11] padded MFMA rows summed over (tile, block) pairs x w (executed area)
 *  [12] steps of the stream plan [13] stream workers [14] split tiles
 *  [15] kernel path of the last sparta_vbs_spmm: 1 stream, 2 per-class, 3 generic */
int sparta_vbs_info(const sparta_vbs_t* A, int64_t* info_out);

/* the sparse-row part of the plan. info_out (int64[4]): [0] rows [1] nonzeros kept as (column, value) pairs
 * [2] rows handled one wave each [3] hub rows (cut into segments) */
int sparta_vbs_sparse_info(const sparta_vbs_t* A, int64_t* info_out);

/* the resident-column image of a small fp32 handle (no reference counterpart: the reference multiplies every matrix block by block, vbr.cpp:323-372; this is the
 * product path of its real matrices -- 8-22 k rows, no dense blocks at any block size it sweeps -- at its operand widths B_COLs = 1024 / 8192,
 * scripts/run_multiplication_experiments_fixed_cluster.sh:6-7).  Built when at least half the rows of the handle are on the sparse-row path -- the launches of its MFMA
 * tiles, if it has any, come first; the sparse part of a mixed block-row ADDS to what they stored -- and a column of B and of C fits LDS (rows, columns <= 40 960); taken by sparta_vbs_spmm when B and C are column-major (the reference's layouts) device or host pointers: NC columns of
 * B are copied into LDS, A (length-sorted rows, 64 to a slice, long rows cut into chunks) streams past them from L2, one launch, B and C cross HBM once; a row's
 * nonzeros are added in ascending column order as in CSR::multiply (csr.cpp:49-65).  SPARTA_COLRES=0 at create time: not built (the row gather takes the product).
 * info_out (int64[12]): [0] slices of 64 slots of the part with most (0: no image) [1] stored entries, padding included [2] rows cut into chunks [3] cells a column set
 * needs in LDS (the largest range of B + 4, or the largest staging image: rows + extra cells of the chunks) [4] longest slot [5] columns per workgroup of the last product on this path (0: the last product took another path)
 * [6] nonzeros [7] 1: every stored value is 1.0f and the image holds columns only (the reference's pattern-only runs, -P 1)
 * [8] parts the rows of C are cut into, [9] K ranges the columns of A are cut into (1, 1: a column of B and of C fits LDS whole; up to 4 x 4: rows, columns <= 163 k --
 * the workgroup of a part walks the ranges one after the other with its sums in registers, B is read once per part)
 * [10] parts of the SECOND image a handle keeps for products of few column sets (4; 0: none -- fewer than 2048 rows, or SPARTA_COLRES_SMALL=0): with at most 64 column sets the
 * product is one round of 4 x the workgroups, each streaming a quarter of A; [11] 1: the last product used it */
int sparta_vbs_colres_info(const sparta_vbs_t* A, int64_t* info_out);
/* HOST-side walk of that image for one column x of B (y = A x; rows of C through crow, NULL = identity; crow[i] + 2^31: row i ADDS to y -- the sparse part of a mixed block-row --,
 * crow[i] = -1: CSR row i is not a sparse row and y[i] is left alone -- a block-row of tiles; info_out as above, [0] = 0 and y untouched when the matrix
 * gets no image): slots in slice order, a slot's entries in order, the chunks of a long row added in chunk order -- the arithmetic of the kernel, for the CPU suite
 * to check the builder with.  Not a product path (and not a fallback: sparta_vbs_spmm never calls it). */
int sparta_colres_host_check(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals, const int64_t* crow, const float* x,
                             float* y, int64_t* info_out);

/* the column-compacted ("union-pattern") tiles of a handle made by sparta_vbs_create_from_csr (fp32, and since the round's second half fp16 / bf16 storage: the same tiles through the 16-bit matrix instruction).  They replace what the reference's builder stores for a cluster at
 * SMALL block widths -- VBR::fill_from_CSR_inplace flags, per block-row, exactly the column blocks its rows touch and stores them back to back
 * (src/general/vbr.cpp:177-228; at -b 1 that is a dense rows x |union| tile plus the ascending column list, the ids its Jaccard distance is defined on,
 * src/general/blocking.cpp:923-994) -- and what VBR::multiply does with it (vbr.cpp:323-372): a block-row whose rows share columns but whose w-wide blocks would be
 * nearly empty is kept as tiles of <= 64 consecutive reordered rows x the columns at least 2-3 of those rows use, multiplied on the matrix cores against the
 * gathered rows of a row-major B (k_union.hip; a column-major B is transposed once per product, or once per sparta_vbs_prepare_b); the nonzeros of the other
 * columns are sparse rows that add.  SPARTA_UNION=0 at create time: not built.  SPARTA_SPMM_EXACT is not available on such a handle (as for any handle with sparse rows).
 * A row's nonzeros in columns too thinly used for the list ride in the tile's TAIL (at most SPARTA_UNION_TAIL = 16 per row) and are added in the tile's epilogue; only what
 * exceeds that is left to the sparse-row kernels.
 * The matrix instruction works on row tiles of 16 rows (fp32 handles: v_mfma_f32_16x16x4_f32) or 32 rows (16-bit handles): a tile of mt rows is multiplied as ceil(mt / 16) resp.
 * ceil(mt / 32) row tiles.
 * info_out (int64[12]): [0] tiles of <= 32 rows [1] tiles of 33..64 rows [2], [3] their 32-deep steps [4] stored elements (tile rows x list entries)
 * [5] list entries [6] nonzeros the tiles hold (lists + tails) [7] persistent workgroups of the launch [8] rows of C the tiles own [9] nonzeros in the tails
 * [10] elements the kernel multiplies (steps x 32 x the tile's rows rounded up to whole row tiles) [11] rows per row tile (16 or 32; 0: no tiles) */
int sparta_vbs_union_info(const sparta_vbs_t* A, int64_t* info_out);
/* HOST-side check of that builder and its device plan for the CPU suite (no GPU; not a product path, and not a fallback: sparta_vbs_spmm never calls it): the hybrid
 * image of the CSR matrix under `grouping` -- w-wide tiles, column-compacted tiles, sparse rows, decided as sparta_vbs_create_from_csr decides them for an fp32 handle --
 * multiplied with ONE column x; the column-compacted tiles are walked in their DEVICE form (step records, list entries, MFMA-fragment-order slices, dealt to
 * `max_workers` workers).  y (double, [rows padded as force_fixed_size pads them], reordered order).  info_out (int64[14]): [0..6] as sparta_vbs_union_info,
 * [7] nonzeros left to the sparse rows [8] stored elements of the w-wide tiles [9] padded rows [10], [11] workers of the two step lists [12] nonzeros in the tails [13] rows the tiles own */
int sparta_union_host_check(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals, const int64_t* grouping, int64_t col_block_size,
                            int64_t row_block_size, int32_t force_fixed_size, int32_t max_workers, const float* x, double* y, int64_t* info_out);

/* the hub part of the plan of a 16-bit handle of 64-wide blocks (no reference counterpart: the reference multiplies every block-row the same way,
 * vbr.cpp:323-372 / cuda_utilities.cpp:828-875).  The long tiles of 33..64 rows -- the dense hub of a power-law matrix under the fixed 64 x 64 grid -- are grouped
 * by the Jaccard similarity of their block-column sets into group tiles of 2 or 4 and multiplied by a GEMM-shaped kernel that stages A and B once per workgroup.
 * info_out (int64[10]): [0] steps (one 64-deep k slice of one group tile) [1] 64-row tiles in the hub plan [2] group tiles [3] tiles per group tile (2 or 4; 0: no
 * hub plan) [4] stored elements of those tiles [5] elements the kernel multiplies (the union of a group's block columns x its tiles) [6] workers [7] K chunks of
 * the step order [8] segments (runs of one group tile on one worker: each one that is not a whole tile leaves partial images for the fix-up) [9] 0.
 * SPARTA_HUB=0 switches the hub plan off (every tile then runs on the 64-row plan as before). */
int sparta_vbs_hub_info(const sparta_vbs_t* A, int64_t* info_out);

/* number of visible HIP devices (0 when none); never initialises a device context */
int sparta_device_count(void);

const char* sparta_last_error(void);
const char* sparta_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPARTA_AMD_H */
