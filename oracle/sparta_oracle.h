/*
 * sparta_oracle.h -- CPU ORACLE for the block-sparse SpMM hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's algorithm (HicrestLaboratory/SPARTA, src/general/{csr,vbr,blocking,utilities}.cpp),
 * function by function, each citing the reference file:line it follows.  It exists to CHECK the product
 * (sparta_amd/csrc: host C++ + HIP kernels); the product never includes, links or calls it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against the golden vectors under
 * tests/golden/ (generated from the compiled reference by tests/golden/make_golden.py) and against the
 * reference's own known-answer tests; tests/test_oracle_vs_ref.py compares it with
 * oracle/_ref/libsparta_ref.so (the real reference) on seeded random inputs when that library is present.
 *
 * Types follow the reference: intT = long (include/definitions.h:4), DataT = DataT_C = float (:5-6).
 * Matrices are passed as flat CSR (rowptr/colidx/vals) instead of the reference's per-row arrays.
 */
#ifndef SPARTA_ORACLE_H
#define SPARTA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* src/general/blocking.cpp:859-921 */
float oracle_hamming_distance_group(const long* row_A, long size_A, long group_size_A, const long* row_B, long size_B,
                                    long group_size_B, long block_size);
/* src/general/blocking.cpp:923-994 */
float oracle_jaccard_distance_group(const long* row_A, long size_A, long group_size_A, const long* row_B, long size_B,
                                    long group_size_B, long block_size);
/* src/general/utilities.cpp:145-173; result needs size_A + size_B entries; returns the result length */
long oracle_merge_rows(const long* A, long size_A, const long* B, long size_B, long* result);

/* src/general/utilities.cpp:8-20 (std::sort == libstdc++ 11 introsort, restated in sparta_oracle.c) */
void oracle_get_permutation(const long* grouping, long n, long* perm);
/* src/general/utilities.cpp:22-43; partition needs n + 1 entries; returns its length */
long oracle_get_partition(const long* grouping, long n, long* partition);
/* src/general/utilities.cpp:45-54 */
void oracle_get_fixed_size_grouping(const long* grouping, long n, long row_block_size, long* result);

/* BlockingEngine::GetGrouping, src/general/blocking.cpp:633-676, for blocking_algo in
 *   0 iterative (:89-154), 2 fixed_size (:554-562), 3 iterative_clocked (:156-243), 4 iterative_queue (:245-338),
 *   5 iterative_max_size == IterativeBlockingKeeper (:433-549; needs libstdc++'s red-black tree, restated in the .c).
 * sim_measure: 0 Hamming, 1 Jaccard.  counters (may be NULL): [comparison_counter, merge_counter].
 * Returns 0, or -1 for an algorithm the oracle does not restate (6 scramble).  Algorithm 1 (m:n structured) runs with the
 * reference's default m = 2, n = 4 here; oracle_get_grouping_mn takes them explicitly. */
int oracle_get_grouping(long rows, const long* rowptr, const long* colidx, int blocking_algo, int sim_measure, float tau,
                        long col_block_size, long row_block_size, int use_groups, int use_pattern, int force_fixed_size,
                        long* grouping, long* counters);
int oracle_get_grouping_mn(long rows, const long* rowptr, const long* colidx, int blocking_algo, int sim_measure, float tau,
                           long col_block_size, long row_block_size, int use_groups, int use_pattern, int force_fixed_size,
                           int structured_m, int structured_n, long* grouping, long* counters);

/* VBR::fill_from_CSR_inplace, src/general/vbr.cpp:135-237.
 * Two calls: with mab == NULL it only fills dims_out = {rows, cols, block_rows, block_cols, nztot, nblocks} (and
 * row_part / nzcount / jab when those are non-NULL and large enough is the caller's job: row_part n+1, nzcount n,
 * jab up to block_rows*block_cols); with mab != NULL (nztot floats) it also scatters the values.
 * vals == NULL means pattern_only (vbr.cpp:217). */
int oracle_vbr_fill_inplace(long cmat_rows, long cmat_cols, const long* rowptr, const long* colidx, const float* vals,
                            const long* grouping, long col_block_size, long row_block_size, int force_fixed_size,
                            long* dims_out, long* row_part, long* nzcount, long* jab, float* mab);

/* VBR::multiply, src/general/vbr.cpp:323-372: C += A*B, B column-major ld = cols, C column-major ld = rows.
 * Where the reference reads B past row `cols` (last, zero-padded block column; vbr.cpp:351,362) the oracle uses 0. */
void oracle_vbr_multiply(long rows, long cols, long block_rows, long block_col_size, const long* row_part, const long* nzcount,
                         const long* jab, const float* mab, const float* B, int B_cols, float* C);
/* same arithmetic restricted to block-rows [ib0, ib1) -- used for sampled CPU baselines */
void oracle_vbr_multiply_range(long rows, long cols, long block_col_size, const long* row_part, const long* nzcount,
                               const long* jab, const float* mab, long ib0, long ib1, const float* B, int B_cols, float* C);

/* CSR::multiply, src/general/csr.cpp:49-65: C += A*B; ldb is explicit (the reference hard-codes `rows`, :61) */
void oracle_csr_multiply(long rows, const long* rowptr, const long* colidx, const float* vals, const float* B, long ldb,
                         long B_cols, float* C);

/* BlockingEngine::CollectBlockingInfo, src/general/blocking.cpp:576-631.
 * info_out: [VBR_nzcount, VBR_nzblocks_count, VBR_longest_row]; avg_height_out: VBR_average_height */
void oracle_collect_blocking_info(long rows, long cols, const long* rowptr, const long* colidx, const long* grouping,
                                  long col_block_size, long* info_out, float* avg_height_out);

#ifdef __cplusplus
}
#endif
#endif
