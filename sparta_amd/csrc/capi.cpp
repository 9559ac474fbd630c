// capi.cpp -- extern "C" glue for the host-side part of include/sparta_amd.h
// (the device part lives in vbs_capi.cpp + the k_*.hip kernel translation units).  No exception leaves this file.
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <exception>
#include <new>

#include "host_core.hpp"

namespace sparta {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }

int validate_csr(const CsrView& a, bool need_sorted) {
    if (a.rows < 0 || a.cols < 0) return fail(SPARTA_ERR_INVALID, "CSR: negative dimension");
    if (a.rows > 0 && (!a.rowptr)) return fail(SPARTA_ERR_INVALID, "CSR: rowptr is NULL");
    if (a.rows == 0) return SPARTA_OK;
    if (a.rowptr[0] != 0) return fail(SPARTA_ERR_INVALID, "CSR: rowptr[0] must be 0");
    // rows are checked on all host threads (10^9-nonzero inputs); the FIRST offending row is what gets reported
    std::atomic<int64_t> bad_row{INT64_MAX};
    std::atomic<int> bad_kind{0};
    parallel_for_dynamic(a.rows, 4096, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; i++) {
            int64_t n = a.rowptr[i + 1] - a.rowptr[i];
            int kind = 0;
            if (n < 0) kind = 1;
            else if (n > 0 && !a.colidx) kind = 2;
            else {
                const int32_t* r = a.colidx + a.rowptr[i];
                for (int64_t k = 0; k < n && !kind; k++) {
                    if (r[k] < 0 || (int64_t)r[k] >= a.cols) kind = 3;
                    // the reference's distance/merge code assumes ascending rows (its readers throw on unsorted
                    // row ids, csr.cpp:259-260, and never sort columns); duplicates make merge_rows ill-defined.
                    else if (need_sorted && k > 0 && r[k] <= r[k - 1]) kind = 4;
                }
            }
            if (kind) {
                int64_t cur = bad_row.load();
                while (i < cur && !bad_row.compare_exchange_weak(cur, i)) {}
                if (bad_row.load() == i) bad_kind.store(kind);
                return;                                                    // (later rows of this chunk cannot be the first)
            }
        }
    });
    if (bad_row.load() != INT64_MAX) {
        // re-derive the kind for the first bad row (another thread may have stored its own kind in between)
        const int64_t i = bad_row.load();
        const int64_t n = a.rowptr[i + 1] - a.rowptr[i];
        if (n < 0) return fail(SPARTA_ERR_INVALID, "CSR: rowptr must be non-decreasing (row " + std::to_string(i) + ")");
        if (n > 0 && !a.colidx) return fail(SPARTA_ERR_INVALID, "CSR: colidx is NULL");
        const int32_t* r = a.colidx + a.rowptr[i];
        for (int64_t k = 0; k < n; k++) {
            if (r[k] < 0 || (int64_t)r[k] >= a.cols) return fail(SPARTA_ERR_INVALID, "CSR: column index out of range in row " + std::to_string(i));
            if (need_sorted && k > 0 && r[k] <= r[k - 1])
                return fail(SPARTA_ERR_INVALID, "CSR: columns must be strictly ascending within a row (row " + std::to_string(i) + ")");
        }
    }
    return SPARTA_OK;
}

}  // namespace sparta

using namespace sparta;

#define SPARTA_TRY try {
#define SPARTA_CATCH                                                                  \
    }                                                                                 \
    catch (const std::bad_alloc&) { return fail(SPARTA_ERR_ALLOC, "out of host memory"); } \
    catch (const std::exception& e) { return fail(SPARTA_ERR_INVALID, e.what()); }    \
    catch (...) { return fail(SPARTA_ERR_INVALID, "unknown C++ exception"); }

extern "C" {

const char* sparta_last_error(void) { return g_last_error.c_str(); }
const char* sparta_version(void) { return "sparta_amd 0.1 (gfx950)"; }

void sparta_reorder_cfg_default(sparta_reorder_cfg* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->blocking_algo = SPARTA_BLOCKING_ITERATIVE_CLOCKED;   // include/input.h:27
    c->sim_measure = SPARTA_SIM_JACCARD;                    // :29
    c->tau = 0.1f;                                          // :33
    c->use_groups = 0;                                      // :21
    c->col_block_size = 3;                                  // :31
    c->row_block_size = 3;                                  // :32
    c->use_pattern = 1;                                     // :22
    c->force_fixed_size = 0;                                // :24
    c->structured_m = 2;                                    // include/blocking.h:20
    c->structured_n = 4;                                    // :21
}

int sparta_reorder(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const sparta_reorder_cfg* cfg,
                   int64_t* grouping_out, sparta_reorder_stats* stats) {
    SPARTA_TRY
    if (!cfg || (!grouping_out && rows > 0)) return fail(SPARTA_ERR_INVALID, "sparta_reorder: NULL argument");
    CsrView a; a.rows = rows; a.cols = cols; a.rowptr = rowptr; a.colidx = colidx;
    return reorder(a, *cfg, grouping_out, stats);
    SPARTA_CATCH
}

int sparta_get_permutation(const int64_t* grouping, int64_t n, int64_t* perm_out) {
    SPARTA_TRY
    if (n < 0 || (n > 0 && (!grouping || !perm_out))) return fail(SPARTA_ERR_INVALID, "sparta_get_permutation: bad argument");
    std::vector<int64_t> p = get_permutation(grouping, n);
    std::copy(p.begin(), p.end(), perm_out);
    return SPARTA_OK;
    SPARTA_CATCH
}

int sparta_get_partition(const int64_t* grouping, int64_t n, int64_t* part_out, int64_t* n_part_out) {
    SPARTA_TRY
    if (n < 0 || !part_out || !n_part_out || (n > 0 && !grouping)) return fail(SPARTA_ERR_INVALID, "sparta_get_partition: bad argument");
    std::vector<int64_t> p = get_partition(grouping, n);
    std::copy(p.begin(), p.end(), part_out);
    *n_part_out = (int64_t)p.size();
    return SPARTA_OK;
    SPARTA_CATCH
}

int sparta_get_fixed_size_grouping(const int64_t* grouping, int64_t n, int64_t row_block_size, int64_t* grouping_out) {
    SPARTA_TRY
    if (n < 0 || row_block_size <= 0 || (n > 0 && (!grouping || !grouping_out)))
        return fail(SPARTA_ERR_INVALID, "sparta_get_fixed_size_grouping: bad argument");
    std::vector<int64_t> g = get_fixed_size_grouping(grouping, n, row_block_size);
    std::copy(g.begin(), g.end(), grouping_out);
    return SPARTA_OK;
    SPARTA_CATCH
}

int sparta_row_distance(int32_t sim_measure, const int64_t* row_a, int64_t size_a, int64_t group_a, const int64_t* row_b,
                        int64_t size_b, int64_t group_b, int64_t block_size, float* dist_out) {
    SPARTA_TRY
    if (size_a < 0 || size_b < 0 || block_size <= 0 || !dist_out || (size_a > 0 && !row_a) || (size_b > 0 && !row_b))
        return fail(SPARTA_ERR_INVALID, "sparta_row_distance: bad argument");
    *dist_out = row_distance(sim_measure, row_a, size_a, group_a, row_b, size_b, group_b, block_size);
    return SPARTA_OK;
    SPARTA_CATCH
}

int sparta_merge_rows(const int64_t* row_a, int64_t size_a, const int64_t* row_b, int64_t size_b, int64_t* out, int64_t* size_out) {
    SPARTA_TRY
    if (size_a < 0 || size_b < 0 || !size_out || (size_a > 0 && !row_a) || (size_b > 0 && !row_b) || (size_a + size_b > 0 && !out))
        return fail(SPARTA_ERR_INVALID, "sparta_merge_rows: bad argument");
    std::vector<int64_t> r = merge_rows(row_a, size_a, row_b, size_b);
    std::copy(r.begin(), r.end(), out);
    *size_out = (int64_t)r.size();
    return SPARTA_OK;
    SPARTA_CATCH
}

int sparta_vbs_build(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                     const int64_t* grouping, int64_t col_block_size, int64_t row_block_size, int32_t force_fixed_size,
                     sparta_vbs_host* out) {
    SPARTA_TRY
    CsrView a; a.rows = rows; a.cols = cols; a.rowptr = rowptr; a.colidx = colidx; a.vals = vals;
    int rc = vbs_build(a, grouping, col_block_size, row_block_size, force_fixed_size != 0, out);
    return rc;
    SPARTA_CATCH
}

int sparta_vbs_partition_check(const int64_t* part, int64_t n_part, int64_t rows) {
    // VBR::partition_check (src/general/vbr.cpp:108-118): 0 = valid, 1 = empty, 2 = last entry != rows, 3 = decreasing
    if (!part || n_part <= 0) return 1;
    if (part[n_part - 1] != rows) return 2;
    for (int64_t i = 1; i < n_part; i++)
        if (part[i] < part[i - 1]) return 3;
    return 0;
}

int sparta_vbs_build_partition(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                               const int64_t* row_partition, int64_t n_part, int64_t block_size, sparta_vbs_host* out) {
    SPARTA_TRY
    if (!out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build_partition: out is NULL");
    std::memset(out, 0, sizeof(*out));
    const int chk = sparta_vbs_partition_check(row_partition, n_part, rows);
    if (chk != 0 || row_partition[0] != 0)          // the reference prints "PARTITION CHECK ERROR" and carries on into undefined behaviour
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_build_partition: partition check failed with error " + std::to_string(chk != 0 ? chk : 4));
    CsrView a; a.rows = rows; a.cols = cols; a.rowptr = rowptr; a.colidx = colidx; a.vals = vals;
    // rows keep their order: grouping[i] = the block-row that holds row i (non-decreasing), built without the reference's sort
    std::vector<int64_t> grouping((size_t)rows);
    for (int64_t ib = 0; ib + 1 < n_part; ib++)
        for (int64_t i = row_partition[ib]; i < row_partition[ib + 1]; i++) grouping[(size_t)i] = ib;
    sparta_vbs_host h;
    int rc = vbs_build_hybrid(a, grouping.data(), block_size, 0, false, 0.0, 32, &h, nullptr, true);
    if (rc != SPARTA_OK) return rc;
    // block-rows of height 0 (repeated partition entries) exist in the reference's arrays with nzcount 0: put them back
    const int64_t block_rows = n_part - 1;
    if (h.block_rows != block_rows) {
        int64_t* rp = (int64_t*)std::malloc(sizeof(int64_t) * (size_t)(block_rows + 1));
        int64_t* nz = (int64_t*)std::malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(block_rows, 1));
        if (!rp || !nz) { std::free(rp); std::free(nz); sparta_vbs_host_free(&h); return fail(SPARTA_ERR_ALLOC, "sparta_vbs_build_partition: out of host memory"); }
        int64_t src = 0;
        for (int64_t ib = 0; ib < block_rows; ib++) {
            rp[ib] = row_partition[ib];
            nz[ib] = row_partition[ib + 1] > row_partition[ib] ? h.nzcount[src++] : 0;
        }
        rp[block_rows] = rows;
        std::free(h.row_part); std::free(h.nzcount);
        h.row_part = rp; h.nzcount = nz; h.block_rows = block_rows;
    }
    *out = h;
    return SPARTA_OK;
    SPARTA_CATCH
}

int sparta_blocking_info(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const int64_t* grouping,
                         int64_t col_block_size, int64_t* info_out, float* avg_height_out) {
    SPARTA_TRY
    CsrView a; a.rows = rows; a.cols = cols; a.rowptr = rowptr; a.colidx = colidx;
    return blocking_info(a, grouping, col_block_size, info_out, avg_height_out);
    SPARTA_CATCH
}

}  // extern "C"
