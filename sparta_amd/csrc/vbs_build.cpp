// vbs_build.cpp -- host-side VBS ("VBR") builder.
//
// Produces exactly the arrays of the reference's VBR::fill_from_CSR_inplace
// (src/general/vbr.cpp:135-237): row_part, nzcount, jab, mab (column-major h x w blocks, blocks of a
// block-row consecutive), including its force_fixed_size padding rules (:143-148).
// Own design: two passes over the block-rows (count, then fill) with a stamped column-block -> slot
// table, O(nnz + touched blocks * log) instead of the reference's std::count over a bit-vector per
// nonzero (:222, O(nnz * block_cols)); both passes run on all host cores (block-rows are independent).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "host_core.hpp"

namespace sparta {
namespace {

template <typename F>
void parallel_for_blocks(int64_t n, F&& f) {
    unsigned hw = std::thread::hardware_concurrency();
    int64_t nt = std::max<int64_t>(1, std::min<int64_t>((int64_t)(hw ? hw : 1), n / 64));
    if (const char* e = std::getenv("SPARTA_HOST_THREADS")) nt = std::max<int64_t>(1, std::min<int64_t>(atoll(e), std::max<int64_t>(n, 1)));
    if (nt <= 1) { f(0, n, 0); return; }
    std::vector<std::thread> th;
    int64_t chunk = (n + nt - 1) / nt;
    for (int64_t t = 0; t < nt; t++) {
        int64_t lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([&f, lo, hi, t] { f(lo, hi, (int)t); });
    }
    for (auto& x : th) x.join();
}

// distinct column blocks touched by reordered rows [r0, r1), ascending
struct BlockCollector {
    std::vector<int32_t> stamp;   // per column block: last block-row that touched it (+1)
    std::vector<int32_t> slot;    // per column block: position inside the current block-row
    std::vector<int32_t> touched;
    explicit BlockCollector(int64_t block_cols) : stamp((size_t)block_cols, 0), slot((size_t)block_cols, 0) {}

    void collect(const CsrView& a, const int64_t* perm, int64_t r0, int64_t r1, int64_t w, int32_t tag) {
        touched.clear();
        for (int64_t r = r0; r < r1; r++) {
            int64_t i = perm[r];
            if (i >= a.rows) continue;                             // padded rows (vbr.cpp:185-186)
            const int32_t* cj = a.row(i);
            int64_t n = a.nnz_of(i);
            int64_t last = -1;
            for (int64_t k = 0; k < n; k++) {
                int64_t jb = (int64_t)cj[k] / w;
                if (jb == last) continue;
                last = jb;
                if (stamp[(size_t)jb] != tag) { stamp[(size_t)jb] = tag; touched.push_back((int32_t)jb); }
            }
        }
        std::sort(touched.begin(), touched.end());                 // jab is ascending (vbr.cpp:195-198)
        for (size_t s = 0; s < touched.size(); s++) slot[(size_t)touched[s]] = (int32_t)s;
    }
};

}  // namespace

int vbs_build(const CsrView& a, const int64_t* grouping, int64_t w, int64_t row_block_size, bool force_fixed,
              sparta_vbs_host* out) {
    return vbs_build_hybrid(a, grouping, w, row_block_size, force_fixed, 0.0, 32, out, nullptr, false);
}

// The same builder, optionally "hybrid" (sp != nullptr, K > 0): block-rows whose blocks would hold fewer than K nonzeros per
// MFMA step (one <=32-row tile x kdep columns of a block) are NOT materialised as dense blocks -- `out` gets nzcount = 0 for
// them -- and come back in `sp` as rows of (column, value) in reordered row order, for the device's sparse-row path
// (sparta_vbs_create_from_csr).  A clustered power-law matrix is 98 % zeros inside its blocks: the dense image of a
// 20 M-nonzero R-MAT matrix is 4 GB, of a 124 M-nonzero one 25 GB, all of it skipped by the kernels that then run.
int vbs_build_hybrid(const CsrView& a, const int64_t* grouping, int64_t w, int64_t row_block_size, bool force_fixed,
                     double K, int64_t kdep, sparta_vbs_host* out, HybridSparse* sp, bool keep_order) {
    if (!out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: out is NULL");
    std::memset(out, 0, sizeof(*out));
    if (w <= 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: col_block_size must be > 0");
    if (force_fixed && row_block_size <= 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: force_fixed_size needs row_block_size > 0");
    if (a.rows <= 0 || a.cols <= 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: empty matrix");
    if (!grouping) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: grouping is NULL");
    const bool hybrid = sp != nullptr && K > 0.0;
    if (int rc = validate_csr(a, hybrid)) return rc;                      // the sparse rows are taken as they are: ascending, no duplicates

    std::vector<int64_t> part = get_partition(grouping, a.rows);          // vbr.cpp:139
    std::vector<int64_t> perm = get_permutation(grouping, a.rows);        // vbr.cpp:140
    if (keep_order) {
        // the rows stay where they are (the reference's permutation is an UNSTABLE sort by group id: it shuffles rows inside a group
        // even when the groups are already contiguous); needs a non-decreasing grouping.  Used for A^T, whose rows are columns of C.
        for (int64_t i = 1; i < a.rows; i++)
            if (grouping[i] < grouping[i - 1]) return fail(SPARTA_ERR_INVALID, "vbs_build: keep_order needs a non-decreasing grouping");
        for (int64_t i = 0; i < a.rows; i++) perm[(size_t)i] = i;
    }

    int64_t rows = a.rows, cols = a.cols;
    if (force_fixed) {                                                    // vbr.cpp:143-148
        rows = ((a.rows - 1) / row_block_size + 1) * row_block_size;
        cols = ((a.cols - 1) / w + 1) * w;
        part.back() = rows;
        for (int64_t i = (int64_t)perm.size(); i < rows; i++) perm.push_back(i);
    }
    const int64_t block_cols = (cols - 1) / w + 1;                        // vbr.cpp:156
    const int64_t block_rows = (int64_t)part.size() - 1;
    if (block_cols > INT32_MAX) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_build: more than 2^31 column blocks");

    out->rows = rows; out->cols = cols; out->block_rows = block_rows; out->block_cols = block_cols; out->block_col_size = w;
    out->row_part = (int64_t*)std::malloc(sizeof(int64_t) * (size_t)(block_rows + 1));
    out->nzcount = (int64_t*)std::calloc((size_t)std::max<int64_t>(block_rows, 1), sizeof(int64_t));
    if (!out->row_part || !out->nzcount) { sparta_vbs_host_free(out); return fail(SPARTA_ERR_ALLOC, "sparta_vbs_build: out of host memory"); }
    std::copy(part.begin(), part.end(), out->row_part);

    // pass 1: number of nonzero blocks per block-row
    parallel_for_blocks(block_rows, [&](int64_t lo, int64_t hi, int) {
        BlockCollector bc(block_cols);
        for (int64_t ib = lo; ib < hi; ib++) {
            bc.collect(a, perm.data(), part[(size_t)ib], part[(size_t)ib + 1], w, (int32_t)(ib - lo + 1));
            out->nzcount[ib] = (int64_t)bc.touched.size();
        }
    });

    // hybrid: decide per block-row, empty the dense image of the sparse ones, collect their rows
    if (hybrid) {
        sp->flag.assign((size_t)block_rows, 0);
        sp->rowptr.assign(1, 0);
        sp->col.clear(); sp->val.clear(); sp->crow.clear();
        // which block-rows qualify, and how many MFMA steps they would cost as tiles: the sparse-row kernels are extra launches behind the
        // MFMA launch (10-15 us of launch and dependency measured on a banded matrix with 119 such rows), so a handful of nearly empty
        // block-rows stays with the tiles (sparta_sparse_min_steps(): 4096 steps ~ 8 per worker ~ 10 us)
        std::vector<uint8_t> qualifies((size_t)block_rows, 0);
        double steps_saved = 0.0;
        for (int64_t ib = 0; ib < block_rows; ib++) {
            const int64_t r0 = part[(size_t)ib], r1 = part[(size_t)ib + 1], h = r1 - r0, nb = out->nzcount[ib];
            if (h <= 0 || nb <= 0) continue;
            int64_t nnz = 0;
            for (int64_t r = r0; r < r1; r++) {
                const int64_t i = perm[(size_t)r];
                if (i >= a.rows) continue;
                const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
                const int64_t n = a.nnz_of(i);
                if (!v) { nnz += n; continue; }
                for (int64_t k = 0; k < n; k++) nnz += v[k] != 0.0f;
            }
            const double steps_br = (double)nb * (double)((w + kdep - 1) / kdep) * (double)((h + 31) / 32);
            if (!((double)nnz < K * steps_br)) continue;
            qualifies[(size_t)ib] = 1;
            steps_saved += steps_br;
        }
        if (steps_saved < (double)sparta::sparse_min_steps()) std::fill(qualifies.begin(), qualifies.end(), 0);
        for (int64_t ib = 0; ib < block_rows; ib++) {
            if (!qualifies[(size_t)ib]) continue;
            const int64_t r0 = part[(size_t)ib], r1 = part[(size_t)ib + 1];
            sp->flag[(size_t)ib] = 1;
            out->nzcount[ib] = 0;
            for (int64_t r = r0; r < r1; r++) {
                const int64_t i = perm[(size_t)r];
                if (i < a.rows) {
                    const int32_t* cj = a.row(i);
                    const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
                    const int64_t n = a.nnz_of(i);
                    for (int64_t k = 0; k < n; k++) {
                        const float x = v ? v[k] : 1.0f;                  // pattern-only matrices store 1 (vbr.cpp:217)
                        if (x != 0.0f) { sp->col.push_back(cj[k]); sp->val.push_back(x); }
                    }
                }
                sp->crow.push_back((int32_t)r);                           // padded rows too: they are rows of C
                sp->rowptr.push_back((int64_t)sp->col.size());
            }
        }
    }

    // offsets
    std::vector<int64_t> jab_off((size_t)block_rows + 1, 0), mab_off((size_t)block_rows + 1, 0);
    for (int64_t ib = 0; ib < block_rows; ib++) {
        int64_t h = part[(size_t)ib + 1] - part[(size_t)ib];
        jab_off[(size_t)ib + 1] = jab_off[(size_t)ib] + out->nzcount[ib];
        mab_off[(size_t)ib + 1] = mab_off[(size_t)ib] + out->nzcount[ib] * h * w;
    }
    out->nblocks = jab_off[(size_t)block_rows];
    out->nztot = mab_off[(size_t)block_rows];
    out->jab = (int64_t*)std::malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(out->nblocks, 1));
    out->mab = (float*)std::calloc((size_t)std::max<int64_t>(out->nztot, 1), sizeof(float));   // zero-filled (vbr.cpp:206)
    if (!out->jab || !out->mab) { sparta_vbs_host_free(out); return fail(SPARTA_ERR_ALLOC, "sparta_vbs_build: out of host memory (nztot = " + std::to_string(out->nztot) + ")"); }

    // pass 2: jab + scatter the values, column-major inside each block (vbr.cpp:224)
    parallel_for_blocks(block_rows, [&](int64_t lo, int64_t hi, int) {
        BlockCollector bc(block_cols);
        for (int64_t ib = lo; ib < hi; ib++) {
            const int64_t r0 = part[(size_t)ib], r1 = part[(size_t)ib + 1], h = r1 - r0;
            if (hybrid && sp->flag[(size_t)ib]) continue;                 // not materialised
            bc.collect(a, perm.data(), r0, r1, w, (int32_t)(ib - lo + 1));
            int64_t* jab = out->jab + jab_off[(size_t)ib];
            for (size_t s = 0; s < bc.touched.size(); s++) jab[s] = bc.touched[s];
            float* base = out->mab + mab_off[(size_t)ib];
            for (int64_t r = r0; r < r1; r++) {
                int64_t i = perm[(size_t)r];
                if (i >= a.rows) continue;                                // vbr.cpp:211-212
                const int32_t* cj = a.row(i);
                const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
                int64_t n = a.nnz_of(i);
                for (int64_t k = 0; k < n; k++) {
                    int64_t j = cj[k];
                    int64_t s = bc.slot[(size_t)(j / w)];
                    // pattern-only matrices store 1 (vbr.cpp:217); duplicates: last one wins (:226)
                    base[s * h * w + h * (j % w) + (r - r0)] = v ? v[k] : 1.0f;
                }
            }
        }
    });
    return SPARTA_OK;
}

// BlockingEngine::CollectBlockingInfo (src/general/blocking.cpp:576-631).  Works on the UNPADDED
// grouping, and subtracts the zero-padding of the last (narrower) block column (:624-627).
int blocking_info(const CsrView& a, const int64_t* grouping, int64_t w, int64_t* info_out, float* avg_height_out) {
    if (w <= 0 || !grouping || !info_out) return fail(SPARTA_ERR_INVALID, "sparta_blocking_info: bad argument");
    if (int rc = validate_csr(a, false)) return rc;
    std::vector<int64_t> part = get_partition(grouping, a.rows);
    std::vector<int64_t> perm = get_permutation(grouping, a.rows);
    const int64_t block_cols = (a.cols + w - 1) / w;                      // ceil (:589)
    const int64_t block_rows = (int64_t)part.size() - 1;
    int64_t nzcount = 0, nzblocks = 0, longest = 0, total_height = 0;
    BlockCollector bc(block_cols);
    for (int64_t ib = 0; ib < block_rows; ib++) {
        const int64_t h = part[(size_t)ib + 1] - part[(size_t)ib];
        bc.collect(a, perm.data(), part[(size_t)ib], part[(size_t)ib + 1], w, (int32_t)(ib + 1));
        const int64_t nb = (int64_t)bc.touched.size();
        longest = std::max(longest, nb);
        nzcount += nb * w * h;
        nzblocks += nb;
        total_height += nb * h;
        if (a.cols % w != 0 && nb > 0 && bc.touched.back() == block_cols - 1) nzcount -= h * (w - a.cols % w);
    }
    info_out[0] = nzcount; info_out[1] = nzblocks; info_out[2] = longest;
    if (avg_height_out) *avg_height_out = (float)total_height / (float)nzblocks;   // :630
    return SPARTA_OK;
}

}  // namespace sparta

extern "C" void sparta_vbs_host_free(sparta_vbs_host* v) {
    if (!v) return;
    std::free(v->row_part); std::free(v->nzcount); std::free(v->jab); std::free(v->mab);
    std::memset(v, 0, sizeof(*v));
}
