// vbs_build.cpp -- host-side VBS ("VBR") builder.
//
// Produces exactly the arrays of the reference's VBR::fill_from_CSR_inplace
// (src/general/vbr.cpp:135-237): row_part, nzcount, jab, mab (column-major h x w blocks, blocks of a
// block-row consecutive), including its force_fixed_size padding rules (:143-148).
// Own design: two passes over the block-rows (count, then fill) with a stamped column-block -> slot
// table, O(nnz + touched blocks * log) instead of the reference's std::count over a bit-vector per
// nonzero (:222, O(nnz * block_cols)); both passes run on all host cores (block-rows are independent).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "host_core.hpp"

namespace sparta {
namespace {

// distinct column blocks touched by reordered rows [r0, r1), ascending
struct BlockCollector {
    std::vector<int32_t> stamp;   // per column block: last block-row that touched it (+1)
    std::vector<int32_t> slot;    // per column block: position inside the current block-row
    std::vector<int32_t> touched;
    std::vector<int32_t> count;   // per column block: stored nonzeros of the current block-row (valid for touched blocks; count_nnz = true)
    bool count_nnz = false;
    bool ascending = true;        // every row of the last collect() had its columns in non-decreasing order
    int64_t entries = 0;          // ... and this many entries in all
    int32_t tag_counter = 0;
    // a fresh tag for the next block-row (a collector serves many block-rows, in any order)
    int32_t next_tag() {
        if (++tag_counter == INT32_MAX) { std::fill(stamp.begin(), stamp.end(), 0); tag_counter = 1; }
        return tag_counter;
    }
    explicit BlockCollector(int64_t block_cols, bool counting = false)
        : stamp((size_t)block_cols, 0), slot((size_t)block_cols, 0), count(counting ? (size_t)block_cols : 0, 0), count_nnz(counting) {}

    void collect(const CsrView& a, const int64_t* perm, int64_t r0, int64_t r1, int64_t w, int32_t tag) {
        touched.clear();
        ascending = true;
        entries = 0;
        for (int64_t r = r0; r < r1; r++) {
            int64_t i = perm[r];
            if (i >= a.rows) continue;                             // padded rows (vbr.cpp:185-186)
            const int32_t* cj = a.row(i);
            int64_t n = a.nnz_of(i);
            int64_t last = -1;
            entries += n;
            const float* v = count_nnz && a.vals ? a.vals + a.rowptr[i] : nullptr;
            for (int64_t k = 0; k < n; k++) {
                if (k > 0 && cj[k] < cj[k - 1]) ascending = false;
                int64_t jb = (int64_t)cj[k] / w;
                if (jb != last) {
                    last = jb;
                    if (stamp[(size_t)jb] != tag) { stamp[(size_t)jb] = tag; touched.push_back((int32_t)jb); if (count_nnz) count[(size_t)jb] = 0; }
                }
                if (count_nnz) count[(size_t)jb] += !v || v[k] != 0.0f;
            }
        }
        std::sort(touched.begin(), touched.end());                 // jab is ascending (vbr.cpp:195-198)
        for (size_t s = 0; s < touched.size(); s++) slot[(size_t)touched[s]] = (int32_t)s;
    }
};

// per-COLUMN nonzero counts of a run of reordered rows: a "part" of a column-compacted tile (UnionPlanHost) -- the rows of one block-row inside one tile
struct ColCounter {
    std::vector<int32_t> cnt;       // per column: stored nonzeros of the counted rows; all zero between uses
    std::vector<int32_t> touched;   // the columns with cnt > 0, in order of first appearance
    explicit ColCounter(int64_t cols) : cnt((size_t)cols, 0) {}
    void count(const CsrView& a, const int64_t* perm, int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; r++) {
            const int64_t i = perm[r];
            if (i >= a.rows) continue;
            const int32_t* cj = a.row(i);
            const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
            const int64_t n = a.nnz_of(i);
            for (int64_t k = 0; k < n; k++) {
                if (v && v[k] == 0.0f) continue;
                if (cnt[(size_t)cj[k]]++ == 0) touched.push_back(cj[k]);
            }
        }
    }
    void reset() { for (int32_t c : touched) cnt[(size_t)c] = 0; touched.clear(); }
};

// the parts of a block-row of h rows that becomes column-compacted tiles: chunks of 64 rows, each ONE tile; a tile costs its rows rounded up to the kernel's MFMA row tile
// (16 rows for fp32 handles, 32 for 16-bit ones: HybridSparse::union_gran)
inline int64_t union_parts(int64_t h) { return (h + 63) / 64; }
inline int64_t union_part_rows(int64_t h, int64_t q) { return std::min<int64_t>(64, h - 64 * q); }

}  // namespace

// what a column of a column-compacted tile costs, in nonzeros of the sparse-row path: a 32-deep step of a 32-row tile is worth K_union of them (SPARTA_UNION_K, default 36:
// the 24 of a w-wide block step + the list entry, the gathered row of B and 128 bytes of A per column); a column is kept in the tile when its rows hold at least that
// many nonzeros (never fewer than 2: a column one row uses is a sparse-row entry)
double union_tile_units(int64_t rows, int gran) { return (double)((rows + gran - 1) / gran * gran) / 32.0; }
double union_col_cost(double units) {
    static const double K_union = [] { const char* e = std::getenv("SPARTA_UNION_K"); return e ? std::max(1.0, atof(e)) : 36.0; }();
    return K_union * (2.0 * units + 1.0) / 3.0 / 32.0;       // (a third of a 32-row tile's column is the gather -- list entry, row of B -- whatever the tile's height)
}
// (1.2 x: a column that merely breaks even -- two nonzeros in a 64-row tile -- stays out: R-MAT 2^16 under the reference's tau 0.001, 1024-row clusters: 4.83 ms with such columns, 4.74 without)
int32_t union_min_count(double units) { return std::max<int32_t>(2, (int32_t)std::ceil(1.2 * union_col_cost(units))); }
int32_t union_tail_cap() {
    static const int32_t cap = [] { const char* e = std::getenv("SPARTA_UNION_TAIL"); return e ? std::max(0, std::min(31, atoi(e))) : 16; }();
    return cap;
}

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
                     double K, int64_t kdep, sparta_vbs_host* out, HybridSparse* sp, bool keep_order, HybridStats* stats_only) {
    if (!out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: out is NULL");
    std::memset(out, 0, sizeof(*out));
    if (w <= 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: col_block_size must be > 0");
    if (force_fixed && row_block_size <= 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: force_fixed_size needs row_block_size > 0");
    if (a.rows <= 0 || a.cols <= 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: empty matrix");
    if (!grouping) return fail(SPARTA_ERR_INVALID, "sparta_vbs_build: grouping is NULL");
    const bool hybrid = sp != nullptr && K > 0.0;
    if (int rc = validate_csr(a, hybrid)) return rc;                      // the sparse rows are taken as they are: ascending, no duplicates

    BuildTrace trace0("vbs_build");
    std::vector<int64_t> part = get_partition(grouping, a.rows);          // vbr.cpp:139
    std::vector<int64_t> perm = get_permutation(grouping, a.rows);        // vbr.cpp:140
    trace0.lap("partition + permutation");
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

    // Hybrid decision, per block-row, from the nonzeros of each of its blocks (cost unit: one nonzero on the sparse-row path; an MFMA step is
    // worth K of them): a block is WELL FILLED when it holds at least K nonzeros per step it costs.  A block-row is
    //   0  dense: all blocks as tiles                                      cost nb * spb * K
    //   1  sparse: no tile at all, every nonzero a sparse-row entry         cost nnz
    //   2  mixed: the well-filled blocks as tiles, the nonzeros of the rest as sparse rows that ADD to what the tiles stored
    //                                                                       cost n_dense * spb * K + nnz_rest + h * (re-reading a row of C)
    // whichever is cheapest (ties: fewer kernels).  Mode 2 is what dense power-law matrices need: the hub columns fill their blocks, the tail does
    // not, and a whole-block-row decision sends the hub's nonzeros through the gather one by one.
    const double rmw_cost = 4.0 / (sp ? sp->esz : 4.0) + 4.0;
    // a block is "well filled" against a higher bar than a whole block-row (SPARTA_SPARSE_K_BLOCK nonzeros per step, default 60): the rows of B
    // its nonzeros would gather are the hub rows -- L2-resident, where the measured break-even of tiles against sparse rows is ~60 nonzeros
    // per step, not the ~15-24 of rows of B that come from HBM (scripts/sparse_k_sweep.py)
    const double K_block = [] { const char* e = std::getenv("SPARTA_SPARSE_K_BLOCK"); return e ? atof(e) : 60.0; }();
    auto spb_of = [&](int64_t h) { return (double)((w + kdep - 1) / kdep) * (double)((h + 31) / 32); };
    struct RowMode { uint8_t mode; int64_t n_dense; double saved; double cost, c_dense; };
    auto decide = [&](const BlockCollector& bc, int64_t h) {
        const int64_t nb = (int64_t)bc.touched.size();
        RowMode r{0, nb, 0.0, 0.0, 0.0};
        if (!hybrid || nb == 0 || h <= 0) return r;
        const double spb = spb_of(h), kb = K * spb, kbb = K_block * spb;
        int64_t n_dense = 0;
        double nnz = 0.0, nnz_rest = 0.0;
        for (int32_t jb : bc.touched) {
            const double c = (double)bc.count[(size_t)jb];
            nnz += c;
            if (c >= kbb) n_dense++; else nnz_rest += c;
        }
        const double c_dense = (double)nb * kb, c_sparse = nnz, c_mixed = (double)n_dense * kb + nnz_rest + (double)h * rmw_cost;
        r.cost = r.c_dense = c_dense;
        if (c_sparse < c_dense && c_sparse <= c_mixed) { r.mode = 1; r.n_dense = 0; r.saved = (c_dense - c_sparse) / K; r.cost = c_sparse; }
        else if (n_dense > 0 && n_dense < nb && c_mixed < c_dense) { r.mode = 2; r.n_dense = n_dense; r.saved = (c_dense - c_mixed) / K; r.cost = c_mixed; }
        return r;
    };
    // mode 3 (sp->want_union): the block-row as column-compacted tiles -- per part (chunk of <= 64 rows) the columns at least union_min_count of its rows use, as one dense
    // tile + the column list; the nonzeros of the other columns are sparse rows that ADD (as in mode 2).   cost  sum_parts |U| * union_col_cost + nnz_rest + rows_rest * rmw
    // Chosen when that is below SPARTA_UNION_MARGIN (default 0.8) x the best of the three modes above.
    const bool want_union = hybrid && sp->want_union && [] { const char* e = std::getenv("SPARTA_UNION"); return !(e && atoi(e) == 0); }();
    const double union_margin = [] { const char* e = std::getenv("SPARTA_UNION_MARGIN"); return e ? atof(e) : 0.8; }();
    struct UnionEval { double cost; int64_t rows_rest, ent_rest, nnz_in, tail_ent; std::vector<int32_t> nu, te; };    // nu / te: per part, list entries / tail entries per row
    const int64_t tail_cap = union_tail_cap();
    auto eval_union = [&](ColCounter& cc, int64_t r0, int64_t r1, int64_t cap) {
        UnionEval u{0.0, 0, 0, 0, 0, {}, {}};
        const int64_t h = r1 - r0;
        for (int64_t q = 0; q < union_parts(h); q++) {
            const int64_t p0 = r0 + 64 * q, p1 = p0 + union_part_rows(h, q);
            const double units = union_tile_units(p1 - p0, sp->union_gran);
            const int32_t cmin = union_min_count(units);
            cc.count(a, perm.data(), p0, p1);
            int64_t nu = 0;
            for (int32_t c : cc.touched) if (cc.cnt[(size_t)c] >= cmin) { nu++; u.nnz_in += cc.cnt[(size_t)c]; }
            u.nu.push_back((int32_t)nu);
            u.cost += (double)nu * union_col_cost(units);
            int64_t te = 0;
            for (int64_t rr = p0; rr < p1; rr++) {                   // the rows' nonzeros in thinly used columns
                const int64_t i = perm[(size_t)rr];
                if (i >= a.rows) continue;
                const int32_t* cj = a.row(i);
                const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
                const int64_t n = a.nnz_of(i);
                int64_t mine = 0;
                for (int64_t k = 0; k < n; k++) mine += (!v || v[k] != 0.0f) && cc.cnt[(size_t)cj[k]] < cmin;
                const int64_t in_tail = std::min(mine, cap);                      // the first `cap` of them ride in the tile's tail, the others are sparse-row entries
                u.tail_ent += in_tail; u.ent_rest += mine - in_tail; u.rows_rest += mine > in_tail;
                te = std::max(te, in_tail);
            }
            u.te.push_back((int32_t)te);
            cc.reset();
        }
        u.nnz_in += u.tail_ent;
        u.cost += (double)u.ent_rest + (double)u.rows_rest * rmw_cost + 0.8 * (double)u.tail_ent;     // (a tail entry: the same row of B as a sparse-row entry, no row of C re-read)
        return u;
    };

    BuildTrace trace("vbs_build");
    // pass 1: blocks per block-row (+ the hybrid mode, and what the block-row would put on the sparse-row path: rows, entries)
    std::vector<uint8_t> mode((size_t)block_rows, 0);
    std::vector<double> saved_row((size_t)block_rows, 0.0);           // MFMA steps the chosen mode saves against "all tiles"
    std::vector<int64_t> sp_rows_of(hybrid ? (size_t)block_rows : 0, 0), sp_ent_of(hybrid ? (size_t)block_rows : 0, 0), nb_all(hybrid ? (size_t)block_rows : 0, 0);
    std::vector<int64_t> nnz_all(hybrid ? (size_t)block_rows : 0, 0);  // stored nonzeros of the block-row (all blocks)
    std::vector<std::vector<int32_t>> nu_parts(want_union ? (size_t)block_rows : 0);   // mode 3: columns kept per part
    std::vector<std::vector<int32_t>> te_parts(want_union ? (size_t)block_rows : 0);   // ... and tail entries per row of the part's tile
    std::vector<int64_t> union_nnz_of(want_union ? (size_t)block_rows : 0, 0);
    const int64_t grain = std::max<int64_t>(1, std::min<int64_t>(64, block_rows / (8 * (int64_t)host_threads()) + 1));
    // (a per-column counter is `cols` integers -- 33 MB on an 8 M-column graph: one per THREAD, made when a block-row first asks for it, not one per chunk of block-rows)
    std::vector<std::unique_ptr<ColCounter>> col_counters((size_t)std::max(1, host_threads()));
    auto counter_of = [&](int t) -> ColCounter& {
        std::unique_ptr<ColCounter>& c = col_counters[(size_t)t % col_counters.size()];
        if (!c) c.reset(new ColCounter(cols));
        return *c;
    };
    parallel_for_dynamic(block_rows, grain, [&](int64_t lo, int64_t hi, int tid) {
        BlockCollector bc(block_cols, hybrid);
        for (int64_t ib = lo; ib < hi; ib++) {
            const int64_t r0 = part[(size_t)ib], r1 = part[(size_t)ib + 1], h = r1 - r0;
            bc.collect(a, perm.data(), r0, r1, w, bc.next_tag());
            const RowMode r = decide(bc, h);
            mode[(size_t)ib] = r.mode;
            out->nzcount[ib] = r.n_dense;
            saved_row[(size_t)ib] = r.saved;
            if (!hybrid) continue;
            nb_all[(size_t)ib] = (int64_t)bc.touched.size();
            for (int32_t jb : bc.touched) nnz_all[(size_t)ib] += bc.count[(size_t)jb];
            // column-compacted tiles?  Only where they CAN win: every column holds at most h nonzeros, so the tiles cost at least nnz / h columns
            if (want_union && h >= 2 && !bc.touched.empty()) {
                const double lower = (double)nnz_all[(size_t)ib] / (double)h * union_col_cost(union_tile_units(std::min<int64_t>(h, 64), sp->union_gran));
                if (lower < union_margin * r.cost) {
                    UnionEval u = eval_union(counter_of(tid), r0, r1, tail_cap);
                    // (... and only where the LISTS carry the block-row: tiles that are mostly tails are a slower sparse-row kernel)
                    if (2 * (u.nnz_in - u.tail_ent) >= nnz_all[(size_t)ib] && u.cost < union_margin * r.cost) {
                        mode[(size_t)ib] = 3;
                        out->nzcount[ib] = 0;
                        saved_row[(size_t)ib] = (r.c_dense - u.cost) / K;
                        sp_rows_of[(size_t)ib] = u.rows_rest; sp_ent_of[(size_t)ib] = u.ent_rest;
                        nu_parts[(size_t)ib].swap(u.nu);
                        te_parts[(size_t)ib].swap(u.te);
                        union_nnz_of[(size_t)ib] = u.nnz_in;
                        continue;
                    }
                }
            }
            if (r.mode == 1) {                                            // every row of the block-row is a sparse row (padded and empty ones too)
                int64_t ent = 0;
                for (int32_t jb : bc.touched) ent += bc.count[(size_t)jb];
                sp_rows_of[(size_t)ib] = h; sp_ent_of[(size_t)ib] = ent;
            } else if (r.mode == 2) {                                     // the nonzeros of the blocks that are not well filled; rows without one are left out
                const double kbb = K_block * spb_of(h);
                int64_t ent = 0, nrows = 0;
                for (int64_t rr = r0; rr < r1; rr++) {
                    const int64_t i = perm[(size_t)rr];
                    if (i >= a.rows) continue;
                    const int32_t* cj = a.row(i);
                    const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
                    const int64_t n = a.nnz_of(i);
                    int64_t mine = 0;
                    for (int64_t k = 0; k < n; k++)
                        mine += (!v || v[k] != 0.0f) && (double)bc.count[(size_t)(cj[k] / w)] < kbb;
                    ent += mine; nrows += mine > 0;
                }
                sp_rows_of[(size_t)ib] = nrows; sp_ent_of[(size_t)ib] = ent;
            }
        }
    });
    trace.lap("pass 1 (count + decide)");

    // hybrid: a handful of nearly empty blocks stays with the tiles -- the sparse-row kernels are extra launches behind the MFMA launch
    // (10-15 us of launch and dependency measured on a banded matrix with 119 such rows; sparse_min_steps(): 4096 steps ~ 8 per worker ~ 10 us);
    // then the sparse rows are collected in row order (offsets from the counts of pass 1, filled on all threads)
    if (hybrid) {
        double steps_saved = 0.0;
        for (double x : saved_row) steps_saved += x;
        if (steps_saved < (double)sparse_min_steps()) {
            for (int64_t ib = 0; ib < block_rows; ib++) {
                if (!mode[(size_t)ib]) continue;
                mode[(size_t)ib] = 0;
                out->nzcount[ib] = nb_all[(size_t)ib];
                sp_rows_of[(size_t)ib] = sp_ent_of[(size_t)ib] = 0;
                if (want_union) { nu_parts[(size_t)ib].clear(); te_parts[(size_t)ib].clear(); union_nnz_of[(size_t)ib] = 0; }
            }
        }
        // steps / stored elements of the column-compacted tiles of block-row ib (mode 3)
        auto union_steps_of = [&](int64_t ib) {
            double st = 0.0;
            const int64_t h = part[(size_t)ib + 1] - part[(size_t)ib];
            for (size_t q = 0; q < nu_parts[(size_t)ib].size(); q++) st += (double)((nu_parts[(size_t)ib][q] + 31) / 32) * union_tile_units(union_part_rows(h, (int64_t)q), sp->union_gran);
            return st;
        };
        auto union_area_of = [&](int64_t ib) {
            double ar = 0.0;
            const int64_t h = part[(size_t)ib + 1] - part[(size_t)ib];
            for (size_t q = 0; q < nu_parts[(size_t)ib].size(); q++) ar += (double)nu_parts[(size_t)ib][q] * (double)union_part_rows(h, (int64_t)q);
            return ar;
        };
        // SMALL matrices: every launch of a product costs ~5-10 us whatever it does, and the MFMA part needs up to five (panel tail copy, one per tile
        // type, fix-up, zero fill) where the sparse-row kernels need the ones they run anyway.  When the tiles of the whole matrix hold fewer nonzeros than
        // their steps + those launches are worth (a launch ~ SPARTA_LAUNCH_NNZ = 50 000 gathered nonzeros ~ 5 us), EVERYTHING goes to the sparse rows:
        // ca-HepPh (118 k nonzeros, 11 tile blocks): 110 -> ~40 us per product, bcsstk18 (80 k nonzeros in 3002 thin blocks): 48 -> ~25 us.
        {
            const double launch_nnz = [] { const char* e = std::getenv("SPARTA_LAUNCH_NNZ"); return e ? atof(e) : 50000.0; }();
            double tile_steps = 0.0, tile_nnz = 0.0, tile_area = 0.0;
            for (int64_t ib = 0; ib < block_rows; ib++) {
                if (mode[(size_t)ib] == 1) continue;
                tile_nnz += (double)(nnz_all[(size_t)ib] - sp_ent_of[(size_t)ib]);
                if (mode[(size_t)ib] == 3) { tile_steps += union_steps_of(ib); tile_area += union_area_of(ib); continue; }
                tile_steps += (double)out->nzcount[ib] * spb_of(part[(size_t)ib + 1] - part[(size_t)ib]);
                tile_area += (double)out->nzcount[ib] * (double)(part[(size_t)ib + 1] - part[(size_t)ib]) * (double)w;
            }
            // ... and only when the tiles are not DENSE (fill below a half): a small dense matrix stays on the MFMA kernels (and keeps its dense image: the
            // exact-order kernel, SPARTA_SPMM_EXACT, has nothing to walk on a handle whose blocks are all sparse rows).  Half-filled hub blocks of a small graph
            // still go: ia-wikiquote (239 k nonzeros, a few well-filled blocks) 67 us all-sparse against 94 with its tiles (two more launches)
            if (tile_steps > 0.0 && tile_nnz < K * tile_steps + 3.0 * launch_nnz && tile_nnz < 0.5 * tile_area) {
                for (int64_t ib = 0; ib < block_rows; ib++) {
                    const int64_t h = part[(size_t)ib + 1] - part[(size_t)ib];
                    if (mode[(size_t)ib] == 1 || h <= 0) continue;
                    // (block-rows without blocks too: as sparse rows without entries they get their zeros from the same launch instead of a fix-up / zero-fill launch)
                    mode[(size_t)ib] = 1;
                    out->nzcount[ib] = 0;
                    sp_rows_of[(size_t)ib] = h;
                    sp_ent_of[(size_t)ib] = nnz_all[(size_t)ib];
                    if (want_union) { nu_parts[(size_t)ib].clear(); te_parts[(size_t)ib].clear(); union_nnz_of[(size_t)ib] = 0; }
                }
            }
        }
        // STRAGGLERS.  A clustered matrix leaves a few rows outside every cluster (the clustering missed them: 58 of 96 000 on the benchmark set's clustered family) -- fully
        // sparse block-rows of a row or two, 14 k nonzeros in all, which alone cost the product two more launches (segments + their reduction: 12 of 167 us).  When the handle
        // has column-compacted tiles anyway and what is left for the sparse-row kernels is that small (< 2 % of the tiles' nonzeros and < 64 k), those block-rows become tiles too,
        // with EVERY column in the list (take_all: a list entry gathers the same row of B a sparse-row entry would, and the launches are gone).
        std::vector<uint8_t> take_all(want_union ? (size_t)block_rows : 0, 0);
        std::vector<int8_t> cap_of(want_union ? (size_t)block_rows : 0, (int8_t)tail_cap);       // tail entries per row a block-row's tiles may hold
        if (want_union) {
            int64_t union_total = 0, sparse_total = 0, add_total = 0;
            for (int64_t ib = 0; ib < block_rows; ib++) {
                union_total += union_nnz_of[(size_t)ib];
                if (mode[(size_t)ib] == 1) sparse_total += sp_ent_of[(size_t)ib];
                else add_total += sp_ent_of[(size_t)ib];                   // (rows that ADD to tiles -- more than tail_cap thinly-used columns: they stay sparse rows)
            }
            const int64_t straggler_cap = [] { const char* e = std::getenv("SPARTA_UNION_STRAGGLERS"); return e ? atoll(e) : (int64_t)65536; }();      // (read per build: 0 switches the rule off)
            if (union_total > 0 && sparse_total + add_total > 0 && sparse_total + add_total <= straggler_cap && (sparse_total + add_total) * 50 <= union_total) {
                for (int64_t ib = 0; ib < block_rows; ib++) {
                    if (mode[(size_t)ib] == 3 && sp_ent_of[(size_t)ib] > 0 && tail_cap > 0 && tail_cap < 31) {
                        // ... and a tile row with more thinly-used columns than the tail holds (the rows that ADD) gets the longest tail the step record can name (31)
                        UnionEval u = eval_union(counter_of(0), part[(size_t)ib], part[(size_t)ib + 1], 31);
                        if (u.ent_rest == 0 || part[(size_t)ib + 1] - part[(size_t)ib] > 4) {
                            cap_of[(size_t)ib] = 31;
                            sp_rows_of[(size_t)ib] = u.rows_rest; sp_ent_of[(size_t)ib] = u.ent_rest;
                            nu_parts[(size_t)ib].swap(u.nu); te_parts[(size_t)ib].swap(u.te);
                            union_nnz_of[(size_t)ib] = u.nnz_in;
                            continue;
                        }
                        // (even that tail overflows in a group of two to four rows -- grouped on a few shared columns: every column in the list, as for the loners below)
                    } else if (mode[(size_t)ib] != 1) continue;
                    const int64_t r0 = part[(size_t)ib], r1 = part[(size_t)ib + 1], h = r1 - r0;
                    ColCounter& cc = counter_of(0);
                    std::vector<int32_t> nu, te;
                    int64_t nnz_in = 0;
                    for (int64_t q = 0; q < union_parts(h); q++) {
                        cc.count(a, perm.data(), r0 + 64 * q, r0 + 64 * q + union_part_rows(h, q));
                        nu.push_back((int32_t)cc.touched.size()); te.push_back(0);
                        for (int32_t c : cc.touched) nnz_in += cc.cnt[(size_t)c];
                        cc.reset();
                    }
                    mode[(size_t)ib] = 3; take_all[(size_t)ib] = 1;
                    out->nzcount[ib] = 0;
                    sp_rows_of[(size_t)ib] = 0; sp_ent_of[(size_t)ib] = 0;
                    nu_parts[(size_t)ib].swap(nu); te_parts[(size_t)ib].swap(te);
                    union_nnz_of[(size_t)ib] = nnz_in;
                }
            }
        }
        if (stats_only) {                                                  // sparta_vbs_plan_stats: the decisions are all that is wanted
            HybridStats st;
            st.block_rows = block_rows; st.rows = rows;
            for (int64_t ib = 0; ib < block_rows; ib++) {
                const int64_t h = part[(size_t)ib + 1] - part[(size_t)ib];
                st.tile_blocks += out->nzcount[ib];
                st.tile_area += out->nzcount[ib] * h * w;
                st.mfma_steps += (double)out->nzcount[ib] * spb_of(h);
                st.sparse_nnz += sp_ent_of[(size_t)ib];
                st.sparse_rows += sp_rows_of[(size_t)ib];
                if (mode[(size_t)ib] == 3) {
                    st.union_block_rows++; st.union_steps += union_steps_of(ib); st.union_nnz += union_nnz_of[(size_t)ib];
                    for (int32_t nu : nu_parts[(size_t)ib]) st.union_cols += nu;
                }
            }
            *stats_only = st;
            return SPARTA_OK;
        }
        sp->flag.assign(mode.begin(), mode.end());
        std::vector<int64_t> row_base((size_t)block_rows + 1, 0), ent_base((size_t)block_rows + 1, 0);
        for (int64_t ib = 0; ib < block_rows; ib++) {
            row_base[(size_t)ib + 1] = row_base[(size_t)ib] + sp_rows_of[(size_t)ib];
            ent_base[(size_t)ib + 1] = ent_base[(size_t)ib] + sp_ent_of[(size_t)ib];
        }
        const int64_t n_sp_rows = row_base[(size_t)block_rows], n_sp_ent = ent_base[(size_t)block_rows];
        sp->rowptr.assign((size_t)n_sp_rows + 1, 0);
        sp->col.resize((size_t)n_sp_ent); sp->val.resize((size_t)n_sp_ent);
        sp->crow.resize((size_t)n_sp_rows); sp->row_add.resize((size_t)n_sp_rows);
        parallel_for_dynamic(block_rows, grain, [&](int64_t lo, int64_t hi, int) {
            BlockCollector bc(block_cols, true);
            for (int64_t ib = lo; ib < hi; ib++) {
                if (!mode[(size_t)ib] || mode[(size_t)ib] == 3) continue;     // (mode 3: filled with its tiles, below)
                const int64_t r0 = part[(size_t)ib], r1 = part[(size_t)ib + 1], h = r1 - r0;
                const bool mixed = mode[(size_t)ib] == 2;
                double kbb = 0.0;
                if (mixed) { bc.collect(a, perm.data(), r0, r1, w, bc.next_tag()); kbb = K_block * spb_of(h); }
                int64_t t = row_base[(size_t)ib], e = ent_base[(size_t)ib];
                for (int64_t r = r0; r < r1; r++) {
                    const int64_t i = perm[(size_t)r];
                    const int64_t before = e;
                    if (i < a.rows) {
                        const int32_t* cj = a.row(i);
                        const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
                        const int64_t n = a.nnz_of(i);
                        for (int64_t k = 0; k < n; k++) {
                            const float x = v ? v[k] : 1.0f;                  // pattern-only matrices store 1 (vbr.cpp:217)
                            if (x == 0.0f || (mixed && (double)bc.count[(size_t)(cj[k] / w)] >= kbb)) continue;
                            sp->col[(size_t)e] = cj[k]; sp->val[(size_t)e] = x; e++;
                        }
                    }
                    if (mixed && e == before) continue;                       // a row of a mixed block-row without such a nonzero: the tiles wrote all of it
                    sp->crow[(size_t)t] = (int32_t)r;                         // (fully sparse block-rows: padded and empty rows too -- they are rows of C)
                    sp->row_add[(size_t)t] = mixed ? 1 : 0;
                    sp->rowptr[(size_t)t + 1] = e;
                    t++;
                }
            }
        });
        trace.lap("sparse rows (collect)");

        // ---- mode 3: the column-compacted tiles.  Every part (chunk of <= 64 rows of ONE block-row) is a tile: 33..64 rows ty 1, <= 32 rows ty 0.  (Parts of different
        // block-rows share no column by construction, so packing them into one tile would save no MFMA -- it would only make one long tile out of several short ones.)
        // Every block-row fills its tiles: ascending column list, dense values, up to tail_cap thinly-used nonzeros per row in the tile's tail, the rest as sparse rows that ADD.
        if (want_union) {
            UnionPlanHost& U = sp->uni;
            std::vector<int64_t> part_base((size_t)block_rows + 1, 0);
            for (int64_t ib = 0; ib < block_rows; ib++) part_base[(size_t)ib + 1] = part_base[(size_t)ib] + (int64_t)nu_parts[(size_t)ib].size();
            std::vector<int32_t> tile_of((size_t)part_base[(size_t)block_rows]);
            for (int64_t ib = 0; ib < block_rows; ib++) {
                if (mode[(size_t)ib] != 3) continue;
                const int64_t r0 = part[(size_t)ib], h = part[(size_t)ib + 1] - r0;
                for (size_t q = 0; q < nu_parts[(size_t)ib].size(); q++) {
                    const int64_t p0 = r0 + 64 * (int64_t)q, len = union_part_rows(h, (int64_t)q);
                    const int ty = len > 32 ? 1 : 0;
                    const int32_t nu = nu_parts[(size_t)ib][q];
                    tile_of[(size_t)part_base[(size_t)ib] + q] = (int32_t)U.tiles[ty].size();
                    const int32_t te = te_parts[(size_t)ib][q];
                    U.tiles[ty].push_back(UnionPlanHost::Tile{(int32_t)p0, (int32_t)len, (int64_t)U.cols[ty].size(), nu, te, (int64_t)U.tail_col[ty].size()});
                    U.cols[ty].resize(U.cols[ty].size() + (size_t)nu);
                    U.tail_col[ty].resize(U.tail_col[ty].size() + (size_t)te * (size_t)(32 * (ty + 1)), -1);          // (-1: slot not used yet)
                }
                U.nnz += union_nnz_of[(size_t)ib];
            }
            for (int ty = 0; ty < 2; ty++) {
                int64_t o = 0;
                U.a_off[ty].assign(U.tiles[ty].size() + 1, 0);
                for (size_t t = 0; t < U.tiles[ty].size(); t++) { U.a_off[ty][t] = o; o += (int64_t)U.tiles[ty][t].nk * 32 * (ty + 1); }
                U.a_off[ty][U.tiles[ty].size()] = o;
                U.a[ty].assign((size_t)o, 0.0f);
            }
            for (int ty = 0; ty < 2; ty++) U.tail_val[ty].assign(U.tail_col[ty].size(), 0.0f);
            parallel_for_dynamic(block_rows, grain, [&](int64_t lo, int64_t hi, int tid) {
                std::vector<int32_t> list;
                for (int64_t ib = lo; ib < hi; ib++) {
                    if (mode[(size_t)ib] != 3) continue;
                    ColCounter& cc = counter_of(tid);
                    const int64_t r0 = part[(size_t)ib], r1 = part[(size_t)ib + 1], h = r1 - r0;
                    int64_t t = row_base[(size_t)ib], e = ent_base[(size_t)ib];
                    for (size_t q = 0; q < nu_parts[(size_t)ib].size(); q++) {
                        const int64_t p0 = r0 + 64 * (int64_t)q, p1 = p0 + union_part_rows(h, (int64_t)q);
                        const int ty = p1 - p0 > 32 ? 1 : 0, mi = ty + 1;
                        const int32_t ti = tile_of[(size_t)part_base[(size_t)ib] + q];
                        const int32_t cmin = take_all[(size_t)ib] ? 1 : union_min_count(union_tile_units(p1 - p0, sp->union_gran));
                        const UnionPlanHost::Tile& tl = U.tiles[ty][(size_t)ti];
                        cc.count(a, perm.data(), p0, p1);
                        list.clear();
                        for (int32_t c : cc.touched) if (cc.cnt[(size_t)c] >= cmin) list.push_back(c);
                        std::sort(list.begin(), list.end());
                        // (cnt doubles as the position table: a kept column -> -(position + 1); the others keep their small positive count)
                        for (size_t k = 0; k < list.size(); k++) { U.cols[ty][(size_t)tl.k0 + k] = list[k]; cc.cnt[(size_t)list[k]] = -(int32_t)(k + 1); }
                        float* img = U.a[ty].data() + U.a_off[ty][(size_t)ti];
                        const int64_t ldt = 32 * mi;
                        int32_t* tc = U.tail_col[ty].data() + tl.tail0;
                        float* tv = U.tail_val[ty].data() + tl.tail0;
                        const int32_t fill_col = list.empty() ? 0 : list[0];   // an unused tail slot points at a column of the tile (any valid row of B) with value 0
                        for (int64_t r = p0; r < p1; r++) {
                            const int64_t i = perm[(size_t)r];
                            const int64_t before = e;
                            int64_t in_tail = 0;
                            if (i < a.rows) {
                                const int32_t* cj = a.row(i);
                                const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
                                const int64_t n = a.nnz_of(i);
                                for (int64_t k = 0; k < n; k++) {
                                    const float x = v ? v[k] : 1.0f;
                                    if (x == 0.0f) continue;
                                    const int32_t c = cc.cnt[(size_t)cj[k]];
                                    if (c < 0) img[(int64_t)(-c - 1) * ldt + (r - p0)] = x;
                                    else if (in_tail < tl.tail_e && in_tail < cap_of[(size_t)ib]) { tc[in_tail * ldt + (r - p0)] = cj[k]; tv[in_tail * ldt + (r - p0)] = x; in_tail++; }
                                    else { sp->col[(size_t)e] = cj[k]; sp->val[(size_t)e] = x; e++; }
                                }
                            }
                            if (e == before) continue;                            // the tile writes all of this row
                            sp->crow[(size_t)t] = (int32_t)r;
                            sp->row_add[(size_t)t] = 1;
                            sp->rowptr[(size_t)t + 1] = e;
                            t++;
                        }
                        for (int64_t x = 0; x < (int64_t)tl.tail_e * ldt; x++) if (tc[x] < 0) tc[x] = fill_col;
                        cc.reset();
                    }
                }
            });
            for (int ty = 0; ty < 2; ty++) for (float x : U.tail_val[ty]) U.tail_nnz += x != 0.0f;
            trace.lap("column-compacted tiles");
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
    trace.lap("offsets + calloc");
    parallel_for_dynamic(block_rows, grain, [&](int64_t lo, int64_t hi, int) {
        BlockCollector bc(block_cols, hybrid);
        for (int64_t ib = lo; ib < hi; ib++) {
            const int64_t r0 = part[(size_t)ib], r1 = part[(size_t)ib + 1], h = r1 - r0;
            if (mode[(size_t)ib] == 1 || mode[(size_t)ib] == 3) continue;  // not materialised as w-wide blocks
            bc.collect(a, perm.data(), r0, r1, w, bc.next_tag());
            const bool mixed = mode[(size_t)ib] == 2;
            if (mixed) {                                                  // keep the well-filled blocks only; slot = -1 marks the others
                const double kbb = K_block * spb_of(h);
                int32_t s2 = 0;
                for (size_t s = 0; s < bc.touched.size(); s++) {
                    const int32_t jb = bc.touched[s];
                    if ((double)bc.count[(size_t)jb] >= kbb) { bc.slot[(size_t)jb] = s2; bc.touched[(size_t)s2++] = jb; }
                    else bc.slot[(size_t)jb] = -1;
                }
                bc.touched.resize((size_t)s2);
            }
            int64_t* jab = out->jab + jab_off[(size_t)ib];
            for (size_t s = 0; s < bc.touched.size(); s++) jab[s] = bc.touched[s];
            float* base = out->mab + mab_off[(size_t)ib];
            // A WELL-FILLED block-row (the hub of a dense power-law part: 64 rows x 10^6 columns, 10^7 entries) row by row is one cache line fetched and written back
            // per entry -- a row walks the whole 268 MB image of the block-row, 256 bytes apart, and its neighbour comes back to every line after it has left the caches
            // (measured: 6.9 s of a 19 s build for 3e9 entries on 256 threads).  With ascending rows (a cursor per row) the same entries go in column windows whose
            // blocks stay in the core's L2: every line of the image is written once.  Same stores, same order inside every row (duplicates: still the last one wins).
            const int64_t win_cols = std::max<int64_t>(w, (65536 / std::max<int64_t>(h, 1)) / w * w);      // h x win_cols x 4 bytes ~ 256 KB
            const int64_t n_win = (cols + win_cols - 1) / win_cols;
            static const bool windowed = [] { const char* e = std::getenv("SPARTA_BUILD_WINDOWED"); return e ? atoi(e) != 0 : true; }();      // (developer A/B)
            if (windowed && bc.ascending && h > 1 && bc.entries >= 8 * h * n_win) {
                std::vector<int64_t> cur((size_t)h, 0);
                for (int64_t c0 = 0; c0 < cols; c0 += win_cols) {
                    const int64_t c1 = c0 + win_cols;
                    for (int64_t r = r0; r < r1; r++) {
                        const int64_t i = perm[(size_t)r];
                        if (i >= a.rows) continue;
                        const int32_t* cj = a.row(i);
                        const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
                        const int64_t n = a.nnz_of(i);
                        int64_t k = cur[(size_t)(r - r0)];
                        for (; k < n && cj[k] < c1; k++) {
                            const int64_t j = cj[k];
                            const int64_t s = bc.slot[(size_t)(j / w)];
                            if (mixed && s < 0) continue;
                            base[s * h * w + h * (j % w) + (r - r0)] = v ? v[k] : 1.0f;
                        }
                        cur[(size_t)(r - r0)] = k;
                    }
                }
                continue;
            }
            for (int64_t r = r0; r < r1; r++) {
                int64_t i = perm[(size_t)r];
                if (i >= a.rows) continue;                                // vbr.cpp:211-212
                const int32_t* cj = a.row(i);
                const float* v = a.vals ? a.vals + a.rowptr[i] : nullptr;
                int64_t n = a.nnz_of(i);
                for (int64_t k = 0; k < n; k++) {
                    int64_t j = cj[k];
                    int64_t s = bc.slot[(size_t)(j / w)];
                    if (mixed && s < 0) continue;                         // a nonzero of a block that went to the sparse rows
                    // pattern-only matrices store 1 (vbr.cpp:217); duplicates: last one wins (:226)
                    base[s * h * w + h * (j % w) + (r - r0)] = v ? v[k] : 1.0f;
                }
            }
        }
    });
    trace.lap("pass 2 (scatter)");
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
