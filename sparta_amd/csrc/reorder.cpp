// reorder.cpp -- host-side row-clustering reorder engine (Jaccard / Hamming on column blocks).
//
// Reproduces the OBSERVABLE result (the grouping vector and the counters) of the reference's
// BlockingEngine::GetGrouping (src/general/blocking.cpp:633-676) for the algorithms its
// experiments use, with a different internal design:
//   * every row is pre-reduced once to its sorted list of distinct column-block ids, and the
//     cluster pattern carries both its column-level content (needed by the lossy merge) and its
//     block-level content (all the distance functions ever look at) -- a comparison is a merge-count
//     of two short int32 arrays instead of a by-value std::vector<long> copy plus a division per
//     element (blocking.cpp:923-994, definitions.h:13);
//   * the inner scan walks a compacted list of still-ungrouped rows;
//   * the per-row `distances` scratch lives on the heap (the reference uses a stack VLA,
//     blocking.cpp:159, which overflows the stack above ~2M rows) but is initialised the way the
//     reference's `float distances[rows] = {-1}` really is: element 0 = -1, all others 0.
// Behaviours that define parity are called out inline with the reference line they mirror.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <random>
#include <set>
#include <thread>
#include <type_traits>
#include <functional>
#include <utility>

#include "host_core.hpp"

namespace sparta {
namespace {

using clk = std::chrono::high_resolution_clock;
inline float us_since(clk::time_point t0) {
    return (float)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count();
}

// distinct column-block ids per row, CSR-of-blocks
struct RowBlocks {
    std::vector<int64_t> ptr;
    std::vector<int32_t> idx;
    const int32_t* row(int64_t i) const { return idx.data() + ptr[i]; }
    int64_t n(int64_t i) const { return ptr[i + 1] - ptr[i]; }
};

RowBlocks build_row_blocks(const CsrView& a, int64_t w) {
    RowBlocks rb;
    rb.ptr.resize(a.rows + 1);
    rb.idx.reserve((size_t)std::min<int64_t>(a.rowptr[a.rows], (int64_t)1 << 40));
    rb.ptr[0] = 0;
    for (int64_t i = 0; i < a.rows; i++) {
        const int32_t* r = a.row(i);
        int64_t n = a.nnz_of(i);
        int64_t last = -1;
        for (int64_t k = 0; k < n; k++) {
            int64_t b = (int64_t)r[k] / w;
            if (b != last) { rb.idx.push_back((int32_t)b); last = b; }
        }
        rb.ptr[i + 1] = (int64_t)rb.idx.size();
    }
    return rb;
}

// |A ∩ B| for two strictly ascending int32 lists
inline int64_t intersect_count(const int32_t* a, int64_t na, const int32_t* b, int64_t nb) {
    int64_t i = 0, j = 0, c = 0;
    while (i < na && j < nb) {
        int32_t x = a[i], y = b[j];
        c += (x == y);
        i += (x <= y);
        j += (y <= x);
    }
    return c;
}

// Distance from block-level quantities.  size_*: column-level lengths (only used by the empty-row
// rules), n*: number of distinct blocks, g*: cluster weights, inter: common blocks.
// Mirrors HammingDistanceGroup (blocking.cpp:859-921) and JaccardDistanceGroup (:923-994), whose
// `count_zeros = 1` makes a block present only in A weigh group_size_B and vice versa (:929-941).
inline float distance_from_counts(int sim, int64_t size_a, int64_t na, int64_t ga, int64_t size_b, int64_t nb, int64_t gb,
                                  int64_t inter) {
    if (sim == SPARTA_SIM_JACCARD) {
        if (size_a == 0 && size_b == 0) return 0.0f;           // :926
        if (size_a == 0 || size_b == 0) return 1.0f;           // :927
        int64_t count = (na - inter) * gb + (nb - inter) * ga;
        return (float)((2.0 * (double)count) / (double)(na * ga + nb * gb + count));   // :993
    }
    if (size_a == 0 && size_b == 0) return 0.0f;               // :863
    if (size_a == 0 || size_b == 0) return (float)std::max(size_a * ga, size_b * gb);   // :864 (column-level sizes)
    int64_t count = (na - inter) * gb + (nb - inter) * ga;
    return (float)count;                                       // :920
}

// Cluster pattern.  The reference keeps a std::vector of COLUMN ids and "unions" each merged row into
// it with merge_rows (src/general/utilities.cpp:145-173), which is lossy.  Closed form of what that
// loop computes for strictly ascending inputs A (pattern) and B (row):
//     B empty                      -> {}                      (the loop never runs, only B's tail is appended)
//     no element of B is <= max(A) -> B                       (breaks at j = 0 before copying anything of A)
//     otherwise, with b* = the largest element of B that is <= max(A):
//                                     {a in A : a < b*}  U  B (A's elements above b* are never copied)
struct Pattern {
    std::vector<int64_t> cols;
    std::vector<int32_t> blks;
    std::vector<int64_t> tmp;

    void rebuild_blocks(int64_t w) {
        blks.clear();
        int64_t last = -1;
        for (int64_t c : cols) {
            int64_t b = c / w;
            if (b != last) { blks.push_back((int32_t)b); last = b; }
        }
    }
    void assign(const int32_t* row, int64_t n, int64_t w) {
        cols.assign(row, row + n);
        rebuild_blocks(w);
    }
    template <typename T>
    static void merge_into(std::vector<int64_t>& out, const int64_t* a, int64_t na, const T* b, int64_t nb) {
        out.clear();
        if (nb == 0) return;
        if (na == 0 || (int64_t)b[0] > a[na - 1]) { out.assign(b, b + nb); return; }
        const int64_t max_a = a[na - 1];
        // b* = last element of B that is <= max(A)
        const T* ub = std::upper_bound(b, b + nb, max_a, [](int64_t v, const T& e) { return v < (int64_t)e; });
        const int64_t bstar = (int64_t) * (ub - 1);
        const int64_t keep = std::lower_bound(a, a + na, bstar) - a;   // elements of A strictly below b*
        out.reserve((size_t)(keep + nb));
        int64_t i = 0, j = 0;
        while (i < keep && j < nb) {
            int64_t x = a[i], y = (int64_t)b[j];
            if (x < y) { out.push_back(x); i++; }
            else if (y < x) { out.push_back(y); j++; }
            else { out.push_back(x); i++; j++; }
        }
        while (i < keep) out.push_back(a[i++]);
        while (j < nb) out.push_back((int64_t)b[j++]);
    }
    void merge(const int32_t* row, int64_t n, int64_t w) {
        merge_into(tmp, cols.data(), (int64_t)cols.size(), row, n);
        cols.swap(tmp);
        rebuild_blocks(w);
    }
};

// Shared counters / configuration of one GetGrouping run.
struct CtxBase {
    const CsrView& a;
    const sparta_reorder_cfg& cfg;
    int sim;
    int64_t w;
    int64_t comparisons = 0, merges = 0;
    float total_merge_tau = 0, total_row_distance = 0;   // float accumulators as in blocking.cpp:162-163
    float t_cmp = 0, t_merge = 0;
    CtxBase(const CsrView& a_, const sparta_reorder_cfg& c) : a(a_), cfg(c) {
        w = c.col_block_size;
        sim = (c.sim_measure & 1) ? SPARTA_SIM_JACCARD : SPARTA_SIM_HAMMING;   // 2,3 are the 'OPENMP' twins of 0,1
    }
};

// FAST policy -- valid when every row's columns are strictly ascending (the normal case): rows and the
// cluster pattern are compared as lists of distinct column-block ids, the merge uses the closed form above.
//
// Candidate filter (exact).  Most (pattern, row) pairs of a large sparse matrix share no column block at all; their distance
// is a closed form of the sizes (inter = 0) and needs no walk over the two lists.  An inverted index block -> rows tells
// which rows CAN intersect the pattern: when a seed opens a cluster, and whenever a merge brings new blocks into the
// pattern, the rows of those blocks' lists are stamped; an unstamped row has inter = 0 by construction.  A stamped row may
// still have inter = 0 (the lossy merge can drop blocks again): it simply takes the full computation.  Same distances, same
// decisions, same counters as before -- only the walk is skipped.  Stamping a block costs its list length, so a seed whose
// blocks' lists add up to more than half of the matrix switches the filter off for its cluster (dense columns).
struct Ctx : CtxBase {
    RowBlocks rb;
    Pattern pat;
    std::vector<int64_t> inv_ptr;            // block -> rows that contain it (ascending)
    std::vector<int32_t> inv_row;
    std::vector<int64_t> row_stamp, blk_stamp;
    int64_t stamp = 0;
    bool filter_on = false, have_index = false, force_filter = false;
    Ctx(const CsrView& a_, const sparta_reorder_cfg& c, bool want_index = true) : CtxBase(a_, c) {
        rb = build_row_blocks(a, w);
        if (!want_index) return;
        const char* e = std::getenv("SPARTA_REORDER_FILTER");             // 0: off, 1: on for every input, unset: on from 2048 rows
        if (e && e[0] == '0') return;
        if (a.rows > INT32_MAX || a.cols <= 0) return;
        force_filter = e && e[0] == '1';
        if (!(e && e[0] == '1') && a.rows < 2048) return;                      // small inputs: the index costs more than it saves
        const int64_t nblk = (a.cols - 1) / w + 1;
        inv_ptr.assign((size_t)nblk + 1, 0);
        for (int32_t b : rb.idx) inv_ptr[(size_t)b + 1]++;
        for (int64_t b = 0; b < nblk; b++) inv_ptr[(size_t)b + 1] += inv_ptr[(size_t)b];
        inv_row.resize(rb.idx.size());
        std::vector<int64_t> fill(inv_ptr.begin(), inv_ptr.end() - 1);
        for (int64_t i = 0; i < a.rows; i++)
            for (int64_t k = rb.ptr[(size_t)i]; k < rb.ptr[(size_t)i + 1]; k++) inv_row[(size_t)fill[(size_t)rb.idx[(size_t)k]]++] = (int32_t)i;
        row_stamp.assign((size_t)a.rows, 0);
        blk_stamp.assign((size_t)nblk, 0);
        have_index = true;
    }
    inline void stamp_new_blocks() {
        for (int32_t b : pat.blks) {
            if (blk_stamp[(size_t)b] == stamp) continue;
            blk_stamp[(size_t)b] = stamp;
            for (int64_t k = inv_ptr[(size_t)b]; k < inv_ptr[(size_t)b + 1]; k++) row_stamp[(size_t)inv_row[(size_t)k]] = stamp;
        }
    }
    inline void assign(int64_t i) {
        pat.assign(a.row(i), a.nnz_of(i), w);
        filter_on = false;
        if (!have_index) return;
        int64_t work = 0;
        for (int32_t b : pat.blks) work += inv_ptr[(size_t)b + 1] - inv_ptr[(size_t)b];
        if (work > a.rows / 2 && !force_filter) return;
        stamp++;
        filter_on = true;
        stamp_new_blocks();
    }
    inline void merge(int64_t j) {
        pat.merge(a.row(j), a.nnz_of(j), w);
        if (filter_on) stamp_new_blocks();
    }
    inline float dist(int64_t gsize, int64_t j) const {
        const int64_t inter = (filter_on && row_stamp[(size_t)j] != stamp)
                                  ? 0
                                  : intersect_count(pat.blks.data(), (int64_t)pat.blks.size(), rb.row(j), rb.n(j));
        return distance_from_counts(sim, (int64_t)pat.cols.size(), (int64_t)pat.blks.size(), gsize, a.nnz_of(j), rb.n(j), 1, inter);
    }
    // what dist(gsize, j) is never below, from the SIZES alone: the intersection holds at most the smaller set, and both distances fall as it grows (same formula, same
    // rounding).  A candidate whose floor is already above tau need not be intersected (blocking_algo 7: most candidates of a power-law matrix collide on a hub block only).
    inline float dist_floor(int64_t gsize, int64_t j) const {
        const int64_t na = (int64_t)pat.blks.size(), nb = rb.n(j);
        return distance_from_counts(sim, (int64_t)pat.cols.size(), na, gsize, a.nnz_of(j), nb, 1, std::min(na, nb));
    }
};

// LITERAL policy -- used when some row is NOT strictly ascending.  The reference never sorts or checks
// the columns of a row (its readers append them in file order, csr.cpp:262; two of its own data/minitest
// matrices have out-of-order rows), and then its two-pointer distance walk and its binary-search merge
// do whatever they do on unsorted data.  To return the same grouping on the same input we run the same
// walks on the column-level data: a run-compressed two-pointer merge-count (blocking.cpp:951-991) and the
// lower_bound-driven copy loop (utilities.cpp:152-171).  std::lower_bound is the same library routine the
// reference calls, so even its probes into unsorted data coincide.
struct LiteralCtx : CtxBase {
    std::vector<int64_t> pat, tmp;
    LiteralCtx(const CsrView& a_, const sparta_reorder_cfg& c) : CtxBase(a_, c) {}
    inline void assign(int64_t i) { pat.assign(a.row(i), a.row(i) + a.nnz_of(i)); }
    inline void merge(int64_t j) {
        const int32_t* b = a.row(j);
        const int64_t nb = a.nnz_of(j);
        tmp.clear();
        auto i = pat.begin();
        int64_t q = 0;
        while (q < nb) {
            const int64_t bv = b[q];
            auto ni = std::lower_bound(i, pat.end(), bv);
            if (ni == pat.end()) break;              // nothing of the pattern from `i` on is kept
            tmp.insert(tmp.end(), i, ni);
            tmp.push_back(bv);
            if (*ni == bv) ++ni;
            i = ni;
            q++;
        }
        for (; q < nb; q++) tmp.push_back(b[q]);
        pat.swap(tmp);
    }
    inline float dist(int64_t gsize, int64_t j) const {
        const int64_t* ra = pat.data();
        const int32_t* rb_ = a.row(j);
        const int64_t na = (int64_t)pat.size(), nb = a.nnz_of(j);
        const int64_t ga = gsize, gb = 1;
        if (na == 0 && nb == 0) return 0.0f;
        if (na == 0 || nb == 0) return sim == SPARTA_SIM_JACCARD ? 1.0f : (float)std::max(na * ga, nb * gb);
        int64_t i = 0, q = 0, count = 0, blocks_a = 0, blocks_b = 0;
        auto skip_a = [&](int64_t pos) { while (i < na && ra[i] / w == pos) i++; };
        auto skip_b = [&](int64_t pos) { while (q < nb && (int64_t)rb_[q] / w == pos) q++; };
        while (i < na && q < nb) {
            const int64_t pa = ra[i] / w, pb = (int64_t)rb_[q] / w;
            if (pa < pb) { count += gb; blocks_a++; skip_a(pa); }
            else if (pa > pb) { count += ga; blocks_b++; skip_b(pb); }
            else { blocks_a++; blocks_b++; skip_a(pa); skip_b(pb); }
        }
        while (i < na) { const int64_t pa = ra[i] / w; count += gb; blocks_a++; skip_a(pa); }
        while (q < nb) { const int64_t pb = (int64_t)rb_[q] / w; count += ga; blocks_b++; skip_b(pb); }
        if (sim == SPARTA_SIM_JACCARD) return (float)((2.0 * (double)count) / (double)(blocks_a * ga + blocks_b * gb + count));
        return (float)count;
    }
};

bool all_rows_strictly_ascending(const CsrView& a) {
    for (int64_t i = 0; i < a.rows; i++) {
        const int32_t* r = a.row(i);
        const int64_t n = a.nnz_of(i);
        for (int64_t k = 1; k < n; k++)
            if (r[k] <= r[k - 1]) return false;
    }
    return true;
}

// the reference's `float distances[cmat.rows] = {-1};` (blocking.cpp:159,255,436): a VLA with a
// one-element initialiser -> element 0 is -1, every other element is 0.
std::vector<float> make_distances(int64_t rows) {
    std::vector<float> d((size_t)rows, 0.0f);
    if (rows > 0) d[0] = -1.0f;
    return d;
}

// Algorithms 3 (IterativeBlockingPatternCLOCKED, blocking.cpp:156-243) and 4 (IterativeBlockingQueue,
// :245-338).  Both visit the still-ungrouped rows after the seed in ascending order and differ only in
// bookkeeping that cannot change the result: CLOCKED also runs its prune test on already-grouped rows,
// whose `distances` entry is never read again.
template <class C>
void clocked(C& c, int64_t* grouping) {
    const int64_t rows = c.a.rows;
    const float tau = c.cfg.tau;
    std::fill(grouping, grouping + rows, (int64_t)-1);
    std::vector<float> dist = make_distances(rows);
    std::vector<int64_t> alive((size_t)rows), next;
    std::iota(alive.begin(), alive.end(), (int64_t)0);
    next.reserve((size_t)rows);

    size_t head = 0;   // alive[head] is the next seed
    while (head < alive.size()) {
        const int64_t i = alive[head];
        grouping[i] = i;                                           // :172 group id = seed row
        c.assign(i);                // :173
        int64_t gsize = 1;
        const float di = dist[(size_t)i];
        auto t0 = clk::now();
        next.clear();
        for (size_t q = head + 1; q < alive.size(); q++) {
            const int64_t j = alive[q];
            float& dj = dist[(size_t)j];
            // triangle-inequality prune (:192-196); note it resets distances[j]
            if (di != -1.0f && dj != -1.0f && std::fabs(di - dj) > tau) {
                dj = -1.0f;
                next.push_back(j);
                continue;
            }
            c.comparisons++;
            const float d = c.dist(gsize, j);
            dj = d;
            if (d <= tau) {                                        // :207 (<=)
                c.total_merge_tau += d;
                c.total_row_distance += (float)(j - i);
                c.merges++;
                grouping[j] = i;
                if (c.cfg.use_pattern) {
                    auto tm = clk::now();
                    c.merge(j);     // :217
                    c.t_merge += us_since(tm);
                }
                if (c.cfg.use_groups) gsize++;                     // :221-224
            } else {
                next.push_back(j);
            }
        }
        c.t_cmp += us_since(t0);
        // compact: rows before `head` are done; keep the survivors
        alive.swap(next);
        head = 0;
    }
}

// Algorithms 3 / 4 in near-linear time, exactly -- for the reference's DEFAULT setting (Jaccard, use_groups = 0) and tau < 1.
//
// With cluster weight 1 the Jaccard distance of a pattern and a row that share no block is exactly 1.0f (2c / 2c), so such
// a row is never merged (tau < 1) and the only thing the scan does to it is maintain `distances[j]`:
//        pruned  (d_i != -1 and d_j != -1 and |d_i - d_j| > tau)  ->  d_j = -1, no comparison counted
//        else                                                     ->  d_j = 1.0, one comparison counted
// After its first such visit a row's entry is -1 or 1.0, and one seed applies ONE of two maps to all of them at once:
//        g ("nothing with 1.0 is pruned"):  -1 -> 1.0,  1.0 -> 1.0            h ("1.0 is pruned"):  -1 -> 1.0,  1.0 -> -1
// (an entry of -1 is never pruned).  So rows are kept in two classes with counters, a row remembers the seed ordinal at which
// it was last touched individually and its class then, and its current class is that class pushed through the g / h sequence
// since -- O(1) with a prefix count of h and the position of the last g.  Per seed, only these rows are handled one by one, in
// ascending order as the reference's scan would meet them (a heap):
//   * CANDIDATES: rows that share a block with the pattern (inverted index; rows of blocks that a merge brings in join the
//     heap on the fly; an empty seed's candidates are the empty rows): exact prune test, exact distance, exact merge;
//   * rows whose entry is still an individual value (the seed's previous candidates that were compared but not merged;
//     initially every row, with the VLA's 0): the prune rule applied to that value, after which they join a class.
// Everything else is two counter updates.  Same grouping, same counters, same float accumulators (only merges add to them,
// and those happen in the same order) as clocked() -- checked against it and against the reference in the tests.
void clocked_sparse(Ctx& c, int64_t* grouping) {
    const CsrView& a = c.a;
    const int64_t rows = a.rows;
    const float tau = c.cfg.tau;
    enum : uint8_t { NEG = 0, ONE = 1, IND = 2 };
    std::fill(grouping, grouping + rows, (int64_t)-1);
    std::vector<uint8_t> cls((size_t)rows, IND);
    std::vector<float> val((size_t)rows, 0.0f);
    std::vector<int32_t> touched((size_t)rows, -1);            // seed ordinal of the last individual handling
    if (rows > 0) val[0] = -1.0f;                                // `float distances[rows] = {-1}`: element 0 is -1, the rest 0
    std::vector<int64_t> fresh, next_fresh;                      // alive rows whose entry is an individual value
    fresh.reserve((size_t)rows);
    for (int64_t j = 0; j < rows; j++) fresh.push_back(j);
    std::vector<int32_t> hpre;                                   // hpre[s] = number of h maps among seeds 0..s
    std::vector<int32_t> lastg;                                  // lastg[s] = last seed <= s whose map is g, or -1
    int64_t cnt[2] = {0, 0};
    std::vector<int64_t> empties;                                // rows without nonzeros, ascending
    for (int64_t j = 0; j < rows; j++) if (a.nnz_of(j) == 0) empties.push_back(j);
    auto current_class = [&](int64_t j, int32_t upto) -> uint8_t {   // class of a stable row after seeds 0..upto
        const int32_t t = touched[(size_t)j];
        const uint8_t k = cls[(size_t)j];
        if (upto <= t) return k;
        const int32_t L = lastg[(size_t)upto];
        if (L > t) return ((hpre[(size_t)upto] - hpre[(size_t)L]) & 1) ? NEG : ONE;
        return ((hpre[(size_t)upto] - hpre[(size_t)t]) & 1) ? (uint8_t)(1 - k) : k;
    };
    // heap of rows to handle individually in this seed: (row, is_candidate)
    std::vector<int64_t> heap;
    auto heap_push = [&](int64_t j) { heap.push_back(j); std::push_heap(heap.begin(), heap.end(), std::greater<int64_t>()); };
    auto heap_pop = [&]() { std::pop_heap(heap.begin(), heap.end(), std::greater<int64_t>()); const int64_t j = heap.back(); heap.pop_back(); return j; };
    std::vector<int64_t> queued((size_t)rows, -1);               // seed ordinal at which the row was last put on the heap
    int64_t seed_row = 0;
    int32_t s = 0;
    while (true) {
        while (seed_row < rows && grouping[seed_row] != -1) seed_row++;
        if (seed_row >= rows) break;
        const int64_t i = seed_row;
        // the seed's own entry
        float di;
        if (cls[(size_t)i] == IND) di = val[(size_t)i];
        else { const uint8_t k = current_class(i, s - 1); di = k == NEG ? -1.0f : 1.0f; cnt[k]--; }
        grouping[i] = i;
        auto t0 = clk::now();
        // candidates: rows of the pattern's blocks (ascending lists), or the empty rows for an empty seed
        c.pat.assign(a.row(i), a.nnz_of(i), c.w);
        c.stamp++;
        heap.clear();
        auto enqueue_block_rows = [&](int64_t after) {
            for (int32_t b : c.pat.blks) {
                if (c.blk_stamp[(size_t)b] == c.stamp) continue;
                c.blk_stamp[(size_t)b] = c.stamp;
                const int32_t* lo = c.inv_row.data() + c.inv_ptr[(size_t)b];
                const int32_t* hi = c.inv_row.data() + c.inv_ptr[(size_t)b + 1];
                for (const int32_t* q = std::upper_bound(lo, hi, (int32_t)after); q < hi; q++) {
                    const int64_t j = *q;
                    c.row_stamp[(size_t)j] = c.stamp;
                    if (grouping[j] == -1 && queued[(size_t)j] != s) { queued[(size_t)j] = s; heap_push(j); }
                }
            }
        };
        enqueue_block_rows(i);
        if (a.nnz_of(i) == 0) {
            for (auto q = std::upper_bound(empties.begin(), empties.end(), i); q != empties.end(); ++q) {
                const int64_t j = *q;
                c.row_stamp[(size_t)j] = c.stamp;
                if (grouping[j] == -1 && queued[(size_t)j] != s) { queued[(size_t)j] = s; heap_push(j); }
            }
        }
        for (int64_t j : fresh)
            if (j != i && grouping[j] == -1 && queued[(size_t)j] != s) { queued[(size_t)j] = s; heap_push(j); }
        next_fresh.clear();
        // the bulk map of this seed
        const bool h_map = di != -1.0f && std::fabs(di - 1.0f) > tau;
        int64_t joined[2] = {0, 0};                              // rows that become stable in this seed, by class
        while (!heap.empty()) {
            const int64_t j = heap_pop();
            if (grouping[j] != -1) continue;
            // entry of row j as the scan finds it
            float dj;
            if (cls[(size_t)j] == IND) dj = val[(size_t)j];
            else { const uint8_t k = current_class(j, s - 1); dj = k == NEG ? -1.0f : 1.0f; cnt[k]--; }
            const bool pruned = di != -1.0f && dj != -1.0f && std::fabs(di - dj) > tau;      // blocking.cpp:192-196
            if (pruned) {
                cls[(size_t)j] = NEG; touched[(size_t)j] = s; joined[NEG]++;
                continue;
            }
            c.comparisons++;
            float d;
            if (c.row_stamp[(size_t)j] == c.stamp) {
                const int64_t inter = intersect_count(c.pat.blks.data(), (int64_t)c.pat.blks.size(), c.rb.row(j), c.rb.n(j));
                d = distance_from_counts(c.sim, (int64_t)c.pat.cols.size(), (int64_t)c.pat.blks.size(), 1, a.nnz_of(j), c.rb.n(j), 1, inter);
            } else {
                d = 1.0f;                                        // no common block: 2c / 2c (an empty row against a non-empty pattern: :927)
            }
            if (d <= tau) {                                      // :207
                c.total_merge_tau += d;
                c.total_row_distance += (float)(j - i);
                c.merges++;
                grouping[j] = i;
                if (c.cfg.use_pattern) {
                    auto tm = clk::now();
                    c.pat.merge(a.row(j), a.nnz_of(j), c.w);
                    enqueue_block_rows(j);
                    c.t_merge += us_since(tm);
                }
            } else if (d == 1.0f) {
                cls[(size_t)j] = ONE; touched[(size_t)j] = s; joined[ONE]++;
            } else {
                cls[(size_t)j] = IND; val[(size_t)j] = d; next_fresh.push_back(j);
            }
        }
        // every other alive row: one of the two maps
        if (h_map) { c.comparisons += cnt[NEG]; std::swap(cnt[NEG], cnt[ONE]); }
        else { c.comparisons += cnt[NEG] + cnt[ONE]; cnt[ONE] += cnt[NEG]; cnt[NEG] = 0; }
        cnt[NEG] += joined[NEG];
        cnt[ONE] += joined[ONE];
        hpre.push_back((hpre.empty() ? 0 : hpre.back()) + (h_map ? 1 : 0));
        lastg.push_back(h_map ? (lastg.empty() ? -1 : lastg.back()) : s);
        fresh.swap(next_fresh);
        c.t_cmp += us_since(t0);
        s++;
    }
}

// Algorithm 0 (IterativeBlockingPattern, blocking.cpp:89-154): no prune, strict `<`, and -- because the
// `if (use_pattern)` there guards only a timer macro -- the pattern merge ALWAYS runs (:128-132).
template <class C>
void plain(C& c, int64_t* grouping) {
    const int64_t rows = c.a.rows;
    const float tau = c.cfg.tau;
    std::fill(grouping, grouping + rows, (int64_t)-1);
    std::vector<int64_t> alive((size_t)rows), next;
    std::iota(alive.begin(), alive.end(), (int64_t)0);
    while (!alive.empty()) {
        const int64_t i = alive[0];
        grouping[i] = i;
        c.assign(i);
        int64_t gsize = 1;
        next.clear();
        for (size_t q = 1; q < alive.size(); q++) {
            const int64_t j = alive[q];
            c.comparisons++;
            const float d = c.dist(gsize, j);
            if (d < tau) {                                         // :124 (<)
                c.merges++;
                grouping[j] = i;
                c.merge(j);
                if (c.cfg.use_groups) gsize++;
            } else {
                next.push_back(j);
            }
        }
        alive.swap(next);
    }
}

// Algorithm 1 (IterativeBlockingPatternMN, blocking.cpp:19-87): the plain algorithm (strict `<`) with an m:n guard taken from
// check_structured_sparsity / update_structured_sparsity (utilities.cpp:56-129): a candidate that passes the distance test is
// merged only if, inside the current run of structured_n merged rows, none of its columns has already been hit structured_m
// times.  Here `use_pattern` does guard the pattern merge (:70-71).  The guard works on column ids, literally as written.
template <class C>
void structured_mn(C& c, int64_t* grouping) {
    const int64_t rows = c.a.rows;
    const float tau = c.cfg.tau;
    const int m = c.cfg.structured_m, n = c.cfg.structured_n;
    std::fill(grouping, grouping + rows, (int64_t)-1);
    std::vector<int64_t> alive((size_t)rows), next, sp, sc, np_, nc_;
    std::iota(alive.begin(), alive.end(), (int64_t)0);
    auto check = [&](const int32_t* row, int64_t len) {
        size_t i = 0; int64_t j = 0;
        while (i < sp.size() && j < len) {
            if (sp[i] < row[j]) i++;
            else if (sp[i] > row[j]) j++;
            else { if (sc[i] >= m) return false; i++; j++; }
        }
        return true;
    };
    auto update = [&](const int32_t* row, int64_t len) {
        np_.clear(); nc_.clear();
        size_t i = 0; int64_t j = 0;
        while (i < sp.size() && j < len) {
            if (sp[i] < row[j]) { np_.push_back(sp[i]); nc_.push_back(sc[i]); i++; }
            else if (sp[i] > row[j]) { np_.push_back(row[j]); nc_.push_back(1); j++; }
            else { np_.push_back(sp[i]); nc_.push_back(sc[i] + 1); i++; j++; }
        }
        for (; i < sp.size(); i++) { np_.push_back(sp[i]); nc_.push_back(sc[i]); }
        for (; j < len; j++) { np_.push_back(row[j]); nc_.push_back(1); }
        sp.swap(np_); sc.swap(nc_);
    };
    while (!alive.empty()) {
        const int64_t i = alive[0];
        grouping[i] = i;
        c.assign(i);
        int64_t gsize = 1;
        int row_counter = 1;
        sp.assign(c.a.row(i), c.a.row(i) + c.a.nnz_of(i));
        sc.assign(sp.size(), 1);
        next.clear();
        for (size_t q = 1; q < alive.size(); q++) {
            const int64_t j = alive[q];
            c.comparisons++;
            const float d = c.dist(gsize, j);
            bool merged = false;
            if (d < tau) {
                bool ok = true;
                if (row_counter % n == 0) { row_counter = 0; sp.clear(); sc.clear(); }
                else ok = check(c.a.row(j), c.a.nnz_of(j));
                if (ok) {
                    c.merges++;
                    grouping[j] = i;
                    if (c.cfg.use_pattern) c.merge(j);
                    if (c.cfg.use_groups) gsize++;
                    update(c.a.row(j), c.a.nnz_of(j));
                    row_counter++;
                    merged = true;
                }
            }
            if (!merged) next.push_back(j);
        }
        alive.swap(next);
    }
}

// Algorithm 5 (-a 5 dispatches to IterativeBlockingKeeper, blocking.cpp:655, :433-549): clusters are
// capped at row_block_size rows; the best rejected candidates are kept in an ordered set and used to
// pad short clusters; complete clusters are numbered `seed`, incomplete ones `seed + rows` (:450,527-533).
//
// The candidate set is trimmed in the reference with
//       auto it = best.end(); advance(it, k); best.erase(it, best.end());          (:509-511)
// i.e. by incrementing PAST end().  On libstdc++ that walk is deterministic: incrementing the header
// node lands on the right-most node (or its left child) and the walk wraps around.  Which elements
// are dropped therefore depends on the red-black tree's shape.  To give the same grouping we perform
// the very same iterator walk on the same container type; this relies on libstdc++ and is guarded.
#if !defined(__GLIBCXX__)
#error "IterativeBlockingKeeper parity relies on libstdc++'s std::set iterator behaviour"
#endif
template <class C>
void keeper(C& c, int64_t* grouping) {
    const int64_t rows = c.a.rows;
    const float tau = c.cfg.tau;
    const int64_t max_h = c.cfg.row_block_size;
    std::fill(grouping, grouping + rows, (int64_t)-1);
    std::vector<float> dist = make_distances(rows);
    std::vector<int64_t> merged;

    for (int64_t i = 0; i < rows; i++) {
        if (grouping[i] != -1) continue;
        std::set<std::pair<float, int64_t>> best;
        merged.clear();
        const int64_t group_number = i + rows;                     // :450
        grouping[i] = group_number;
        merged.push_back(i);
        c.assign(i);
        int64_t gsize = 1;
        const float di = dist[(size_t)i];
        auto t0 = clk::now();
        for (int64_t j = i + 1; j < rows; j++) {
            if (gsize == max_h) break;                             // :463-466
            float& dj = dist[(size_t)j];
            if (di != -1.0f && dj != -1.0f && std::fabs(di - dj) > tau) { dj = -1.0f; continue; }   // :469-473
            if (grouping[j] != -1) continue;
            c.comparisons++;
            const float d = c.dist(gsize, j);                 // :480 weight is ALWAYS the cluster size
            dj = d;
            if (d <= tau) {
                c.total_merge_tau += d;
                c.total_row_distance += (float)(j - i);
                c.merges++;
                grouping[j] = group_number;
                merged.push_back(j);
                if (c.cfg.use_pattern) {
                    auto tm = clk::now();
                    c.merge(j);
                    c.t_merge += us_since(tm);
                }
                gsize++;                                           // :501 unconditional
            } else {
                best.insert({d, j});
                // (:507) size_t comparison in the reference; the right-hand side is >= 1 here
                if (best.size() > (size_t)(max_h - (int64_t)merged.size())) {
                    auto it = best.end();
                    std::advance(it, max_h - (int64_t)merged.size());   // walks past end(): see note above
                    best.erase(it, best.end());
                }
            }
        }
        if (gsize < max_h) {                                       // :517-525 pad from the kept candidates
            for (auto it = best.begin(); it != best.end() && gsize != max_h; ++it) {
                grouping[it->second] = group_number;
                merged.push_back(it->second);
                gsize++;
            }
        }
        if (gsize == max_h)                                        // :527-533 complete blocks sort first
            for (int64_t r : merged) grouping[r] -= rows;
        c.t_cmp += us_since(t0);
    }
}

}  // namespace

// ---- public helpers ---------------------------------------------------------------------------

// ---- Algorithm 7: LSH-bucketed clustering -- an EXTENSION, not in the reference -----------------
// The reference's scans compare every seed with every later row: fine for 10^4..10^5 rows, hopeless for 10^6..10^7 (and its
// stack VLA crashes above ~2 M rows).  This keeps the cluster semantics of algorithm 3 -- seeds in ascending row order, exact
// distance of the evolving pattern and the candidate row (same Jaccard / Hamming functions, same cluster weights, same lossy
// merge), merge when dist <= tau -- but only LOOKS at rows that an LSH index proposes:
//   0. rows with identical block sets are folded first (distance 0: any tau merges them, and the merge leaves the pattern
//      unchanged); power-law graphs have huge families of them (every row that only touches the hub block);
//   1. every distinct row gets K = bands x r minhash values of its block set (multiply-shift hashes of a pre-mixed block id),
//      band b's key = hash of its r values; per band the (key, row) pairs are sorted: rows with sim s collide in a given band
//      with probability s^r, in at least one with 1 - (1 - s^r)^bands;
//   2. a seed's candidates are the ungrouped later rows in the buckets of the PATTERN's signature (signature of a union =
//      element-wise minimum, so after a merge only the bands whose key changed are looked up again), ordered by the number of
//      colliding bands (the estimate of the similarity), at most `minhash_max_eval` exact comparisons per seed and
//      kScanLimit entries per bucket look-up, so that uninformative giant buckets (rows that share only a hub block collide
//      in many bands and are still too far apart to merge) cost a bounded amount.
// Deterministic (fixed hash constants).  Quality is checked against the exact algorithm on inputs both can handle
// (tests/test_host_golden.py::test_minhash_*): it cannot be bit-identical to the reference and does not claim to be.

inline uint64_t mix64(uint64_t x) {              // splitmix64 finaliser
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

template <class F>
void parallel_chunks(int64_t n, int n_threads, F&& f) {
    n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n));
    if (n_threads == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    const int64_t per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; t++) {
        const int64_t lo = t * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        th.emplace_back([&f, lo, hi] { f(lo, hi); });
    }
    for (auto& t : th) t.join();
}

void minhash_lsh(Ctx& c, int64_t* grouping) {
    const CsrView& a = c.a;
    const sparta_reorder_cfg& cfg = c.cfg;
    const int64_t rows = a.rows;
    const float tau = cfg.tau;
    std::fill(grouping, grouping + rows, (int64_t)-1);
    if (rows == 0) return;
    int n_threads = std::min(16, sparta::host_threads());                 // (the CPUs the process may use: host_core.hpp)
    if (const char* e = std::getenv("SPARTA_REORDER_THREADS")) n_threads = std::max(1, atoi(e));
    int64_t kScanLimit = 256;                                    // entries looked at per bucket look-up
    if (const char* e = std::getenv("SPARTA_MINHASH_SCAN")) kScanLimit = std::max(1, atoi(e));
    int64_t patience = 32;                                       // consecutive failed candidates (best estimates first) before a seed gives up
    if (const char* e = std::getenv("SPARTA_MINHASH_PATIENCE")) patience = std::max(1, atoi(e));

    // ---- 0. fold identical rows: rep[i] = first row with the same block set (and, for Hamming's empty-row rule, the same nnz = 0 state)
    std::vector<int32_t> rep((size_t)rows);
    std::vector<int32_t> uniq;                                   // ascending representatives
    std::vector<int64_t> dup_ptr;                                // duplicates of uniq[u] (excluding itself): dup_row[dup_ptr[u] .. dup_ptr[u+1])
    std::vector<int32_t> dup_row;
    {
        std::vector<uint64_t> fh((size_t)rows);
        parallel_chunks(rows, n_threads, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                uint64_t h = 0x1234567 + (uint64_t)c.rb.n(i);
                const int32_t* r = c.rb.row(i);
                for (int64_t k = 0; k < c.rb.n(i); k++) h = mix64(h ^ (uint64_t)(uint32_t)r[k]);
                fh[(size_t)i] = h;
            }
        });
        std::vector<int32_t> order((size_t)rows);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return fh[(size_t)x] != fh[(size_t)y] ? fh[(size_t)x] < fh[(size_t)y] : x < y; });
        auto same = [&](int32_t x, int32_t y) {
            return c.rb.n(x) == c.rb.n(y) && std::equal(c.rb.row(x), c.rb.row(x) + c.rb.n(x), c.rb.row(y));
        };
        for (size_t p = 0; p < order.size();) {                  // a run of equal hashes: group by true equality (first occurrence = representative)
            size_t q = p;
            while (q < order.size() && fh[(size_t)order[q]] == fh[(size_t)order[p]]) q++;
            for (size_t x = p; x < q; x++) {
                const int32_t i = order[x];
                rep[(size_t)i] = i;
                for (size_t y = p; y < x; y++)
                    if (rep[(size_t)order[y]] == order[y] && same(order[y], i)) { rep[(size_t)i] = order[y]; break; }
            }
            p = q;
        }
        std::vector<int64_t> cnt((size_t)rows, 0);
        for (int64_t i = 0; i < rows; i++) {
            if (rep[(size_t)i] == i) uniq.push_back((int32_t)i); else cnt[(size_t)rep[(size_t)i]]++;
        }
        std::vector<int64_t> where((size_t)rows, -1);
        dup_ptr.assign(uniq.size() + 1, 0);
        for (size_t u = 0; u < uniq.size(); u++) { where[(size_t)uniq[u]] = (int64_t)u; dup_ptr[u + 1] = dup_ptr[u] + cnt[(size_t)uniq[u]]; }
        dup_row.resize((size_t)dup_ptr.back());
        std::vector<int64_t> fill(dup_ptr.begin(), dup_ptr.end() - 1);
        for (int64_t i = 0; i < rows; i++)
            if (rep[(size_t)i] != i) dup_row[(size_t)fill[(size_t)where[(size_t)rep[(size_t)i]]]++] = (int32_t)i;
        // `where` doubles as row -> index in uniq below
        for (int64_t i = 0; i < rows; i++) rep[(size_t)i] = (int32_t)where[(size_t)i];             // -1 for non-representatives
    }
    const int64_t U = (int64_t)uniq.size();
    const bool verbose = std::getenv("SPARTA_MINHASH_VERBOSE") != nullptr;
    auto t_ph = clk::now();
    if (verbose) fprintf(stderr, "minhash: %lld rows, %lld distinct block sets\n", (long long)rows, (long long)U);

    // ---- 1. signatures and band tables over the representatives
    int bands = cfg.minhash_bands > 0 ? cfg.minhash_bands : 16;
    int r = cfg.minhash_rows;
    if (r <= 0) {                                                // largest r whose threshold (1/bands)^(1/r) stays below 0.75 x (1 - tau)
        const double s0 = c.sim == SPARTA_SIM_JACCARD ? std::max(0.05, 1.0 - (double)tau) : 0.5;
        r = 1;
        for (int t = 2; t <= 6; t++)
            if (std::pow(1.0 / bands, 1.0 / t) <= 0.75 * s0) r = t;
    }
    bands = std::min(bands, 64);
    r = std::min(r, 8);
    const int K = bands * r;
    int64_t max_eval = cfg.minhash_max_eval > 0 ? cfg.minhash_max_eval : 512;
    const int64_t cap = cfg.minhash_max_rows > 0 ? cfg.minhash_max_rows : INT64_MAX;
    std::vector<uint64_t> ha((size_t)K), hb((size_t)K);
    for (int k = 0; k < K; k++) { ha[(size_t)k] = mix64(2 * (uint64_t)k + 1) | 1ull; hb[(size_t)k] = mix64(0xabcdef01ull + (uint64_t)k); }
    std::vector<uint32_t> sig((size_t)U * (size_t)K);
    parallel_chunks(U, n_threads, [&](int64_t lo, int64_t hi) {
        for (int64_t u = lo; u < hi; u++) {
            uint32_t* sg = sig.data() + (size_t)u * K;
            for (int k = 0; k < K; k++) sg[k] = UINT32_MAX;
            const int32_t i = uniq[(size_t)u];
            const int32_t* rw = c.rb.row(i);
            for (int64_t q = 0; q < c.rb.n(i); q++) {
                const uint64_t x = mix64((uint64_t)(uint32_t)rw[q]);
                for (int k = 0; k < K; k++) sg[k] = std::min(sg[k], (uint32_t)((ha[(size_t)k] * x + hb[(size_t)k]) >> 32));
            }
        }
    });
    auto band_key = [&](const uint32_t* sg, int b) {
        uint64_t h = 0x51ed270b + (uint64_t)b;
        for (int t = 0; t < r; t++) h = mix64(h ^ sg[b * r + t]);
        return h;
    };
    // Inside a bucket (equal band keys) the rows are ordered by ONE MORE minhash value (the first value of the next band): two rows agree on it with probability = their
    // similarity, so a seed's true neighbours sit right around its own place in the bucket and a look-up that scans kScanLimit entries OUTWARD from there meets them
    // first.  (Round 5: with the bucket in row order and the scan running forward from the seed, a bucket of ~1000 rows -- 96 000 rows, 2 values per band -- gave up 3/4
    // of a cluster's members: the benchmark set's 2000 clusters of 48 rows came back as 3417 groups, 624 of them whole.)
    struct Entry { uint64_t key; uint32_t key2; int32_t u; };
    auto second_key = [&](const uint32_t* sg, int b) { return sg[((b + 1) % bands) * r]; };
    auto entry_less = [](const Entry& x, const Entry& y) { return x.key != y.key ? x.key < y.key : (x.key2 != y.key2 ? x.key2 < y.key2 : x.u < y.u); };
    std::vector<std::vector<Entry>> table((size_t)bands);
    std::vector<int32_t> slot((size_t)bands * (size_t)U);
    parallel_chunks(bands, n_threads, [&](int64_t lo, int64_t hi) {
        for (int64_t b = lo; b < hi; b++) {
            std::vector<Entry>& t = table[(size_t)b];
            t.resize((size_t)U);
            for (int64_t u = 0; u < U; u++) t[(size_t)u] = Entry{band_key(sig.data() + (size_t)u * K, (int)b), second_key(sig.data() + (size_t)u * K, (int)b), (int32_t)u};
            std::sort(t.begin(), t.end(), entry_less);
            int32_t* ps = slot.data() + (size_t)b * (size_t)U;   // where each row sits in this band's table: a seed's first look-up needs no search
            for (int64_t q = 0; q < U; q++) ps[t[(size_t)q].u] = (int32_t)q;
        }
    });

    if (verbose) fprintf(stderr, "minhash: bands %d x %d values, signatures + tables %.2f s\n", bands, r, us_since(t_ph) * 1e-6);
    float t_look = 0;
    // ---- 2. seeds in ascending row order
    std::vector<int32_t> seen((size_t)U, -1);                    // candidate of which seed (index in uniq)
    std::vector<uint16_t> hits((size_t)U, 0);
    std::vector<int32_t> cand, fresh, sorted;
    std::vector<uint32_t> sigP((size_t)K);
    std::vector<uint64_t> keyP((size_t)bands);
    auto t_all = clk::now();
    for (int64_t su = 0; su < U; su++) {
        const int64_t i = uniq[(size_t)su];
        if (grouping[i] != -1) continue;
        int64_t members = 0, gsize = 1, evals = 0, failed = 0;
        auto take = [&](int64_t u, float d) {                    // representative u and its duplicates join the cluster of seed i
            const int64_t j = uniq[(size_t)u];
            grouping[j] = i;
            for (int64_t k = dup_ptr[(size_t)u]; k < dup_ptr[(size_t)u + 1]; k++) {
                grouping[dup_row[(size_t)k]] = i;
                c.merges++;                                       // a duplicate of a row already in the cluster: distance to the pattern as measured for it
                c.total_merge_tau += (j == i ? 0.0f : d);
                c.total_row_distance += (float)(dup_row[(size_t)k] - i);
            }
            members += 1 + (dup_ptr[(size_t)u + 1] - dup_ptr[(size_t)u]);
            if (cfg.use_groups) gsize += (j == i ? 0 : 1) + (dup_ptr[(size_t)u + 1] - dup_ptr[(size_t)u]);
        };
        c.assign(i);
        take(su, 0.0f);
        std::copy(sig.begin() + (size_t)su * K, sig.begin() + (size_t)(su + 1) * K, sigP.begin());
        for (int b = 0; b < bands; b++) keyP[(size_t)b] = band_key(sigP.data(), b);
        uint64_t look = bands >= 64 ? ~0ull : ((1ull << bands) - 1);         // bands to look up in this round
        bool first_round = true;                                 // the pattern's signature is still the seed's own
        while (look && members < cap && evals < max_eval && failed < patience) {
            fresh.clear();
            auto tl = clk::now();
            for (int b = 0; b < bands; b++) {
                if (!((look >> b) & 1)) continue;
                const std::vector<Entry>& t = table[(size_t)b];
                const uint64_t key = keyP[(size_t)b];
                // the pattern's place in the bucket (a seed's first look-up: its own slot), then outward: one entry above, one below, ... while the key matches
                const int64_t at = first_round ? (int64_t)slot[(size_t)b * (size_t)U + (size_t)su]
                                               : (int64_t)(std::lower_bound(t.begin(), t.end(), Entry{key, second_key(sigP.data(), b), (int32_t)su}, entry_less) - t.begin());
                int64_t up = first_round ? at + 1 : at, dn = at - 1;
                const int64_t tn = (int64_t)t.size();
                auto propose = [&](int32_t u) {
                    if (u <= (int32_t)su || grouping[uniq[(size_t)u]] != -1) return;            // (rows before the seed are all grouped already; counted against the scan all the same)
                    if (seen[(size_t)u] != (int32_t)su) { seen[(size_t)u] = (int32_t)su; hits[(size_t)u] = 0; fresh.push_back(u); }
                    else if (hits[(size_t)u] == UINT16_MAX) return;                              // already evaluated for this seed
                    hits[(size_t)u]++;
                };
                for (int64_t n = 0; n < kScanLimit;) {
                    const bool can_up = up < tn && t[(size_t)up].key == key, can_dn = dn >= 0 && t[(size_t)dn].key == key;
                    if (!can_up && !can_dn) break;
                    if (can_up) { propose(t[(size_t)up].u); up++; n++; }
                    if (can_dn && n < kScanLimit) { propose(t[(size_t)dn].u); dn--; n++; }
                }
            }
            look = 0;
            first_round = false;
            if (verbose) t_look += us_since(tl);
            // rows proposed before and not yet evaluated stay in `cand`; order all pending ones by collisions (desc), row (asc)
            cand.insert(cand.end(), fresh.begin(), fresh.end());
            cand.erase(std::remove_if(cand.begin(), cand.end(), [&](int32_t u) { return hits[(size_t)u] == UINT16_MAX || grouping[uniq[(size_t)u]] != -1; }),
                       cand.end());
            {                                                    // counting sort by collisions, most first (stable: discovery order inside a class)
                int64_t cnt[66] = {0};
                auto klass = [&](int32_t u) { return std::min<int>(hits[(size_t)u], 64); };
                for (int32_t u : cand) cnt[klass(u)]++;
                int64_t start[66], acc = 0;
                for (int hh = 64; hh >= 0; hh--) { start[hh] = acc; acc += cnt[hh]; }
                sorted.resize(cand.size());
                for (int32_t u : cand) sorted[(size_t)start[klass(u)]++] = u;
                cand.swap(sorted);
            }
            size_t pos = 0;
            for (; pos < cand.size() && members < cap && evals < max_eval && failed < patience; pos++) {
                const int32_t u = cand[pos];
                const int64_t j = uniq[(size_t)u];
                hits[(size_t)u] = UINT16_MAX;
                if (members + 1 + (dup_ptr[(size_t)u + 1] - dup_ptr[(size_t)u]) > cap) continue;   // the row and its identical copies would not fit: left to another seed
                evals++;
                c.comparisons++;
                if (c.dist_floor(gsize, j) > tau) { failed++; continue; }      // (hopeless by size: the exact distance is above tau too -- same decisions, no intersection)
                const float d = c.dist(gsize, j);
                if (!(d <= tau)) { failed++; continue; }
                failed = 0;
                c.total_merge_tau += d;
                c.total_row_distance += (float)(j - i);
                c.merges++;
                take(u, d);
                if (cfg.use_pattern) {
                    auto tm = clk::now();
                    c.merge(j);
                    c.t_merge += us_since(tm);
                    const uint32_t* sj = sig.data() + (size_t)u * K;
                    for (int b = 0; b < bands; b++) {
                        bool changed = false;
                        for (int t = 0; t < r; t++)
                            if (sj[b * r + t] < sigP[(size_t)(b * r + t)]) { sigP[(size_t)(b * r + t)] = sj[b * r + t]; changed = true; }
                        if (changed) { keyP[(size_t)b] = band_key(sigP.data(), b); look |= 1ull << b; }
                    }
                    if (look) { pos++; break; }                  // the pattern moved: propose again before going on
                }
            }
            cand.erase(cand.begin(), cand.begin() + (std::ptrdiff_t)pos);
            if (!look && cand.empty()) break;
            if (!look && pos == 0) break;
        }
        cand.clear();
    }
    c.t_cmp += us_since(t_all);
    if (verbose) fprintf(stderr, "minhash: seed loop %.2f s (bucket look-ups %.2f s), %lld comparisons, %lld merges\n", us_since(t_all) * 1e-6, t_look * 1e-6,
                         (long long)c.comparisons, (long long)c.merges);
}


// src/general/utilities.cpp:8-20.  The reference sorts row indices with std::sort (introsort, NOT
// stable) under `grouping[i] < grouping[j]`; the order of rows inside one group is therefore whatever
// libstdc++'s introsort leaves.  Calling the same standard algorithm with the same strict-weak order
// on the same initial sequence (iota) gives the same permutation on the same standard library.
std::vector<int64_t> get_permutation(const int64_t* grouping, int64_t n) {
    std::vector<int64_t> v((size_t)n);
    std::iota(v.begin(), v.end(), (int64_t)0);
    std::sort(v.begin(), v.end(), [grouping](int64_t x, int64_t y) { return grouping[x] < grouping[y]; });
    return v;
}

// src/general/utilities.cpp:22-43: start offset of every distinct id in the sorted grouping + sentinel n
std::vector<int64_t> get_partition(const int64_t* grouping, int64_t n) {
    std::vector<int64_t> s(grouping, grouping + n);
    std::sort(s.begin(), s.end());
    std::vector<int64_t> part;
    for (int64_t i = 0; i < n; i++)
        if (i == 0 ? (s[0] != -1) : (s[i] != s[i - 1])) part.push_back(i);
    // the reference starts from current_group = -1 (:29): a leading run of -1 ids opens no partition
    part.push_back(n);
    return part;
}

// src/general/utilities.cpp:45-54
std::vector<int64_t> get_fixed_size_grouping(const int64_t* grouping, int64_t n, int64_t row_block_size) {
    std::vector<int64_t> perm = get_permutation(grouping, n);
    std::vector<int64_t> out((size_t)n, -1);
    for (int64_t i = 0; i < n; i++) out[(size_t)perm[(size_t)i]] = i / row_block_size;
    return out;
}

float row_distance(int sim_measure, const int64_t* a, int64_t na, int64_t ga, const int64_t* b, int64_t nb, int64_t gb,
                   int64_t block_size) {
    auto blocks = [block_size](const int64_t* r, int64_t n) {
        std::vector<int32_t> out;
        int64_t last = -1;
        for (int64_t k = 0; k < n; k++) {
            int64_t q = r[k] / block_size;
            if (q != last) { out.push_back((int32_t)q); last = q; }
        }
        return out;
    };
    std::vector<int32_t> ba = blocks(a, na), bb = blocks(b, nb);
    int64_t inter = intersect_count(ba.data(), (int64_t)ba.size(), bb.data(), (int64_t)bb.size());
    int sim = (sim_measure & 1) ? SPARTA_SIM_JACCARD : SPARTA_SIM_HAMMING;
    return distance_from_counts(sim, na, (int64_t)ba.size(), ga, nb, (int64_t)bb.size(), gb, inter);
}

std::vector<int64_t> merge_rows(const int64_t* a, int64_t na, const int64_t* b, int64_t nb) {
    std::vector<int64_t> out;
    Pattern::merge_into(out, a, na, b, nb);
    return out;
}

// BlockingEngine::GetGrouping (src/general/blocking.cpp:633-676)
int reorder(const CsrView& a, const sparta_reorder_cfg& cfg, int64_t* grouping_out, sparta_reorder_stats* stats) {
    if (cfg.col_block_size <= 0) return fail(SPARTA_ERR_INVALID, "sparta_reorder: col_block_size must be > 0");
    const bool needs_rbs = cfg.blocking_algo == SPARTA_BLOCKING_FIXED_SIZE || cfg.blocking_algo == SPARTA_BLOCKING_ITERATIVE_MAX_SIZE ||
                           cfg.force_fixed_size;
    if (needs_rbs && cfg.row_block_size <= 0) return fail(SPARTA_ERR_INVALID, "sparta_reorder: row_block_size must be > 0");
    const bool iterative = cfg.blocking_algo == SPARTA_BLOCKING_ITERATIVE || cfg.blocking_algo == SPARTA_BLOCKING_ITERATIVE_CLOCKED ||
                           cfg.blocking_algo == SPARTA_BLOCKING_ITERATIVE_QUEUE || cfg.blocking_algo == SPARTA_BLOCKING_ITERATIVE_MAX_SIZE ||
                           cfg.blocking_algo == SPARTA_BLOCKING_ITERATIVE_STRUCTURED;
    if (cfg.blocking_algo == SPARTA_BLOCKING_ITERATIVE_STRUCTURED && (cfg.structured_m <= 0 || cfg.structured_n <= 0))
        return fail(SPARTA_ERR_INVALID, "sparta_reorder: structured_m and structured_n must be > 0 for blocking_algo 1");
    if (int rc = validate_csr(a, false)) return rc;

    auto t0 = clk::now();
    sparta_reorder_stats st{};
    const bool fast = iterative && all_rows_strictly_ascending(a);
    // algorithms 3 / 4: the near-linear exact form where it applies (Jaccard, weight 1, tau < 1, ascending rows, index built)
    auto run_clocked = [&](auto& c, int64_t* g) {
        if constexpr (std::is_same<std::decay_t<decltype(c)>, Ctx>::value) {
            const char* e = std::getenv("SPARTA_REORDER_SCALABLE");          // 0: never, 1: whenever it applies, unset: from 2048 rows
            const bool applies = c.have_index && c.sim == SPARTA_SIM_JACCARD && !cfg.use_groups && cfg.tau < 1.0f && a.rows <= INT32_MAX;
            if (applies && !(e && e[0] == '0')) { clocked_sparse(c, g); return; }
        }
        clocked(c, g);
    };
    auto run = [&](auto& c) {
        switch (cfg.blocking_algo) {
            case SPARTA_BLOCKING_ITERATIVE_MAX_SIZE: keeper(c, grouping_out); break;
            case SPARTA_BLOCKING_ITERATIVE: plain(c, grouping_out); break;
            case SPARTA_BLOCKING_ITERATIVE_STRUCTURED: structured_mn(c, grouping_out); break;
            default: run_clocked(c, grouping_out); break;
        }
        st.comparison_counter = c.comparisons; st.merge_counter = c.merges;
        if (cfg.blocking_algo != SPARTA_BLOCKING_ITERATIVE && cfg.blocking_algo != SPARTA_BLOCKING_ITERATIVE_STRUCTURED) {
            st.average_merge_tau = c.total_merge_tau / (float)c.merges;          // :239-240 (NaN when no merge, as in the reference)
            st.average_row_distance = c.total_row_distance / (float)c.merges;
        }
        st.timer_comparisons = c.t_cmp; st.timer_merges = c.t_merge;
    };
    switch (cfg.blocking_algo) {
        case SPARTA_BLOCKING_ITERATIVE_CLOCKED:
        case SPARTA_BLOCKING_ITERATIVE_QUEUE:
        case SPARTA_BLOCKING_ITERATIVE_MAX_SIZE:
        case SPARTA_BLOCKING_ITERATIVE_STRUCTURED:
        case SPARTA_BLOCKING_ITERATIVE: {
            if (fast) { Ctx c(a, cfg); run(c); }
            else { LiteralCtx c(a, cfg); run(c); }
            break;
        }
        case SPARTA_BLOCKING_MINHASH: {                                          // extension: LSH-bucketed clustering for large inputs
            if (a.rows > INT32_MAX) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_reorder: blocking_algo 7 handles up to 2^31 - 1 rows");
            if (!all_rows_strictly_ascending(a)) return fail(SPARTA_ERR_INVALID, "sparta_reorder: blocking_algo 7 needs strictly ascending columns in every row");
            Ctx c(a, cfg, false);
            minhash_lsh(c, grouping_out);
            st.comparison_counter = c.comparisons; st.merge_counter = c.merges;
            st.average_merge_tau = c.total_merge_tau / (float)c.merges;
            st.average_row_distance = c.total_row_distance / (float)c.merges;
            st.timer_comparisons = c.t_cmp; st.timer_merges = c.t_merge;
            break;
        }
        case SPARTA_BLOCKING_FIXED_SIZE:                                         // :554-562
            for (int64_t i = 0; i < a.rows; i++) grouping_out[i] = i / cfg.row_block_size;
            break;
        case SPARTA_BLOCKING_SCRAMBLE: {                                         // :565-574, seed 123
            std::vector<int64_t> g((size_t)a.rows);
            std::iota(g.begin(), g.end(), (int64_t)0);
            std::shuffle(g.begin(), g.end(), std::default_random_engine(123));
            std::copy(g.begin(), g.end(), grouping_out);
            break;
        }
        default:
            return fail(SPARTA_ERR_INVALID, "sparta_reorder: unknown blocking_algo " + std::to_string(cfg.blocking_algo));
    }
    if (cfg.force_fixed_size && cfg.blocking_algo != SPARTA_BLOCKING_FIXED_SIZE) {   // :670-673
        std::vector<int64_t> g = get_fixed_size_grouping(grouping_out, a.rows, cfg.row_block_size);
        std::copy(g.begin(), g.end(), grouping_out);
    }
    st.timer_total = us_since(t0);
    if (stats) *stats = st;
    return SPARTA_OK;
}

}  // namespace sparta
