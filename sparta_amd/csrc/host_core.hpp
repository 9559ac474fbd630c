// host_core.hpp -- internal declarations shared by the host-side C++ of libsparta_amd.so
// (reorder engine, VBS builder, C-ABI glue).  Not installed; the public surface is
// include/sparta_amd.h (C-ABI) and include/sparta_compat.hpp (reference-shaped C++ API).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <exception>
#include <string>
#include <thread>
#include <vector>
#include "sparta_amd.h"

namespace sparta {

// host threads for the builders (SPARTA_HOST_THREADS overrides).  Default: the CPUs this process may actually USE -- the hardware threads, capped by the cgroup's CPU quota
// (/sys/fs/cgroup/cpu.max, or cpu.cfs_quota_us / cpu.cfs_period_us under cgroup v1): a container that sees 256 hardware threads behind a quota of 16 CPUs builds a power-law part
// in 9.1 s on 256 threads and in 6.7 s on 16 (round 5, part 0 of configs[3] at 1 %: every lap of the builders -- they are memory-bound, and 256 runnable threads on 16 CPUs' worth
// of time only evict each other's cache lines).
inline int host_threads() {
    if (const char* e = std::getenv("SPARTA_HOST_THREADS")) return std::max(1, atoi(e));
    static const int n = [] {
        const unsigned hw = std::thread::hardware_concurrency();
        int t = (int)(hw ? hw : 1);
        double quota = 0.0, period = 0.0;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (std::fscanf(f, "%31s %lf", q, &period) == 2 && q[0] != 'm') quota = atof(q);
            std::fclose(f);
        } else {
            if (FILE* fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(fq, "%lf", &quota) != 1) quota = 0.0; std::fclose(fq); }
            if (FILE* fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(fp, "%lf", &period) != 1) period = 0.0; std::fclose(fp); }
        }
        if (quota > 0.0 && period > 0.0) t = std::min(t, std::max(1, (int)(quota / period + 0.5)));
        return t;
    }();
    return n;
}

// f(lo, hi, thread) over [0, n) in chunks of `grain` items handed out dynamically (an atomic cursor): block-rows of a power-law matrix
// differ by orders of magnitude in cost, static ranges leave one thread with the hubs.  f must only touch what belongs to its items.
template <typename F>
void parallel_for_dynamic(int64_t n, int64_t grain, F&& f) {
    if (n <= 0) return;
    grain = std::max<int64_t>(1, grain);
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(host_threads(), (n + grain - 1) / grain));
    if (nt <= 1) { f((int64_t)0, n, 0); return; }
    // An exception inside a worker (bad_alloc on inputs of 1e8..1e9 nonzeros is plausible) must reach the caller's handlers, not std::terminate: the first one is
    // kept, the cursor is pushed past the end so that the others stop, every started thread is joined -- also when a thread cannot be created -- and it is rethrown
    std::atomic<int64_t> cursor{0};
    std::exception_ptr first_error;
    std::atomic<bool> failed{false};
    auto work = [&](int t) {
        try {
            for (;;) {
                const int64_t lo = cursor.fetch_add(grain, std::memory_order_relaxed);
                if (lo >= n) break;
                f(lo, std::min(n, lo + grain), t);
            }
        } catch (...) {
            if (!failed.exchange(true)) first_error = std::current_exception();
            cursor.store(n, std::memory_order_relaxed);
        }
    };
    std::vector<std::thread> th;
    th.reserve((size_t)nt);
    try {
        for (int64_t t = 1; t < nt; t++) th.emplace_back(work, (int)t);
    } catch (...) {                                      // (std::system_error: no more threads) the caller's thread and the ones that started do the work
    }
    work(0);
    for (auto& x : th) x.join();
    if (failed.load()) std::rethrow_exception(first_error);
}

// SPARTA_BUILD_TRACE=1: phase timings of the host builders on stderr (developer aid)
struct BuildTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    const char* who;
    explicit BuildTrace(const char* w) : on(std::getenv("SPARTA_BUILD_TRACE") != nullptr), t0(std::chrono::steady_clock::now()), who(w) {}
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[sparta build] %s: %-28s %8.3f s\n", who, what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

// Flat CSR view (the reference keeps one heap array per row: include/matrices.h:22-28; we use the
// flat form everywhere and adapt at the compat layer).
struct CsrView {
    int64_t rows = 0, cols = 0;
    const int64_t* rowptr = nullptr;
    const int32_t* colidx = nullptr;
    const float* vals = nullptr;   // nullptr == pattern only
    int64_t nnz_of(int64_t i) const { return rowptr[i + 1] - rowptr[i]; }
    const int32_t* row(int64_t i) const { return colidx + rowptr[i]; }
};

// thread-local error message behind sparta_last_error()
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

// validates rowptr monotonicity, column range and strictly ascending columns per row
int validate_csr(const CsrView& a, bool need_sorted);

// ---- reorder engine (reorder.cpp) ----------------------------------------------------------
int reorder(const CsrView& a, const sparta_reorder_cfg& cfg, int64_t* grouping_out, sparta_reorder_stats* stats);
std::vector<int64_t> get_permutation(const int64_t* grouping, int64_t n);
std::vector<int64_t> get_partition(const int64_t* grouping, int64_t n);
std::vector<int64_t> get_fixed_size_grouping(const int64_t* grouping, int64_t n, int64_t row_block_size);
float row_distance(int sim_measure, const int64_t* a, int64_t na, int64_t ga, const int64_t* b, int64_t nb, int64_t gb,
                   int64_t block_size);
std::vector<int64_t> merge_rows(const int64_t* a, int64_t na, const int64_t* b, int64_t nb);

// ---- VBS builder (vbs_build.cpp) -------------------------------------------------------------
int vbs_build(const CsrView& a, const int64_t* grouping, int64_t col_block_size, int64_t row_block_size,
              bool force_fixed_size, sparta_vbs_host* out);
// Column-compacted ("union-pattern") tiles: what the reference's builder stores for a cluster at SMALL block widths -- exactly the columns its rows touch, back to
// back (src/general/vbr.cpp:177-228 at -b 1) -- kept as ONE dense (rows x |U|) tile + the ascending column list U per group of <= 64 consecutive reordered rows.
// The device multiplies a tile with the |U| gathered rows of B on the matrix cores (k_union.hip).  Built by vbs_build_hybrid for the block-rows whose rows share
// columns (mode 3), consumed by create_core (vbs_union.cpp lays the tiles out per worker in MFMA fragment order).
struct UnionPlanHost {
    // ty 0: tiles of <= 32 rows (one 32-row MFMA tile per wave and step), ty 1: tiles of 33..64 rows (two)
    struct Tile { int32_t c_row, mt; int64_t k0; int32_t nk, tail_e; int64_t tail0; };   // first reordered row (= row of C), rows, offset of its column list in cols[ty], columns; tail: below
    std::vector<Tile> tiles[2];
    std::vector<int32_t> cols[2];      // column ids, tile after tile, ascending inside a tile (a tile = the rows of ONE block-row: a chunk of <= 64 of them)
    std::vector<float> a[2];           // values, tile after tile, tile t at a_off[ty][t]: element (row i, list position k) at k * (32 * (ty + 1)) + i; zeros where a row lacks the column
    std::vector<int64_t> a_off[2];
    // the TAIL of a tile: up to union_tail_cap() nonzeros per row in columns too thinly used for the list -- a cluster's rows have a few columns of their own.  The kernel adds
    // them in the tile's epilogue, straight into the accumulators (one 128-byte piece of a row of B per entry and wave), instead of a sparse-row launch that re-reads and
    // re-writes the rows of C.  tail_e = entries per row (the longest row's, 0: none), entry e of row i at tail0 + e * (32 * (ty + 1)) + i; rows with fewer hold (col, 0.0f).
    std::vector<int32_t> tail_col[2];
    std::vector<float> tail_val[2];
    int64_t nnz = 0;                   // nonzeros held by the tiles (lists + tails)
    int64_t tail_nnz = 0;              // ... of which in the tails
    bool empty() const { return tiles[0].empty() && tiles[1].empty(); }
};

// hybrid build (sparta_vbs_create_from_csr): the block-rows left to the device's sparse-row path, as rows of (column, value)
struct HybridSparse {
    UnionPlanHost uni;                // flag 3 block-rows: their column-compacted tiles (the nonzeros of thinly used columns are sparse rows that ADD, as for flag 2)
    bool want_union = false;          // in: build them (SPARTA_UNION=0 switches the path off)
    int union_gran = 32;              // in: rows per MFMA row tile of the kernel that will multiply them (fp32 handles 16: v_mfma_f32_16x16x4_f32; 16-bit handles 32)
    std::vector<uint8_t> flag;        // per block-row: 1 = not in the dense image at all; 2 = mixed: its well-filled blocks are in the dense image,
                                      // the nonzeros of the others are sparse rows that ADD to what the tiles wrote (row_add)
    std::vector<uint8_t> row_add;     // per sparse row: 1 = its block-row also has tiles: the sparse-row kernels add to C instead of storing
    double esz = 4.0;                 // bytes per element of B on the device (what a nonzero costs against re-reading a row of C)
    std::vector<int64_t> rowptr;      // rows of the flagged block-rows, in reordered order
    std::vector<int32_t> col, crow;   // crow: reordered row index (= row of C)
    std::vector<float> val;
};
// what the hybrid builder WOULD build (sparta_vbs_plan_stats): decided by the same pass, nothing materialised
struct HybridStats {
    int64_t tile_blocks = 0, tile_area = 0;   // blocks kept as dense MFMA tiles, their stored elements (h x w each)
    double mfma_steps = 0.0;                  // steps those tiles cost (one <= 32-row tile x kdep columns of a block)
    int64_t sparse_nnz = 0, sparse_rows = 0;  // entries / rows left to the sparse-row kernels
    int64_t block_rows = 0, rows = 0;
    int64_t union_block_rows = 0, union_cols = 0, union_nnz = 0;   // block-rows kept as column-compacted tiles, their list entries (k columns), the nonzeros those hold
    double union_steps = 0.0;                 // 32-deep steps of 32-row MFMA tiles they cost
};
double union_tile_units(int64_t rows, int gran);   // MFMA cost of a tile of `rows` rows per 32-deep step, in 32-row tiles: rows rounded up to the kernel's row tile, / 32
double union_col_cost(double units);  // a list entry of a column-compacted tile of `units` (union_tile_units) in nonzeros of the sparse-row path
int32_t union_min_count(double units);  // fewest nonzeros of a part's rows that keep a column in its tile
int32_t union_tail_cap();           // most tail entries per row of a tile (SPARTA_UNION_TAIL, default 16; 0: no tails)
int vbs_build_hybrid(const CsrView& a, const int64_t* grouping, int64_t col_block_size, int64_t row_block_size, bool force_fixed_size,
                     double K, int64_t kdep, sparta_vbs_host* out, HybridSparse* sp, bool keep_order = false, HybridStats* stats_only = nullptr);
// fewest MFMA steps the nearly empty block-rows of a matrix must be worth before they leave the tiles for the sparse-row kernels
// (SPARTA_SPARSE_MIN_STEPS overrides; default 4096)
inline int64_t sparse_min_steps() { const char* e = std::getenv("SPARTA_SPARSE_MIN_STEPS"); return e ? atoll(e) : 4096; }
int blocking_info(const CsrView& a, const int64_t* grouping, int64_t col_block_size, int64_t* info_out, float* avg_height_out);

}  // namespace sparta
