// vbs_capi.cpp -- the device image of a VBS matrix (sparta_vbs) and the C-ABI entry points that create, multiply and destroy
// it: sparta_vbs_create* / sparta_vbs_spmm* / sparta_pack_blocks (include/sparta_amd.h).  Host code only: kernels live in the
// k_*.hip translation units and are reached through the launch functions of vbs_device.hpp.
//
// What it replaces in the reference: the CPU triple loop VBR::multiply (src/general/vbr.cpp:323-372)
// and the "one library GEMM per nonzero block" GPU back-ends (src/cuda/cuda_utilities.cpp:39-887,
// src/cuda/cutlass_bellpack_lib.cu:380-1019).  Not a translation of either: there is ONE fused kernel
// family, no vendor BLAS/SPARSE call, no per-block launch.
//
#include <memory>

#include <cmath>
#include "vbs_device.hpp"

using namespace sparta_dev;

// No exception leaves this file: the plan builder and the image builders grow large std::vectors (step lists, packed 16-bit A,
// sparse rows), so std::bad_alloc is a real outcome on 10^8..10^9-nonzero inputs and must come back as a status code.
#define SPARTA_GUARD_BEGIN try {
#define SPARTA_GUARD_END(name_)                                                                                       \
    }                                                                                                                  \
    catch (const std::bad_alloc&) { return sparta::fail(SPARTA_ERR_ALLOC, std::string(name_) + ": out of host memory"); } \
    catch (const std::exception& e) { return sparta::fail(SPARTA_ERR_INVALID, std::string(name_) + ": " + e.what()); }  \
    catch (...) { return sparta::fail(SPARTA_ERR_INVALID, std::string(name_) + ": unknown C++ exception"); }

namespace {

bool force_generic() { const char* e = std::getenv("SPARTA_FORCE_GENERIC"); return e && e[0] == '1'; }

// set for the duration of a product call whose stream is being captured into a graph: anything that allocates, synchronises or times
// (scratch growth, the first-call autotune, the gathered step lists) then fails with SPARTA_ERR_UNSUPPORTED instead of breaking the
// capture -- run the same call once outside the capture first (same n_cols, layouts and shard_rows), after which the call is launches only
thread_local bool g_capturing = false;
struct CaptureScope {
    explicit CaptureScope(hipStream_t st, bool device_ptrs) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        g_capturing = device_ptrs && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    }
    ~CaptureScope() { g_capturing = false; }
};
int capture_refusal(const char* what) {
    return sparta::fail(SPARTA_ERR_UNSUPPORTED, std::string("sparta_vbs_spmm: the stream is being captured and this call still has to ") + what +
                                                 " -- run it once outside the capture first");
}

int ensure_scratch(void** ptr, size_t* have, size_t need) {
    if (*have >= need) return SPARTA_OK;
    if (g_capturing) return capture_refusal("allocate scratch memory");
    if (*ptr) { (void)hipFree(*ptr); *ptr = nullptr; *have = 0; }
    HIP_TRY(hipMalloc(ptr, need));
    *have = need;
    return SPARTA_OK;
}

void destroy_impl(sparta_vbs* v) {
    if (!v) return;
    DeviceGuard g(v->device);
    if (v->d_A) (void)hipFree(v->d_A);
    if (v->d_jab) (void)hipFree(v->d_jab);
    for (int c = 0; c < 4; c++)
        if (v->d_tiles[c]) (void)hipFree(v->d_tiles[c]);
    if (v->d_brows) (void)hipFree(v->d_brows);
    for (int ty = 0; ty < 2; ty++) {
        if (v->d_steps[ty]) (void)hipFree(v->d_steps[ty]);
        if (v->d_steps_g[ty]) (void)hipFree(v->d_steps_g[ty]);
        if (v->d_wrange[ty]) (void)hipFree(v->d_wrange[ty]);
    }
    if (v->d_a_frag) (void)hipFree(v->d_a_frag);
    if (v->d_hub_steps) (void)hipFree(v->d_hub_steps);
    if (v->d_hub_steps_g) (void)hipFree(v->d_hub_steps_g);
    if (v->d_hub_tiles) (void)hipFree(v->d_hub_tiles);
    if (v->d_hub_wrange) (void)hipFree(v->d_hub_wrange);
    if (v->d_hub_A) (void)hipFree(v->d_hub_A);
    if (v->d_fix) (void)hipFree(v->d_fix);
    if (v->d_fix_slots) (void)hipFree(v->d_fix_slots);
    if (v->d_big_fix) (void)hipFree(v->d_big_fix);
    if (v->d_ws) (void)hipFree(v->d_ws);
    if (v->d_btail) (void)hipFree(v->d_btail);
    if (v->d_tune) (void)hipFree(v->d_tune);
    if (v->d_B16) (void)hipFree(v->d_B16);
    if (v->d_Bt) (void)hipFree(v->d_Bt);
    if (v->d_Ct) (void)hipFree(v->d_Ct);
    if (v->d_clk) (void)hipFree(v->d_clk);
    if (v->tev0) (void)hipEventDestroy(v->tev0);
    if (v->tev1) (void)hipEventDestroy(v->tev1);
    if (v->d_sp_rowptr) (void)hipFree(v->d_sp_rowptr);
    if (v->d_sp_col) (void)hipFree(v->d_sp_col);
    if (v->d_sp_val) (void)hipFree(v->d_sp_val);
    if (v->d_sp_crow) (void)hipFree(v->d_sp_crow);
    if (v->d_sp_list) (void)hipFree(v->d_sp_list);
    if (v->d_sp_segs) (void)hipFree(v->d_sp_segs);
    if (v->d_sp_stream_begin) (void)hipFree(v->d_sp_stream_begin);
    if (v->d_sp_long) (void)hipFree(v->d_sp_long);
    if (v->d_sp_part) (void)hipFree(v->d_sp_part);
    if (v->d_cr_col) (void)hipFree(v->d_cr_col);
    if (v->d_cr_val) (void)hipFree(v->d_cr_val);
    if (v->d_cr_meta) (void)hipFree(v->d_cr_meta);
    if (v->d_cr_parts) (void)hipFree(v->d_cr_parts);
    if (v->d_cr_mode) (void)hipFree(v->d_cr_mode);
    if (v->cr_small.col) (void)hipFree(v->cr_small.col);
    if (v->cr_small.val) (void)hipFree(v->cr_small.val);
    if (v->cr_small.meta) (void)hipFree(v->cr_small.meta);
    if (v->cr_small.dest) (void)hipFree(v->cr_small.dest);
    if (v->cr_small.longs) (void)hipFree(v->cr_small.longs);
    if (v->cr_small.parts) (void)hipFree(v->cr_small.parts);
    if (v->d_cr_dest) (void)hipFree(v->d_cr_dest);
    if (v->d_cr_longs) (void)hipFree(v->d_cr_longs);
    for (int ty = 0; ty < kUnionTypes; ty++) {
        if (v->d_u_rec[ty]) (void)hipFree(v->d_u_rec[ty]);
        if (v->d_u_ids[ty]) (void)hipFree(v->d_u_ids[ty]);
        if (v->d_u_a[ty]) (void)hipFree(v->d_u_a[ty]);
        if (v->d_u_wrange[ty]) (void)hipFree(v->d_u_wrange[ty]);
        if (v->d_u_tail[ty]) (void)hipFree(v->d_u_tail[ty]);
    }
    if (v->d_Brm) (void)hipFree(v->d_Brm);
    if (v->d_B) (void)hipFree(v->d_B);
    if (v->d_C) (void)hipFree(v->d_C);
    if (v->ev0) (void)hipEventDestroy(v->ev0);
    if (v->ev1) (void)hipEventDestroy(v->ev1);
    for (int c = 0; c < 4; c++)
        for (int e = 0; e < 2; e++)
            if (v->cev[c][e]) (void)hipEventDestroy(v->cev[c][e]);
    delete v;
}




// ---- resident-column product (k_colres.hip): A as length-sorted slots, 64 to a slice -----------------------------------------------------------
// Only for handles whose rows of C are ALL sparse rows (nothing for a tile launch to store first) and small enough that a column of B (and of C, with the
// extra cells of the long rows) fits LDS.  SPARTA_COLRES=0: never built.  SPARTA_COLRES_LMAX: longest slot (default max(32, nnz / 2048): at 1024 lanes per
// workgroup a slot of twice the average load per lane is not the critical path).
constexpr int64_t kColresCells = 160 * 1024 / 4;                  // floats of LDS a workgroup may hold
struct ColresHost {
    std::vector<uint16_t> col;                                    // per batch of 4 steps and lane: four columns (relative to their K range) ...
    std::vector<float> val;                                       // ... and four values (empty: every stored value is 1.0f -- a unit image, the reference's -P 1)
    std::vector<ColresPartDev> parts;
    std::vector<int32_t> meta, dest;                              // per part: see ColresPartDev
    std::vector<uint8_t> mode;                                    // per row of C (padded to 4): 0 not this kernel's, 1 store, 2 add
    std::vector<ColresLong> longs;
    std::vector<int32_t> krange;                                  // [n_ranges + 1]
    int32_t max_slices = 0, max_cells = 0, lmax = 0;              // slices of the part with most; cells a column set needs in LDS (largest range of B + its zero cell / largest staging image)
    int64_t entries = 0;
};
// force_parts > 1: that many parts of the rows of C (about equal staging images), one K range -- the image for products of few column sets
bool build_colres(int64_t rows, int64_t cols, const std::vector<int64_t>& rowptr, const std::vector<int32_t>& col, const std::vector<float>& val,
                  const std::vector<int32_t>& crow, ColresHost& H, int force_parts = 0) {
    if (const char* e = std::getenv("SPARTA_COLRES")) if (atoi(e) == 0) return false;
    const int64_t n = (int64_t)crow.size(), nnz = rowptr.empty() ? 0 : rowptr.back();
    // The sparse rows may be SOME of the rows of C (the others belong to block-rows of MFMA tiles, whose launches come first and which this kernel leaves alone), and a sparse row may belong to
    // a MIXED block-row (bit 31 of crow: its well-filled blocks are tiles, this kernel ADDS the rest): H.mode says per row of C.  Not worth a second image and B / C through LDS for a
    // handful of sparse rows: at least half the rows.
    if (n > rows || 2 * n < rows || nnz == 0 || rows > kColresMaxParts * kColresCells || cols > kColresMaxRanges * (kColresCells - 8)) return false;
    std::vector<int32_t> ord_of((size_t)rows, -1);                                    // sparse row of every row of C (-1: none)
    H.mode.assign((size_t)((rows + 3) / 4 * 4), 0);                                    // 0: not this kernel's row, 1: store, 2: add to what the tile launches stored
    for (int64_t t = 0; t < n; t++) {
        const int32_t r = crow[(size_t)t] & 0x7fffffff;
        if (r >= rows || ord_of[(size_t)r] >= 0) return false;                         // not distinct rows of C
        ord_of[(size_t)r] = (int32_t)t;
        H.mode[(size_t)r] = crow[(size_t)t] < 0 ? 2 : 1;
    }
    int64_t lmax = std::max<int64_t>(32, nnz / 2048);
    if (const char* e = std::getenv("SPARTA_COLRES_LMAX")) lmax = std::max(1, atoi(e));
    constexpr int64_t kWaves = kColresWaves;
    // ---- K ranges: the rows of B a workgroup holds at a time (+ the zero cell), equal, multiples of 4 ----
    int64_t max_range = kColresCells - 8;
    if (const char* e = std::getenv("SPARTA_COLRES_RANGE")) max_range = std::max<int64_t>(64, std::min<int64_t>(max_range, atoi(e)) / 4 * 4);      // (tests: small matrices in several ranges)
    const int64_t n_ranges = (cols + max_range - 1) / max_range;
    if (n_ranges > kColresMaxRanges) return false;
    const int64_t range_len = ((cols + n_ranges - 1) / n_ranges + 3) / 4 * 4;
    H.krange.assign((size_t)n_ranges + 1, 0);
    for (int64_t r = 0; r <= n_ranges; r++) H.krange[(size_t)r] = (int32_t)std::min(cols, r * range_len);
    // ---- parts: contiguous ranges of the rows of C whose staging image (rows + the extra cells of their chunked rows) fits ----
    int64_t max_plane = kColresCells;
    if (const char* e = std::getenv("SPARTA_COLRES_PLANE")) max_plane = std::max<int64_t>(64, std::min<int64_t>(max_plane, atoi(e)) / 4 * 4);
    auto chunks_of = [&](int64_t r) -> int64_t {
        const int64_t t = ord_of[(size_t)r];
        if (t < 0) return 1;                                                           // (a row of another kernel: a cell of the staging image nobody writes or reads)
        return std::max<int64_t>(1, (rowptr[(size_t)t + 1] - rowptr[(size_t)t] + lmax - 1) / lmax);
    };
    if (force_parts > 1) {
        int64_t total_cells = 0;
        for (int64_t r = 0; r < rows; r++) total_cells += chunks_of(r);
        max_plane = std::min<int64_t>(max_plane, ((total_cells + force_parts - 1) / force_parts + 256 + 3) / 4 * 4);
    }
    std::vector<int64_t> part_begin{0};
    {
        int64_t cells = 0;
        for (int64_t r = 0; r < rows; r += 4) {                                       // (four rows at a time: every part starts at a multiple of 4 -- 16-byte stores of C)
            int64_t add = 4;
            for (int64_t q = r; q < std::min(rows, r + 4); q++) add += chunks_of(q) - 1;
            if (add > max_plane) return false;
            if (cells + add > max_plane) { part_begin.push_back(r); cells = 0; }
            cells += add;
        }
        part_begin.push_back(rows);
    }
    const int64_t n_parts = (int64_t)part_begin.size() - 1;
    if (n_parts > kColresMaxParts) return false;
    // Several parts / ranges are built only on request (SPARTA_COLRES_CUTS=1; tested, bit-identical to the uncut image): measured on the reference's two larger real matrices at N = 8192
    // they LOSE to the windowed row gather -- social_location (58 k x 58 k, 3.7 nonzeros per row: 2 parts x 2 ranges) 4.26 against 2.60 ms, ia-wikiquote (21.6 k x 94 k: 3 ranges) 3.25
    // against 2.75: one column per workgroup (4-byte LDS reads, A streamed N times), and rows of 2-4 nonzeros spread over the ranges pad every slice to a batch per range (3.3 x the nonzeros).
    {
        const char* e = std::getenv("SPARTA_COLRES_CUTS");
        if (force_parts > 1) { if (n_ranges != 1 || n_parts < 2) return false; }
        else if (n_parts * n_ranges > 1 && !(e && atoi(e) != 0)) return false;
    }
    bool unit = true;
    for (int64_t k = 0; k < nnz && unit; k++) unit = val[(size_t)k] == 1.0f;
    if (const char* e = std::getenv("SPARTA_COLRES_UNIT")) unit = unit && atoi(e) != 0;      // (0: keep the value array of a pattern matrix -- developer A/B)
    struct Slot { int64_t p0; int32_t len, dest; };
    int64_t total = 0;                                                                // batches so far (all parts, ranges, waves)
    struct Fill { int64_t batch0; int32_t wb; };                                      // where a (sorted slice, range) starts and how wide it is
    for (int64_t p = 0; p < n_parts; p++) {
        const int64_t r0 = part_begin[(size_t)p], r1 = part_begin[(size_t)p + 1], prow = r1 - r0, rows_pad = (prow + 3) / 4 * 4;
        std::vector<Slot> slots;
        const size_t longs0 = H.longs.size();
        int64_t n_extra = 0;
        for (int64_t r = r0; r < r1; r++) {
            const int64_t t = ord_of[(size_t)r];
            if (t < 0) continue;
            const int64_t p0 = rowptr[(size_t)t], len = rowptr[(size_t)t + 1] - p0, chunks = chunks_of(r);
            if (chunks > 1) H.longs.push_back(ColresLong{(int32_t)(r - r0), (int32_t)(rows_pad + n_extra), (int32_t)(chunks - 1), 0});
            for (int64_t c = 0; c < chunks; c++) {
                const int64_t o = c * lmax;
                slots.push_back(Slot{p0 + o, (int32_t)std::max<int64_t>(0, std::min<int64_t>(lmax, len - o)), c == 0 ? (int32_t)(r - r0) : (int32_t)(rows_pad + n_extra + c - 1)});
            }
            n_extra += chunks - 1;
        }
        const int64_t plane = (rows_pad + n_extra + 3) / 4 * 4;
        if (plane > kColresCells || (int64_t)slots.size() > (int64_t)colres_max_slices(1) * 64) return false;
        std::stable_sort(slots.begin(), slots.end(), [](const Slot& a, const Slot& b) { return a.len > b.len; });
        const int64_t n_slices = ((int64_t)slots.size() + 63) / 64;
        // where the entries of a slot fall into the K ranges
        std::vector<int32_t> cut((size_t)slots.size() * (size_t)(n_ranges + 1), 0);
        for (size_t q = 0; q < slots.size(); q++) {
            const int32_t* c0 = col.data() + slots[q].p0;
            for (int64_t r = 0; r <= n_ranges; r++)
                cut[q * (size_t)(n_ranges + 1) + (size_t)r] = (int32_t)(std::lower_bound(c0, c0 + slots[q].len, H.krange[(size_t)r]) - c0);
        }
        ColresPartDev pd;
        pd.r0 = (int32_t)r0; pd.rows = (int32_t)prow; pd.n_slices = (int32_t)n_slices; pd.n_long = (int32_t)(H.longs.size() - longs0); pd.plane = (int32_t)plane;
        pd.meta = (int32_t)H.meta.size(); pd.dest = (int32_t)H.dest.size(); pd.longs = (int32_t)longs0;
        pd.all_store = 1; pd.pad = 0;
        for (int64_t r = r0; r < r1; r++) if (H.mode[(size_t)r] != 1) pd.all_store = 0;
        const size_t m_wslice = H.meta.size(), m_woff = m_wslice + 17, m_bnd = m_woff + (size_t)n_ranges * 17;
        H.meta.resize(m_bnd + (size_t)n_ranges * (size_t)n_slices, 0);
        const size_t d0 = H.dest.size();
        H.dest.resize(d0 + (size_t)n_slices * 64, -1);
        // wave-major order: wave w owns the slices w, w + 16, ... of the sorted list (equally long streams)
        std::vector<int64_t> at_slice((size_t)n_slices, 0);
        {
            int64_t k_slice = 0;
            for (int64_t w = 0; w < kWaves; w++) {
                H.meta[m_wslice + (size_t)w] = (int32_t)k_slice;
                for (int64_t s = w; s < n_slices; s += kWaves) at_slice[(size_t)s] = k_slice++;
            }
            H.meta[m_wslice + (size_t)kWaves] = (int32_t)k_slice;
        }
        std::vector<Fill> fill((size_t)n_slices * (size_t)n_ranges);
        for (int64_t r = 0; r < n_ranges; r++) {
            for (int64_t w = 0; w < kWaves; w++) {
                H.meta[m_woff + (size_t)r * 17 + (size_t)w] = (int32_t)total;
                const int64_t first = total;
                for (int64_t s = w; s < n_slices; s += kWaves) {
                    int32_t longest = 0;
                    for (size_t q = (size_t)s * 64; q < std::min(slots.size(), (size_t)s * 64 + 64); q++)
                        longest = std::max(longest, cut[q * (size_t)(n_ranges + 1) + (size_t)r + 1] - cut[q * (size_t)(n_ranges + 1) + (size_t)r]);
                    const int32_t wb = std::max(1, (longest + 3) / 4);                  // never an empty slice: the kernel parks a slice's sums behind its last batch
                    fill[(size_t)s * (size_t)n_ranges + (size_t)r] = Fill{total, wb};
                    total += wb;
                    H.meta[m_bnd + (size_t)r * (size_t)n_slices + (size_t)at_slice[(size_t)s]] = (int32_t)(total - first);
                    if (total * 256 > (int64_t)INT32_MAX / 2) return false;
                }
            }
            H.meta[m_woff + (size_t)r * 17 + (size_t)kWaves] = (int32_t)total;
        }
        // the entries.  Past a slot's end in a range: the range's length -- one more cell of LDS, which the kernel clears -- with the value 0.0f: no per-lane condition, and nothing of B
        // is multiplied by a padding zero
        H.col.resize((size_t)(total + 1) * 256, 0);                                     // (+ one batch: a wave without slices still reads the first line of its empty stream)
        if (!unit) H.val.resize((size_t)(total + 1) * 256, 0.0f);
        for (int64_t s = 0; s < n_slices; s++) {
            for (int64_t r = 0; r < n_ranges; r++) {
                const Fill f = fill[(size_t)s * (size_t)n_ranges + (size_t)r];
                const uint16_t pad = (uint16_t)(H.krange[(size_t)r + 1] - H.krange[(size_t)r]);
                for (size_t x = (size_t)f.batch0 * 256; x < (size_t)(f.batch0 + f.wb) * 256; x++) H.col[x] = pad;
                for (int64_t l = 0; l < 64; l++) {
                    const size_t q = (size_t)s * 64 + (size_t)l;
                    if (q >= slots.size()) continue;
                    const int32_t b = cut[q * (size_t)(n_ranges + 1) + (size_t)r], e = cut[q * (size_t)(n_ranges + 1) + (size_t)r + 1];
                    for (int32_t k = b; k < e; k++) {
                        const size_t at = ((size_t)(f.batch0 + (k - b) / 4) * 64 + (size_t)l) * 4 + (size_t)((k - b) % 4);
                        H.col[at] = (uint16_t)(col[(size_t)(slots[q].p0 + k)] - H.krange[(size_t)r]);
                        if (!unit) H.val[at] = val[(size_t)(slots[q].p0 + k)];
                    }
                }
            }
        }
        for (size_t q = 0; q < slots.size(); q++) H.dest[d0 + (size_t)at_slice[q / 64] * 64 + q % 64] = slots[q].dest;
        H.parts.push_back(pd);
        H.max_slices = std::max(H.max_slices, (int32_t)n_slices);
        H.max_cells = std::max(H.max_cells, (int32_t)plane);
    }
    for (size_t x = (size_t)total * 256; x < H.col.size(); x++) H.col[x] = (uint16_t)(H.krange[1] - H.krange[0]);
    // a wave that owns no slice of a part (every part of fewer than 16 slices has such waves) still prefetches the destination cells of "its first slice" = one
    // slice behind the part's last (k_colres.hip: dst[I] = dest[(sl0 + ...) * 64 + lane]); behind the LAST part that is past the table: one padding slice of -1
    H.dest.resize(H.dest.size() + 64, -1);
    H.max_cells = std::max(H.max_cells, (int32_t)((range_len + 4) / 4 * 4));
    H.entries = total * 256;
    H.lmax = (int32_t)lmax;
    return true;
}
// columns of B per workgroup: as many as LDS holds next to each other (columns of B first, the staging image of C after them, in the same cells)
// ... as many as fit: with one workgroup per CU the phases (columns of B in, stream of A, columns of C out) take the sum of their times, and the stream of A costs the same
// per workgroup whatever NC (measured): fewer passes win.  SPARTA_COLRES_NC caps it (read per call: developer A/B, tests).
int colres_columns(const sparta_vbs_t* A, int n_cols) {
    if (A->cr_slices == 0) return 0;
    const int64_t cells = A->cr_plane;                             // what a column set needs in LDS: the largest range of B (+ its zero cell) or the largest staging image
    const int fit = (int)std::min<int64_t>(std::min<int64_t>(4, kColresCells / cells), n_cols);
    int nc = fit;
    // (a product of fewer workgroups than CUs is ONE round whatever NC: the stream of A with three or four columns per cell costs twice the LDS time of one or two --
    // bcsstk18 at N = 128: 16.5 / 14.7 / 17.9 us with 1 / 2 / 3 columns)
    if (nc > 2 && (int64_t)((n_cols + 1) / 2) * A->cr_parts <= 256) nc = 2;             // (one round with two columns as well)
    else if (nc > 1) {
        // several rounds of workgroups: a round costs ~13 us + 3.4 us per column (measured per workgroup on the reference's real matrices: 16.5 / 18.7 / 23.4 / 27 us with 1..4
        // columns), and a last round is paid in full however few workgroups it has: N = 1024 on 256 CUs is 2 full rounds with two columns, 1.34 -> 2 with three
        // (bcsstk18 0.045 against 0.052 ms)
        int best = nc;
        double best_t = 1e300;
        for (int c = 1; c <= fit; c++) {
            const double rounds = std::ceil(std::ceil((double)n_cols / c) * A->cr_parts / 256.0), t = rounds * (13.0 + 3.4 * c);
            if (t <= best_t) { best_t = t; best = c; }
        }
        nc = best;
    }
    if (const char* e = std::getenv("SPARTA_COLRES_NC")) nc = std::min(fit, std::max(1, atoi(e)));
    while (nc > 1 && A->cr_slices > colres_max_slices(nc)) nc--;
    return nc >= 1 && A->cr_slices <= colres_max_slices(nc) ? nc : 0;
}

}  // namespace

extern "C" {

int sparta_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// `ext` (sparta_vbs_create_from_csr): the sparse-row part of the matrix decided and collected by the hybrid host builder.  Fully sparse
// block-rows (ext->flag 1) have nzcount = 0 in the arrays given here and get neither tiles nor zero-fill records; MIXED block-rows (flag 2)
// keep their well-filled blocks in nzcount / jab / mab as tiles like any other, and the nonzeros of their other blocks are sparse rows that
// add to C behind the tile launches (see the ordering note at the sparse-row launch in spmm_impl).
static int create_core(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t w, const int64_t* row_part,
                       const int64_t* nzcount, const int64_t* jab, const float* mab, int64_t br0, int64_t br1, int32_t dtype,
                       int32_t device, const sparta::HybridSparse* ext) {
    using sparta::fail;
    if (!out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: out is NULL");
    *out = nullptr;
    sparta::BuildTrace trace("vbs_create");
    if (rows <= 0 || cols <= 0 || block_rows <= 0 || w <= 0 || !row_part || !nzcount)
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: bad dimensions or NULL index array");
    if (br0 < 0 || br1 > block_rows || br0 >= br1) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: bad block-row range");
    if (dtype != SPARTA_F32 && dtype != SPARTA_F16 && dtype != SPARTA_BF16) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: bad dtype");
    const bool h16 = dtype != SPARTA_F32;
    if (h16 && w % 32 != 0)
        return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: SPARTA_F16 / SPARTA_BF16 need block_col_size % 32 == 0 (only the stream kernels have a 16-bit form)");
    if (rows > INT32_MAX || w > (1 << 20)) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: rows >= 2^31 or w > 2^20");
    const int64_t block_cols = (cols - 1) / w + 1;

    // validate the partition and locate the range inside jab / mab
    if (row_part[0] != 0 || row_part[block_rows] != rows) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: row_part must span [0, rows]");
    int64_t jab_lo = 0, mab_lo = 0, jab_hi = 0, mab_hi = 0, jo = 0, mo = 0;
    for (int64_t ib = 0; ib < block_rows; ib++) {
        const int64_t h = row_part[ib + 1] - row_part[ib];
        if (h < 0 || nzcount[ib] < 0 || nzcount[ib] > block_cols) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: invalid row_part / nzcount");
        if (h > INT32_MAX / 2 || nzcount[ib] > INT32_MAX) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: block-row too large");
        if (ib == br0) { jab_lo = jo; mab_lo = mo; }
        jo += nzcount[ib];
        mo += nzcount[ib] * h * w;
        if (ib == br1 - 1) { jab_hi = jo; mab_hi = mo; }
    }
    const int64_t nblocks = jab_hi - jab_lo, nztot = mab_hi - mab_lo;
    if (nblocks > 0 && (!jab || !mab)) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: jab / mab is NULL");

    int ndev = sparta_device_count();
    const bool plan_debug = std::getenv("SPARTA_PLAN_DEBUG") != nullptr && ndev <= 0;   // developer aid: print the plan statistics on a host without a GPU, then fail as usual
    if (ndev <= 0 && !plan_debug) return fail(SPARTA_ERR_NO_DEVICE, "sparta_vbs_create: no HIP device visible (this path has no CPU fallback)");
    if (!plan_debug && (device < 0 || device >= ndev)) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: device index out of range");

    // ---- sparse-row path: which block-rows are better served as rows of (column, value) --------------------------------
    // An MFMA step (one <=32-row tile x 32 k x 128 columns) takes the time of ~12 nonzeros on the sparse-row path (2.1 ns per
    // step across the 512 workers vs 512 B of B per nonzero and 128-column slab at ~3 TB/s): a block-row whose blocks hold
    // fewer than SPARTA_SPARSE_K (default 24) nonzeros per step goes there (measured break-even, scripts/sparse_k_sweep.py:
    // ~60 nonzeros per step while B fits the L2s, ~15 when its rows come from HBM).  SPARTA_SPARSE_K=0 switches the path off.
    std::vector<uint8_t> sparse_flag;
    std::vector<int64_t> sp_rowptr;
    std::vector<int32_t> sp_col, sp_crow, sp_list;
    std::vector<float> sp_val;
    std::vector<SpSegRec> sp_segs;
    std::vector<SpLongRec> sp_long;
    // a wave keeps SP_BATCH (16) rows of B in flight: a row of n nonzeros takes ~n / 16 memory latencies whatever else the GPU is doing, so rows
    // longer than kSpLong are cut into kSpSeg-nonzero segments that run on different waves (SPARTA_SPARSE_SEG overrides kSpSeg)
    // The segment length follows the size of the sparse part (decided below, once it is known): short segments keep a small
    // problem parallel (R-MAT 2^16: 128 -> 148 us, 512 -> 187 us), long ones save partial rows on a large one (2^20: 3.99 vs 3.75 ms).
    int64_t kSpSeg = 0;
    if (const char* e = std::getenv("SPARTA_SPARSE_SEG")) kSpSeg = std::max(8, atoi(e));
    int64_t n_sp_short = 0, n_sp_long = 0;
    // 16-bit handles: the values the kernels multiply are the ROUNDED ones (a value that rounds to zero is a zero)
    const bool bf16h = dtype == SPARTA_BF16;
    auto stored = [&](float x) -> float {
        if (!h16) return x;
        const uint16_t u = to_h16(x, bf16h);
        if (bf16h) { const uint32_t v32 = (uint32_t)u << 16; float f; std::memcpy(&f, &v32, 4); return f; }
        _Float16 hh; std::memcpy(&hh, &u, 2); return (float)hh;
    };
    if (ext) {
        // flag 1: the block-row has no tile at all (skipped by the MFMA plans); flag 2 (mixed): its well-filled blocks are tiles like any other
        // (nzcount / jab hold only those), the nonzeros of its other blocks are sparse rows that ADD to C (bit 31 of their crow entry)
        sparse_flag.assign(ext->flag.size(), 0);
        bool any = false;
        // (flag 3: column-compacted tiles, UnionPlanHost -- no w-wide block either; their thinly used columns are sparse rows that add, as for flag 2)
        for (size_t q = 0; q < ext->flag.size(); q++) { sparse_flag[q] = ext->flag[q] == 1 || ext->flag[q] == 3; any = any || sparse_flag[q]; }
        if (!any) sparse_flag.clear();
        // the entries that survive the storage type (a value that rounds to zero is a zero) and lie inside the matrix: counted, then
        // copied, on all host threads (10^8..10^9 entries on the power-law configs)
        const int64_t n_ext = (int64_t)ext->crow.size();
        std::vector<int64_t> keep((size_t)n_ext + 1, 0);
        sparta::parallel_for_dynamic(n_ext, 2048, [&](int64_t lo, int64_t hi, int) {
            for (int64_t t = lo; t < hi; t++) {
                int64_t c = 0;
                for (int64_t k = ext->rowptr[(size_t)t]; k < ext->rowptr[(size_t)t + 1]; k++)
                    c += stored(ext->val[(size_t)k]) != 0.0f && ext->col[(size_t)k] < cols;
                keep[(size_t)t + 1] = c;
            }
        });
        for (int64_t t = 0; t < n_ext; t++) keep[(size_t)t + 1] += keep[(size_t)t];
        sp_rowptr.swap(keep);
        sp_col.resize((size_t)sp_rowptr.back()); sp_val.resize((size_t)sp_rowptr.back()); sp_crow.resize((size_t)n_ext);
        sparta::parallel_for_dynamic(n_ext, 2048, [&](int64_t lo, int64_t hi, int) {
            for (int64_t t = lo; t < hi; t++) {
                int64_t o = sp_rowptr[(size_t)t];
                for (int64_t k = ext->rowptr[(size_t)t]; k < ext->rowptr[(size_t)t + 1]; k++) {
                    const float a = stored(ext->val[(size_t)k]);
                    if (a != 0.0f && ext->col[(size_t)k] < cols) { sp_col[(size_t)o] = ext->col[(size_t)k]; sp_val[(size_t)o] = a; o++; }
                }
                sp_crow[(size_t)t] = ext->crow[(size_t)t] | ((size_t)t < ext->row_add.size() && ext->row_add[(size_t)t] ? (int32_t)0x80000000 : 0);
            }
        });
    } else {
        double K = 24.0;
        if (const char* e = std::getenv("SPARTA_SPARSE_K")) K = atof(e);
        if (K > 0.0) {
            sparse_flag.assign((size_t)(br1 - br0), 0);
            int64_t jo2 = 0, mo2 = 0, n_flagged = 0;
            const int64_t row0 = row_part[br0];
            std::vector<int64_t> cnt;
            sp_rowptr.push_back(0);
            // pass 1: which block-rows qualify and what they would cost as tiles (a handful stays with the tiles: see vbs_build_hybrid)
            std::vector<uint8_t> qualifies((size_t)(br1 - br0), 0);
            {
                double steps_saved = 0.0;
                int64_t mo1 = 0;
                for (int64_t ib = br0; ib < br1; ib++) {
                    const int64_t h = row_part[ib + 1] - row_part[ib], nb = nzcount[ib];
                    const float* blk = mab + mab_lo + mo1;
                    const int64_t n_el = nb * h * w;
                    int64_t nnz = 0;
                    for (int64_t q = 0; q < n_el; q++) nnz += stored(blk[q]) != 0.0f;
                    const int64_t kdep = h16 && w % 64 == 0 ? 64 : 32;                 // k depth of a step of the kernels this handle would use
                    const double steps_br = (double)nb * (double)((w + kdep - 1) / kdep) * (double)((h + 31) / 32);
                    if (h > 0 && nb > 0 && (double)nnz < K * steps_br) { qualifies[(size_t)(ib - br0)] = 1; steps_saved += steps_br; }
                    mo1 += n_el;
                }
                if (steps_saved < (double)sparta::sparse_min_steps()) std::fill(qualifies.begin(), qualifies.end(), 0);
            }
            for (int64_t ib = br0; ib < br1; ib++) {
                const int64_t h = row_part[ib + 1] - row_part[ib], nb = nzcount[ib];
                const float* blk = mab + mab_lo + mo2;
                const int64_t n_el = nb * h * w;
                if (qualifies[(size_t)(ib - br0)] && (int64_t)sp_col.size() + n_el < ((int64_t)1 << 40)) {
                    sparse_flag[(size_t)(ib - br0)] = 1;
                    n_flagged++;
                    // rows of this block-row: (column, value) in the reference's order (blocks ascending, k ascending: vbr.cpp:358-363)
                    cnt.assign((size_t)h, 0);
                    for (int64_t b = 0; b < nb; b++)
                        for (int64_t k = 0; k < w; k++)
                            for (int64_t i = 0; i < h; i++) cnt[(size_t)i] += (stored(blk[(b * w + k) * h + i]) != 0.0f) && jab[jab_lo + jo2 + b] * w + k < cols;
                    const size_t base_row = sp_crow.size();
                    for (int64_t i = 0; i < h; i++) {
                        sp_crow.push_back((int32_t)(row_part[ib] - row0 + i));
                        sp_rowptr.push_back(sp_rowptr.back() + cnt[(size_t)i]);
                    }
                    sp_col.resize((size_t)sp_rowptr.back());
                    sp_val.resize((size_t)sp_rowptr.back());
                    for (int64_t i = 0; i < h; i++) cnt[(size_t)i] = sp_rowptr[base_row + (size_t)i];
                    for (int64_t b = 0; b < nb; b++) {
                        const int64_t c0 = jab[jab_lo + jo2 + b] * w;
                        for (int64_t k = 0; k < w && c0 + k < cols; k++)
                            for (int64_t i = 0; i < h; i++) {
                                const float a = stored(blk[(b * w + k) * h + i]);
                                if (a != 0.0f) { sp_col[(size_t)cnt[(size_t)i]] = (int32_t)(c0 + k); sp_val[(size_t)cnt[(size_t)i]++] = a; }
                            }
                    }
                }
                jo2 += nb;
                mo2 += n_el;
            }
            if (n_flagged == 0) sparse_flag.clear();
        }
    }
    // short rows: one wave each; rows with more than kSpLong nonzeros (hubs): segments of kSpSeg, one wave each + a reduction
    if (kSpSeg == 0) {
        const int64_t total = sp_rowptr.empty() ? 0 : sp_rowptr.back();
        // (below 1 M nonzeros the GPU has more idle waves than rows: 32-nonzero segments -- ca-HepPh: rows of up to 256 nonzeros on one wave each took
        // 57 us for 115 k nonzeros, 16 batches of gathers one after the other)
        kSpSeg = total < ((int64_t)1 << 20) ? 32 : (total < ((int64_t)4 << 20) ? 128 : (total < ((int64_t)16 << 20) ? 256 : 512));
    }
    int64_t kSpLong = 2 * kSpSeg;
    // Column windows.  The rows of B a launch gathers at any one time should lie close together: on a power-law matrix with millions of columns the nonzeros of the
    // rows in flight are spread over all of B (GBs), every gather is an HBM access (measured with FETCH_SIZE: 327 of 336 GB gathered came from HBM on a part of
    // the 8 M-row R-MAT) and the kernels run at the random-row rate of HBM.  So the columns are cut into windows of win_w columns (16-32 MB of B at N = 256..512
    // in 16 bits), a row of more than 128 nonzeros is cut into segments that also end where its columns cross into another window (once a segment holds sp_minseg
    // nonzeros), and the segments are PROCESSED window by window -- all rows' segments of window 0, then of window 1, ... (workgroups start in the order of the
    // list).  What is in flight then gathers from one window, which the Infinity Cache and the L2s hold: the same part 62.4 -> 35.1 ms, R-MAT 2^20 at 0.1 %
    // (B = 512, bf16) 11.6 -> 8.1 ms per part.  A row's partial rows are still added in segment order (the sum does not depend on the processing order).
    // Width / segment minimum 65536 / 64 -> 32768 / 128 (the defaults): another 3 % on both configs (34.8 -> 33.9 ms, 8.05 -> 7.80).
    // SPARTA_SP_WINDOW_COLS: unset = automatic (on from 262144 columns and 4 M nonzeros on this path), 0 = off, > 0 = the window width in columns;
    // SPARTA_SP_LONG / SPARTA_SP_MINSEG: the row length above which a row is cut / the nonzeros a segment holds before a window boundary ends it.
    const int64_t sp_total = sp_rowptr.empty() ? 0 : sp_rowptr.back();
    // Round 4: the windows are dealt to eight streams, one per XCD (below: every L2 then holds its OWN window instead of a copy of everybody's), and are 8192 columns wide
    // (2 MB of B per L2 at the 256-byte row chunks the large parts run with).  Parts of configs[4] / configs[3] at 1 %, sparse-row kernels, ms: one list of 32768-column
    // windows 25.8 / 31.5 / 10.7 -> eight streams of 8192-column windows 24.5 / 28.9 / 10.3 (4096 .. 32768 columns: within 1 %).  SPARTA_SP_XCD=0: one list, 32768 columns.
    bool sp_xcd = true;
    if (const char* e = std::getenv("SPARTA_SP_XCD")) sp_xcd = atoi(e) != 0;
    int64_t win_w = (cols >= 4 * 65536 && sp_total >= ((int64_t)4 << 20)) ? (sp_xcd ? 8192 : 32768) : 0, sp_minseg = 128;
    std::vector<int32_t> sp_stream_begin;
    if (const char* e = std::getenv("SPARTA_SP_WINDOW_COLS")) win_w = std::max(0, atoi(e));
    if (win_w > 0) kSpLong = 128;
    if (const char* e = std::getenv("SPARTA_SP_LONG")) kSpLong = std::max(8, atoi(e));
    if (const char* e = std::getenv("SPARTA_SP_MINSEG")) sp_minseg = std::max(1, atoi(e));
    // the segments of row t (appended to out, or only counted when out is null)
    auto cut_row = [&](size_t t, std::vector<SpSegRec>* out) -> int32_t {
        const int64_t p0 = sp_rowptr[t], n = sp_rowptr[t + 1] - p0;
        int32_t n_seg = 0;
        if (win_w > 0) {
            const int64_t L = std::max<int64_t>(kSpSeg, 512);
            int64_t o = 0;
            while (o < n) {
                // the segment [o, e): up to L nonzeros, ending early at the first window boundary behind its first sp_minseg nonzeros
                int64_t e = std::min(n, o + L);
                if (e - o > sp_minseg) {
                    const int64_t wend = ((int64_t)sp_col[(size_t)(p0 + o + sp_minseg - 1)] / win_w + 1) * win_w;      // end of the window of the sp_minseg-th nonzero
                    const int32_t* b = sp_col.data() + p0 + o + sp_minseg;
                    const int32_t* bend = sp_col.data() + p0 + e;
                    const int32_t* f = std::lower_bound(b, bend, (int32_t)std::min<int64_t>(wend, INT32_MAX));
                    e = o + sp_minseg + (f - b);
                }
                if (out) out->push_back(SpSegRec{p0 + o, (int32_t)(e - o), 0});
                n_seg++;
                o = e;
            }
            return n_seg;
        }
        // a hub row's segments run in parallel, its partial rows are added one after the other: with segments of L nonzeros the chain is L / 16 gather
        // batches + n / L additions -- shortest near L = sqrt(1.6 n) (a batch ~ 10 additions), never below the size-dependent base, never above 512
        // (ia-wikiquote, 239 k nonzeros with rows of 10^4: 32-nonzero segments everywhere took 271 us, mostly the reduction of its hub rows)
        int64_t L = ((int64_t)std::sqrt(1.6 * (double)n) + 15) / 16 * 16;
        L = std::max(kSpSeg, std::min<int64_t>(512, L));
        for (int64_t o = 0; o < n; o += L) { if (out) out->push_back(SpSegRec{p0 + o, (int32_t)std::min<int64_t>(L, n - o), 0}); n_seg++; }
        return n_seg;
    };
    {
        // rows in order; the segments of the long ones are cut on all threads (chunks of rows, concatenated in row order)
        const int64_t n_rows_sp = (int64_t)sp_crow.size();
        const int nt = sparta::host_threads();
        const int64_t chunk = std::max<int64_t>(1024, (n_rows_sp + 4 * nt - 1) / (4 * std::max(nt, 1)));
        const int64_t n_chunks = (n_rows_sp + chunk - 1) / std::max<int64_t>(chunk, 1);
        std::vector<std::vector<SpSegRec>> segs_of((size_t)n_chunks);
        std::vector<std::vector<SpLongRec>> long_of((size_t)n_chunks);
        std::vector<std::vector<int32_t>> short_of((size_t)n_chunks);
        sparta::parallel_for_dynamic(n_chunks, 1, [&](int64_t lo, int64_t hi, int) {
            for (int64_t c = lo; c < hi; c++)
                for (int64_t t = c * chunk; t < std::min(n_rows_sp, (c + 1) * chunk); t++) {
                    if (sp_rowptr[(size_t)t + 1] - sp_rowptr[(size_t)t] <= kSpLong) { short_of[(size_t)c].push_back((int32_t)t); continue; }
                    SpLongRec lr{(int32_t)t, (int32_t)segs_of[(size_t)c].size(), 0, 0};      // seg_begin: within the chunk for now
                    lr.n_seg = cut_row((size_t)t, &segs_of[(size_t)c]);
                    long_of[(size_t)c].push_back(lr);
                }
        });
        for (int64_t c = 0; c < n_chunks; c++) {
            const int32_t base = (int32_t)sp_segs.size();
            if ((int64_t)sp_segs.size() + (int64_t)segs_of[(size_t)c].size() > INT32_MAX)
                return sparta::fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: too many sparse-row segments for 32-bit indexing");
            sp_list.insert(sp_list.end(), short_of[(size_t)c].begin(), short_of[(size_t)c].end());
            sp_segs.insert(sp_segs.end(), segs_of[(size_t)c].begin(), segs_of[(size_t)c].end());
            for (SpLongRec lr : long_of[(size_t)c]) { lr.seg_begin += base; sp_long.push_back(lr); }
        }
    }
    // a segment's partial row lives at its index in row order (pad); the order of the list is the order the waves take them in: window by window
    // (a stable counting sort on the window of the segment's first column)
    for (size_t i = 0; i < sp_segs.size(); i++) sp_segs[i].pad = (int32_t)i;
    if (win_w > 0 && !sp_segs.empty()) {
        const int64_t n_win = (cols + win_w - 1) / win_w;
        std::vector<int64_t> start((size_t)n_win + 1, 0);
        for (const SpSegRec& g : sp_segs) start[(size_t)(sp_col[(size_t)g.p0] / win_w) + 1]++;
        for (int64_t k = 0; k < n_win; k++) start[(size_t)k + 1] += start[(size_t)k];
        std::vector<SpSegRec> sorted(sp_segs.size());
        for (const SpSegRec& g : sp_segs) sorted[(size_t)start[(size_t)(sp_col[(size_t)g.p0] / win_w)]++] = g;
        sp_segs.swap(sorted);
        // XCD-affine order (k_sparse.hip: sparse_segments_xcd_kernel): the windows are dealt to eight streams -- heaviest first, each to the stream with the fewest
        // nonzeros so far -- and the list becomes stream 0's windows in column order, then stream 1's, ...; workgroup b takes from stream b % 8 = its XCD.
        if (sp_xcd && sp_segs.size() >= 64) {
            std::vector<int64_t> w_nnz((size_t)n_win, 0), w_begin((size_t)n_win + 1, 0);
            for (const SpSegRec& g : sp_segs) { const size_t k = (size_t)(sp_col[(size_t)g.p0] / win_w); w_nnz[k] += g.cnt; w_begin[k + 1]++; }
            for (int64_t k = 0; k < n_win; k++) w_begin[(size_t)k + 1] += w_begin[(size_t)k];
            std::vector<int64_t> order((size_t)n_win);
            for (int64_t k = 0; k < n_win; k++) order[(size_t)k] = k;
            std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return w_nnz[(size_t)a] > w_nnz[(size_t)b]; });
            int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            std::vector<int8_t> stream_of((size_t)n_win, 0);
            for (int64_t k : order) {
                int best = 0;
                for (int x = 1; x < 8; x++) if (load[x] < load[best]) best = x;
                stream_of[(size_t)k] = (int8_t)best;
                load[best] += w_nnz[(size_t)k] + 16 * (w_begin[(size_t)k + 1] - w_begin[(size_t)k]);       // (a segment costs about a batch whatever it holds)
            }
            std::vector<SpSegRec> streamed;
            streamed.reserve(sp_segs.size());
            sp_stream_begin.assign(9, 0);
            for (int x = 0; x < 8; x++) {
                sp_stream_begin[(size_t)x] = (int32_t)streamed.size();
                for (int64_t k = 0; k < n_win; k++)
                    if (stream_of[(size_t)k] == x) streamed.insert(streamed.end(), sp_segs.begin() + w_begin[(size_t)k], sp_segs.begin() + w_begin[(size_t)k + 1]);
            }
            sp_stream_begin[8] = (int32_t)streamed.size();
            sp_segs.swap(streamed);
        }
    }
    n_sp_short = (int64_t)sp_list.size(); n_sp_long = (int64_t)sp_long.size();
    trace.lap("sparse rows (device form)");
    const uint8_t* skip = sparse_flag.empty() ? nullptr : sparse_flag.data();

    // ---- plan: row tiles per class -----------------------------------------------------------------
    std::vector<TileDesc> tiles[4];
    std::vector<BlockRowDesc> brows;
    std::vector<int32_t> jab32((size_t)std::max<int64_t>(nblocks, 1));
    for (int64_t q = 0; q < nblocks; q++) {
        const int64_t jb = jab[jab_lo + q];
        if (jb < 0 || jb >= block_cols) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: jab entry out of range");
        jab32[(size_t)q] = (int32_t)jb;
    }
    int64_t exec_area = 0;
    {
        int64_t jo2 = 0, mo2 = 0;
        const int64_t row0 = row_part[br0];
        for (int64_t ib = br0; ib < br1; ib++) {
            const int64_t h = row_part[ib + 1] - row_part[ib];
            const int64_t nb = nzcount[ib];
            if (h > 0) {
                BlockRowDesc br{mo2, jo2, (int32_t)nb, (int32_t)h, (int32_t)(row_part[ib] - row0), 0};
                brows.push_back(br);
                int64_t r0 = skip && skip[ib - br0] ? h : 0;     // sparse-row block-rows get no tiles
                while (r0 < h) {
                    const int64_t rem = h - r0;
                    int cls;
                    int64_t mt;
                    if (rem > 32) { cls = 2; mt = std::min<int64_t>(rem, 64); }
                    else if (rem > 16) { cls = 1; mt = rem; }
                    else { cls = 0; mt = rem; }
                    const bool tail = (cols % w != 0) && nb > 0 && jab[jab_lo + jo2 + nb - 1] == block_cols - 1;
                    TileDesc t{mo2 + r0, jo2, (int32_t)nb, (int32_t)h, (int32_t)(row_part[ib] - row0 + r0),
                               (int32_t)mt | (tail ? TILE_TAIL : 0)};
                    tiles[cls].push_back(t);
                    const int64_t padded = cls == 0 ? 16 : ((mt + 31) / 32) * 32;
                    exec_area += padded * w * nb;
                    r0 += mt;
                }
            }
            jo2 += nb;
            mo2 += nb * h * w;
        }
    }

    // ---- schedule: per class, 8 contiguous chunks of ~equal cost (one per XCD: neighbouring block-rows gather the
    // same B panels, so they should share an L2), each chunk sorted by descending cost (the hardware hands workgroups
    // to free slots in blockIdx order => longest-processing-time-first per XCD), interleaved so that entry t is XCD
    // t % 8's (t / 8)-th item; short chunks are padded with empty tiles (nb = 0, mt = 0: nothing loaded or stored).
    int64_t n_real[4];
    for (int c = 0; c < 4; c++) n_real[c] = (int64_t)tiles[c].size();
    {
        const char* ord = std::getenv("SPARTA_TILE_ORDER");
        const bool natural = ord && std::strcmp(ord, "natural") == 0;
        for (int c = 0; c < 4; c++) {
            std::vector<TileDesc>& L = tiles[c];
            if (L.empty()) continue;
            const int64_t rows_pad = c == 0 ? 16 : (c == 1 ? 32 : 64);
            auto cost = [&](const TileDesc& t) { return (int64_t)t.nb * rows_pad + rows_pad / 4; };
            int64_t total = 0;
            for (const TileDesc& t : L) total += cost(t);
            std::vector<std::vector<TileDesc>> chunk(8);
            int64_t acc_cost = 0;
            for (const TileDesc& t : L) {
                int x = (int)std::min<int64_t>(7, (acc_cost * 8) / std::max<int64_t>(total, 1));
                chunk[(size_t)x].push_back(t);
                acc_cost += cost(t);
            }
            size_t maxlen = 0;
            for (auto& ch : chunk) {
                if (!natural) std::stable_sort(ch.begin(), ch.end(), [&](const TileDesc& a, const TileDesc& b) { return cost(a) > cost(b); });
                maxlen = std::max(maxlen, ch.size());
            }
            std::vector<TileDesc> arranged(maxlen * 8, TileDesc{0, 0, 0, 1, 0, 0});
            for (size_t x = 0; x < 8; x++)
                for (size_t j = 0; j < chunk[x].size(); j++) arranged[j * 8 + x] = chunk[x][j];
            L.swap(arranged);
        }
    }

    // ---- stream plans (persistent kernels): see build_stream_plans ----
    StreamPlanHost plan;
    {
        StreamPlanIn pin{cols, w, br0, br1, jab_lo, mab_lo, row_part, nzcount, jab, mab, dtype, device, skip};
        trace.lap("tile lists");
        if (int rc = build_stream_plans(pin, plan)) return rc;
        trace.lap("stream plans");
    }
    UnionDevPlan uplan;
    if (ext && !ext->uni.empty()) {
        int n_cus = 256;
        if (!plan_debug) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) n_cus = pr.multiProcessorCount; }
        // (workgroups per CU the kernel is built for: k_union.hip's launch bound and LDS; SPARTA_UNION_WPC: developer A/B)
        const int wpc = [] { const char* e = std::getenv("SPARTA_UNION_WPC"); return e ? std::max(1, atoi(e)) : 3; }();
        if (int rc = build_union_plan(ext->uni, wpc * n_cus, uplan, dtype, n_cus)) return rc;
        trace.lap("column-compacted tiles (plan)");
    }
    std::vector<StepRec>(&steps)[2] = plan.steps;
    std::vector<int32_t>(&wrange)[2] = plan.wrange;
    std::vector<FixRec>& fix = plan.fix;
    std::vector<int32_t>& fix_slots = plan.fix_slots;
    const int n_workers = plan.n_workers, n_split = plan.n_split;
    const int64_t kp = plan.kp;

    // the handle is owned by `hold` until it is handed to the caller: any early return or exception below frees it and every
    // device allocation made so far
    struct VbsDeleter { void operator()(sparta_vbs* q) const { destroy_impl(q); } };
    std::unique_ptr<sparta_vbs, VbsDeleter> hold(new (std::nothrow) sparta_vbs);
    if (plan_debug) return fail(SPARTA_ERR_NO_DEVICE, "sparta_vbs_create: no HIP device visible (this path has no CPU fallback)");
    sparta_vbs* v = hold.get();
    if (!v) return fail(SPARTA_ERR_ALLOC, "sparta_vbs_create: out of host memory");
    v->device = device; v->dtype = dtype;
    v->zero_ranges = plan.zero_ranges;
    v->rows = row_part[br1] - row_part[br0]; v->cols = cols; v->block_rows = br1 - br0; v->w = w;
    v->nblocks = nblocks; v->nztot = nztot; v->exec_area = exec_area;

    DeviceGuard guard(device);
    if (!guard.ok) return fail(SPARTA_ERR_HIP, "sparta_vbs_create: hipSetDevice failed");
#define CREATE_TRY(expr)                                                                                     \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
            return fail(e_ == hipErrorOutOfMemory ? SPARTA_ERR_ALLOC : SPARTA_ERR_HIP, m_);                  \
        }                                                                                                    \
    } while (0)
    // A is padded by 128 floats so that no (masked-off) lane ever forms an address past the allocation
    if (!h16) {
        v->a_bytes = (nztot + 128) * (int64_t)sizeof(float);
        CREATE_TRY(hipMalloc((void**)&v->d_A, (size_t)v->a_bytes));
        CREATE_TRY(hipMemset(v->d_A, 0, (size_t)v->a_bytes));
        if (nztot > 0) CREATE_TRY(hipMemcpy(v->d_A, mab + mab_lo, (size_t)nztot * sizeof(float), hipMemcpyHostToDevice));
    } else {
        // the look-ahead of the pipeline reads up to 5 slices past the last one (never multiplied): pad
        const std::vector<uint16_t>& a0 = plan.a16_steps[0];
        const std::vector<uint16_t>& a1 = plan.a16_steps[1];
        v->a_bytes = ((int64_t)a0.size() + (int64_t)a1.size() + 8 * 64 * 64) * (int64_t)sizeof(uint16_t);
        CREATE_TRY(hipMalloc((void**)&v->d_A, (size_t)v->a_bytes));
        CREATE_TRY(hipMemset(v->d_A, 0, (size_t)v->a_bytes));
        if (!a0.empty()) CREATE_TRY(hipMemcpy(v->d_A, a0.data(), a0.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        if (!a1.empty()) CREATE_TRY(hipMemcpy((uint16_t*)v->d_A + a0.size(), a1.data(), a1.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        v->kp16 = (int)kp;
    }
    CREATE_TRY(hipMalloc((void**)&v->d_jab, jab32.size() * sizeof(int32_t)));
    CREATE_TRY(hipMemcpy(v->d_jab, jab32.data(), jab32.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    for (int c = 0; c < 4; c++) {
        v->n_tiles[c] = (int64_t)tiles[c].size();
        v->n_real_tiles[c] = n_real[c];
        if (tiles[c].empty()) continue;
        CREATE_TRY(hipMalloc((void**)&v->d_tiles[c], tiles[c].size() * sizeof(TileDesc)));
        CREATE_TRY(hipMemcpy(v->d_tiles[c], tiles[c].data(), tiles[c].size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    }
    v->n_brows = (int64_t)brows.size();
    if (!brows.empty()) {
        CREATE_TRY(hipMalloc((void**)&v->d_brows, brows.size() * sizeof(BlockRowDesc)));
        CREATE_TRY(hipMemcpy(v->d_brows, brows.data(), brows.size() * sizeof(BlockRowDesc), hipMemcpyHostToDevice));
    }
    if (!steps[0].empty() || !steps[1].empty() || !fix.empty() || !plan.zero_ranges.empty() || plan.n_hub_steps > 0) {
        v->has_tail = (cols % w) != 0;
        v->n_workers = n_workers; v->n_fix = (int32_t)fix.size(); v->n_split = n_split; v->n_slots = (int32_t)fix_slots.size();
        std::vector<int32_t> big_fix;
        for (size_t q = 0; q < fix.size(); q++) {
            v->max_tile_slots = std::max(v->max_tile_slots, fix[q].n_slots);
            if (fix[q].n_slots > 2 * kFixGroup) big_fix.push_back((int32_t)q);
        }
        v->n_big_fix = (int32_t)big_fix.size();
        if (!big_fix.empty()) {
            CREATE_TRY(hipMalloc((void**)&v->d_big_fix, big_fix.size() * sizeof(int32_t)));
            CREATE_TRY(hipMemcpy(v->d_big_fix, big_fix.data(), big_fix.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        for (int ty = 0; ty < 2; ty++) {
            std::vector<StepRec>& st = steps[ty];
            v->n_steps[ty] = (int64_t)st.size();
            if (st.empty()) continue;
            // the pipeline prefetches up to 5 steps (and up to two 8-record batches) past a range end: pad with harmless copies
            for (int k = 0; k < 32; k++) { StepRec d = st[(size_t)v->n_steps[ty] - 1]; d.mt_flags = (d.mt_flags & ~(STEP_LAST | STEP_SPLIT)) | STEP_FIRST; st.push_back(d); }
            CREATE_TRY(hipMalloc((void**)&v->d_steps[ty], st.size() * sizeof(StepRec)));
            CREATE_TRY(hipMemcpy(v->d_steps[ty], st.data(), st.size() * sizeof(StepRec), hipMemcpyHostToDevice));
            v->h_steps[ty] = st;
            v->n_plan_tiles[ty] = plan.n_plan_tiles[ty];
            v->tiles_row_aligned[ty] = plan.tiles_row_aligned[ty];
            if (ty == 0) v->wide16 = plan.wide16;
            CREATE_TRY(hipMalloc((void**)&v->d_wrange[ty], wrange[ty].size() * sizeof(int32_t)));
            CREATE_TRY(hipMemcpy(v->d_wrange[ty], wrange[ty].data(), wrange[ty].size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        if (!plan.a_frag.empty()) {
            // a SECOND copy of the one-tile part of A (4 KB per step whatever the tile height): if the device cannot hold it the handle is still
            // good -- d_a_frag stays null and the launch code runs the LDS-staged kernel on the legacy image (launch_f32_stream)
            if (hipMalloc((void**)&v->d_a_frag, plan.a_frag.size() * sizeof(float)) == hipSuccess) {
                CREATE_TRY(hipMemcpy(v->d_a_frag, plan.a_frag.data(), plan.a_frag.size() * sizeof(float), hipMemcpyHostToDevice));
                v->a_bytes += (int64_t)(plan.a_frag.size() * sizeof(float));    // the device image holds the one-tile part of A twice (two layouts)
            } else {
                (void)hipGetLastError();
                v->d_a_frag = nullptr;
            }
        }
        if (plan.n_hub_steps > 0) {
            v->hub_g = plan.hub_g; v->hub_workers = plan.hub_workers; v->n_hub_steps = plan.n_hub_steps; v->hub_area = plan.hub_area;
            v->hub_union_area = plan.hub_union_area; v->n_hub_tiles = plan.n_hub_tiles; v->n_hub_groups = plan.n_hub_groups;
            v->hub_chunks = plan.hub_chunks; v->hub_segments = plan.hub_segments;
            CREATE_TRY(hipMalloc((void**)&v->d_hub_steps, plan.hub_steps.size() * sizeof(HubStep)));
            CREATE_TRY(hipMemcpy(v->d_hub_steps, plan.hub_steps.data(), plan.hub_steps.size() * sizeof(HubStep), hipMemcpyHostToDevice));
            CREATE_TRY(hipMalloc((void**)&v->d_hub_tiles, plan.hub_tiles.size() * sizeof(HubTile)));
            CREATE_TRY(hipMemcpy(v->d_hub_tiles, plan.hub_tiles.data(), plan.hub_tiles.size() * sizeof(HubTile), hipMemcpyHostToDevice));
            CREATE_TRY(hipMalloc((void**)&v->d_hub_wrange, plan.hub_wrange.size() * sizeof(int32_t)));
            CREATE_TRY(hipMemcpy(v->d_hub_wrange, plan.hub_wrange.data(), plan.hub_wrange.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            const size_t a_elems = plan.hub_a16_elems;
            CREATE_TRY(hipMalloc((void**)&v->d_hub_A, (a_elems + 64) * sizeof(uint16_t)));
            if (a_elems > 0) CREATE_TRY(hipMemcpy(v->d_hub_A, plan.hub_a16.get(), a_elems * sizeof(uint16_t), hipMemcpyHostToDevice));
            v->a_bytes += (int64_t)(a_elems * sizeof(uint16_t));
            v->exec_area += plan.hub_union_area;
            v->h_hub_steps.swap(plan.hub_steps);
        }
        if (!fix.empty()) {
            CREATE_TRY(hipMalloc((void**)&v->d_fix, fix.size() * sizeof(FixRec)));
            CREATE_TRY(hipMemcpy(v->d_fix, fix.data(), fix.size() * sizeof(FixRec), hipMemcpyHostToDevice));
        }
        CREATE_TRY(hipMalloc((void**)&v->d_fix_slots, std::max<size_t>(fix_slots.size(), 1) * sizeof(int32_t)));
        if (!fix_slots.empty()) CREATE_TRY(hipMemcpy(v->d_fix_slots, fix_slots.data(), fix_slots.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    for (int ty = 0; ty < kUnionTypes; ty++) {
        if (uplan.n_steps[ty] == 0) continue;
        v->u_workers[ty] = uplan.n_workers[ty]; v->u_steps[ty] = uplan.n_steps[ty]; v->u_tiles[ty] = uplan.n_tiles[ty]; v->u_steps_total += uplan.n_steps[ty];
        CREATE_TRY(hipMalloc((void**)&v->d_u_rec[ty], uplan.rec[ty].size() * sizeof(UnionRec)));
        CREATE_TRY(hipMemcpy(v->d_u_rec[ty], uplan.rec[ty].data(), uplan.rec[ty].size() * sizeof(UnionRec), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc((void**)&v->d_u_ids[ty], uplan.ids[ty].size() * sizeof(int32_t)));
        CREATE_TRY(hipMemcpy(v->d_u_ids[ty], uplan.ids[ty].data(), uplan.ids[ty].size() * sizeof(int32_t), hipMemcpyHostToDevice));
        const size_t a_bytes_ty = h16 ? uplan.a16[ty].size() * sizeof(uint16_t) : uplan.a[ty].size() * sizeof(float);
        CREATE_TRY(hipMalloc((void**)&v->d_u_a[ty], a_bytes_ty));
        CREATE_TRY(hipMemcpy(v->d_u_a[ty], h16 ? (const void*)uplan.a16[ty].data() : (const void*)uplan.a[ty].data(), a_bytes_ty, hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc((void**)&v->d_u_wrange[ty], uplan.wrange[ty].size() * sizeof(int32_t)));
        CREATE_TRY(hipMemcpy(v->d_u_wrange[ty], uplan.wrange[ty].data(), uplan.wrange[ty].size() * sizeof(int32_t), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&v->d_u_tail[ty], uplan.tail[ty].size() * sizeof(uint32_t)));
        CREATE_TRY(hipMemcpy(v->d_u_tail[ty], uplan.tail[ty].data(), uplan.tail[ty].size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        v->a_bytes += (int64_t)(a_bytes_ty + uplan.ids[ty].size() * sizeof(int32_t) + uplan.tail[ty].size() * sizeof(uint32_t));
        v->exec_area += uplan.n_steps[ty] * 32 * uplan.type_rows[ty];
        v->u_exec_area += uplan.n_steps[ty] * 32 * uplan.type_rows[ty];
    }
    if (ext) {
        v->u_area = uplan.area; v->u_cols = uplan.cols; v->u_nnz = ext->uni.nnz; v->u_tail_nnz = ext->uni.tail_nnz; v->u_rows = uplan.rows;
        for (int hh = 0; hh < 2; hh++) { v->u_tiles_h[hh] = uplan.tiles_by_height[hh]; v->u_steps_h[hh] = uplan.steps_by_height[hh]; }
    }
    if (!sp_crow.empty()) {
        v->n_sp_rows = (int64_t)sp_crow.size(); v->n_sp_short = n_sp_short; v->n_sp_long = n_sp_long; v->sp_nnz = sp_rowptr.back();
        CREATE_TRY(hipMalloc((void**)&v->d_sp_rowptr, sp_rowptr.size() * sizeof(int64_t)));
        CREATE_TRY(hipMemcpy(v->d_sp_rowptr, sp_rowptr.data(), sp_rowptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc((void**)&v->d_sp_col, (sp_col.size() + 64) * sizeof(int32_t)));      // +64: a batch reads up to SP_BATCH entries at once
        CREATE_TRY(hipMemset(v->d_sp_col, 0, (sp_col.size() + 64) * sizeof(int32_t)));
        CREATE_TRY(hipMalloc((void**)&v->d_sp_val, (sp_val.size() + 64) * sizeof(float)));
        CREATE_TRY(hipMemset(v->d_sp_val, 0, (sp_val.size() + 64) * sizeof(float)));
        if (!sp_col.empty()) {
            CREATE_TRY(hipMemcpy(v->d_sp_col, sp_col.data(), sp_col.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            CREATE_TRY(hipMemcpy(v->d_sp_val, sp_val.data(), sp_val.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        CREATE_TRY(hipMalloc((void**)&v->d_sp_crow, sp_crow.size() * sizeof(int32_t)));
        CREATE_TRY(hipMemcpy(v->d_sp_crow, sp_crow.data(), sp_crow.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        if (!sp_list.empty()) {
            CREATE_TRY(hipMalloc((void**)&v->d_sp_list, sp_list.size() * sizeof(int32_t)));
            CREATE_TRY(hipMemcpy(v->d_sp_list, sp_list.data(), sp_list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        if (!sp_long.empty()) {
            v->n_sp_segs = (int64_t)sp_segs.size();
            CREATE_TRY(hipMalloc(&v->d_sp_segs, sp_segs.size() * sizeof(SpSegRec)));
            CREATE_TRY(hipMemcpy(v->d_sp_segs, sp_segs.data(), sp_segs.size() * sizeof(SpSegRec), hipMemcpyHostToDevice));
            if (!sp_stream_begin.empty()) {
                CREATE_TRY(hipMalloc((void**)&v->d_sp_stream_begin, 9 * sizeof(int32_t)));
                CREATE_TRY(hipMemcpy(v->d_sp_stream_begin, sp_stream_begin.data(), 9 * sizeof(int32_t), hipMemcpyHostToDevice));
                for (int x = 0; x < 8; x++) v->sp_max_stream = std::max<int64_t>(v->sp_max_stream, sp_stream_begin[(size_t)x + 1] - sp_stream_begin[(size_t)x]);
            }
            CREATE_TRY(hipMalloc(&v->d_sp_long, sp_long.size() * sizeof(SpLongRec)));
            CREATE_TRY(hipMemcpy(v->d_sp_long, sp_long.data(), sp_long.size() * sizeof(SpLongRec), hipMemcpyHostToDevice));
        }
    }
    // resident-column image (k_colres.hip): at least half the rows of C are sparse rows (the tile launches, if any, come first on the stream), a column of B fits LDS
    if (!h16 && !sp_crow.empty()) {
        ColresHost H;
        if (build_colres(v->rows, cols, sp_rowptr, sp_col, sp_val, sp_crow, H)) {
            CREATE_TRY(hipMalloc(&v->d_cr_col, H.col.size() * sizeof(uint16_t)));
            CREATE_TRY(hipMemcpy(v->d_cr_col, H.col.data(), H.col.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            if (!H.val.empty()) {
                CREATE_TRY(hipMalloc(&v->d_cr_val, H.val.size() * sizeof(float)));
                CREATE_TRY(hipMemcpy(v->d_cr_val, H.val.data(), H.val.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            CREATE_TRY(hipMalloc((void**)&v->d_cr_meta, H.meta.size() * sizeof(int32_t)));
            CREATE_TRY(hipMemcpy(v->d_cr_meta, H.meta.data(), H.meta.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            CREATE_TRY(hipMalloc(&v->d_cr_mode, H.mode.size()));
            CREATE_TRY(hipMemcpy(v->d_cr_mode, H.mode.data(), H.mode.size(), hipMemcpyHostToDevice));
            CREATE_TRY(hipMalloc(&v->d_cr_parts, H.parts.size() * sizeof(ColresPartDev)));
            CREATE_TRY(hipMemcpy(v->d_cr_parts, H.parts.data(), H.parts.size() * sizeof(ColresPartDev), hipMemcpyHostToDevice));
            CREATE_TRY(hipMalloc((void**)&v->d_cr_dest, H.dest.size() * sizeof(int32_t)));
            CREATE_TRY(hipMemcpy(v->d_cr_dest, H.dest.data(), H.dest.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            if (!H.longs.empty()) {
                CREATE_TRY(hipMalloc(&v->d_cr_longs, H.longs.size() * sizeof(ColresLong)));
                CREATE_TRY(hipMemcpy(v->d_cr_longs, H.longs.data(), H.longs.size() * sizeof(ColresLong), hipMemcpyHostToDevice));
            }
            v->cr_slices = H.max_slices; v->cr_long = (int32_t)H.longs.size(); v->cr_plane = H.max_cells; v->cr_lmax = H.lmax;
            v->cr_parts = (int32_t)H.parts.size(); v->cr_ranges = (int32_t)H.krange.size() - 1;
            for (size_t r = 0; r < H.krange.size(); r++) v->cr_krange[r] = H.krange[r];
            v->cr_entries = H.entries;
            v->a_bytes += (int64_t)(H.col.size() * sizeof(uint16_t) + H.val.size() * sizeof(float));
            v->cr_unit = H.val.empty();
            // the four-part image for products of few column sets (SPARTA_COLRES_SMALL=0: not built)
            const char* es = std::getenv("SPARTA_COLRES_SMALL");
            if (v->cr_parts == 1 && v->cr_ranges == 1 && v->rows >= 2048 && !(es && atoi(es) == 0)) {
                ColresHost S;
                if (build_colres(v->rows, cols, sp_rowptr, sp_col, sp_val, sp_crow, S, 4) && S.val.empty() == H.val.empty()) {
                    auto& cs = v->cr_small;
                    CREATE_TRY(hipMalloc(&cs.col, S.col.size() * sizeof(uint16_t)));
                    CREATE_TRY(hipMemcpy(cs.col, S.col.data(), S.col.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
                    if (!S.val.empty()) {
                        CREATE_TRY(hipMalloc(&cs.val, S.val.size() * sizeof(float)));
                        CREATE_TRY(hipMemcpy(cs.val, S.val.data(), S.val.size() * sizeof(float), hipMemcpyHostToDevice));
                    }
                    CREATE_TRY(hipMalloc((void**)&cs.meta, S.meta.size() * sizeof(int32_t)));
                    CREATE_TRY(hipMemcpy(cs.meta, S.meta.data(), S.meta.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                    CREATE_TRY(hipMalloc((void**)&cs.dest, S.dest.size() * sizeof(int32_t)));
                    CREATE_TRY(hipMemcpy(cs.dest, S.dest.data(), S.dest.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                    if (!S.longs.empty()) {
                        CREATE_TRY(hipMalloc(&cs.longs, S.longs.size() * sizeof(ColresLong)));
                        CREATE_TRY(hipMemcpy(cs.longs, S.longs.data(), S.longs.size() * sizeof(ColresLong), hipMemcpyHostToDevice));
                    }
                    CREATE_TRY(hipMalloc(&cs.parts, S.parts.size() * sizeof(ColresPartDev)));
                    CREATE_TRY(hipMemcpy(cs.parts, S.parts.data(), S.parts.size() * sizeof(ColresPartDev), hipMemcpyHostToDevice));
                    cs.slices = S.max_slices; cs.plane = S.max_cells; cs.n_parts = (int32_t)S.parts.size();
                    v->a_bytes += (int64_t)(S.col.size() * sizeof(uint16_t) + S.val.size() * sizeof(float));
                }
            }
        }
    }
    CREATE_TRY(hipEventCreate(&v->ev0));
    CREATE_TRY(hipEventCreate(&v->ev1));
    CREATE_TRY(hipEventCreate(&v->tev0));
    CREATE_TRY(hipEventCreate(&v->tev1));
#undef CREATE_TRY
    trace.lap("upload");
    *out = hold.release();
    return SPARTA_OK;
}

int sparta_vbs_create_range(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t w, const int64_t* row_part,
                            const int64_t* nzcount, const int64_t* jab, const float* mab, int64_t br0, int64_t br1, int32_t dtype,
                            int32_t device) {
    SPARTA_GUARD_BEGIN
    return create_core(out, rows, cols, block_rows, w, row_part, nzcount, jab, mab, br0, br1, dtype, device, nullptr);
    SPARTA_GUARD_END("sparta_vbs_create")
}

static int create_from_csr_impl(sparta_vbs_t** out, int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                                const int64_t* grouping, int64_t col_block_size, int64_t row_block_size, int32_t force_fixed_size,
                                int32_t dtype, int32_t device, bool keep_order) {
    using sparta::fail;
    if (!out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create_from_csr: out is NULL");
    *out = nullptr;
    if (dtype != SPARTA_F32 && dtype != SPARTA_F16 && dtype != SPARTA_BF16) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create_from_csr: bad dtype");
    sparta_vbs_host h;
    std::memset(&h, 0, sizeof(h));
    int rc = SPARTA_OK;
    try {
        sparta::CsrView a;
        a.rows = rows; a.cols = cols; a.rowptr = rowptr; a.colidx = colidx; a.vals = vals;
        double K = 24.0;
        if (const char* e = std::getenv("SPARTA_SPARSE_K")) K = atof(e);
        const int64_t kdep = (dtype != SPARTA_F32 && col_block_size % 64 == 0) ? 64 : 32;
        sparta::HybridSparse sp;
        sp.esz = dtype == SPARTA_F32 ? 4.0 : 2.0;
        sp.want_union = true;                                              // column-compacted tiles (k_union.hip: fp32 and 16-bit forms)
        sp.union_gran = dtype == SPARTA_F32 ? 16 : 32;
        rc = sparta::vbs_build_hybrid(a, grouping, col_block_size, row_block_size, force_fixed_size != 0, K, kdep, &h, K > 0.0 ? &sp : nullptr, keep_order);
        if (rc == SPARTA_OK)
            rc = create_core(out, h.rows, h.cols, h.block_rows, col_block_size, h.row_part, h.nzcount, h.jab, h.mab, 0, h.block_rows, dtype, device,
                             K > 0.0 ? &sp : nullptr);
        if (rc == SPARTA_OK && K > 0.0) (*out)->ext_sparse = true;
    } catch (const std::bad_alloc&) {
        rc = fail(SPARTA_ERR_ALLOC, "sparta_vbs_create_from_csr: out of host memory");
    } catch (const std::exception& e) {
        rc = fail(SPARTA_ERR_INVALID, std::string("sparta_vbs_create_from_csr: ") + e.what());
    }
    sparta_vbs_host_free(&h);
    return rc;
}

int sparta_vbs_plan_stats(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals, const int64_t* grouping,
                          int64_t col_block_size, int64_t row_block_size, int32_t force_fixed_size, int32_t dtype, int64_t* stats) {
    SPARTA_GUARD_BEGIN
    using sparta::fail;
    if (!stats) return fail(SPARTA_ERR_INVALID, "sparta_vbs_plan_stats: stats is NULL");
    if (dtype != SPARTA_F32 && dtype != SPARTA_F16 && dtype != SPARTA_BF16) return fail(SPARTA_ERR_INVALID, "sparta_vbs_plan_stats: bad dtype");
    sparta::CsrView a;
    a.rows = rows; a.cols = cols; a.rowptr = rowptr; a.colidx = colidx; a.vals = vals;
    double K = 24.0;
    if (const char* e = std::getenv("SPARTA_SPARSE_K")) K = atof(e);
    if (!(K > 0.0)) K = 1e-9;                                             // the sparse-row path switched off: everything is a tile
    const int64_t kdep = (dtype != SPARTA_F32 && col_block_size % 64 == 0) ? 64 : 32;
    sparta::HybridSparse sp;
    sp.esz = dtype == SPARTA_F32 ? 4.0 : 2.0;
    sp.want_union = true;
    sp.union_gran = dtype == SPARTA_F32 ? 16 : 32;
    sparta::HybridStats st;
    sparta_vbs_host h;
    std::memset(&h, 0, sizeof(h));
    const int rc = sparta::vbs_build_hybrid(a, grouping, col_block_size, row_block_size, force_fixed_size != 0, K, kdep, &h, &sp, false, &st);
    sparta_vbs_host_free(&h);
    if (rc != SPARTA_OK) return rc;
    stats[0] = st.tile_blocks; stats[1] = st.tile_area; stats[2] = (int64_t)(st.mfma_steps + 0.5); stats[3] = st.sparse_nnz; stats[4] = st.sparse_rows;
    stats[5] = st.block_rows; stats[6] = st.rows; stats[7] = (int64_t)(st.union_steps + 0.5);     // [7]: 32 x 32 steps of the column-compacted tiles (k_union.hip), not in [2]
    return SPARTA_OK;
    SPARTA_GUARD_END("sparta_vbs_plan_stats")
}

int sparta_vbs_create_from_csr(sparta_vbs_t** out, int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals,
                               const int64_t* grouping, int64_t col_block_size, int64_t row_block_size, int32_t force_fixed_size,
                               int32_t dtype, int32_t device) {
    return create_from_csr_impl(out, rows, cols, rowptr, colidx, vals, grouping, col_block_size, row_block_size, force_fixed_size, dtype, device, false);
}

// A^T of a VBS as a device handle: with it, B * A (dense x VBS) is an ordinary product -- C^T = A^T * B^T, and a column-major
// M x rows B IS a row-major rows x M B^T (same bytes), a column-major M x cols C IS a row-major cols x M C^T.  The reference's own
// "inverted" product (cublas_blockmat_multiplyBA, src/cuda/cuda_utilities.cpp:553-721) is not a B * A; this one is.
int sparta_vbs_create_transposed(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t w, const int64_t* row_part,
                                 const int64_t* nzcount, const int64_t* jab, const float* mab, int32_t dtype, int32_t device) {
    using sparta::fail;
    if (!out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create_transposed: out is NULL");
    *out = nullptr;
    if (rows <= 0 || cols <= 0 || block_rows <= 0 || w <= 0 || !row_part || !nzcount)
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_create_transposed: bad dimensions or NULL index array");
    if (rows > INT32_MAX || cols > INT32_MAX) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create_transposed: more than 2^31 - 1 rows or columns");
    try {
        // CSR of A^T: row j = column j of A, entries (reordered row i, value) ascending in i; exact zeros of the blocks are dropped
        std::vector<int64_t> rp((size_t)cols + 1, 0);
        int64_t jo = 0, mo = 0;
        for (int64_t ib = 0; ib < block_rows; ib++) {
            const int64_t h = row_part[ib + 1] - row_part[ib], nb = nzcount[ib];
            if (h < 0 || nb < 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create_transposed: invalid row_part / nzcount");
            for (int64_t b = 0; b < nb; b++) {
                const int64_t c0 = jab[jo + b] * w;
                if (c0 < 0 || c0 >= cols) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create_transposed: jab entry out of range");
                for (int64_t k = 0; k < w && c0 + k < cols; k++)
                    for (int64_t i = 0; i < h; i++) rp[(size_t)(c0 + k) + 1] += mab[mo + (b * w + k) * h + i] != 0.0f;
            }
            jo += nb; mo += nb * h * w;
        }
        for (int64_t j = 0; j < cols; j++) rp[(size_t)j + 1] += rp[(size_t)j];
        std::vector<int32_t> ci((size_t)rp[(size_t)cols]);
        std::vector<float> va((size_t)rp[(size_t)cols]);
        std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
        jo = 0; mo = 0;
        for (int64_t ib = 0; ib < block_rows; ib++) {
            const int64_t h = row_part[ib + 1] - row_part[ib], nb = nzcount[ib];
            for (int64_t b = 0; b < nb; b++) {
                const int64_t c0 = jab[jo + b] * w;
                for (int64_t k = 0; k < w && c0 + k < cols; k++)
                    for (int64_t i = 0; i < h; i++) {
                        const float a = mab[mo + (b * w + k) * h + i];
                        if (a != 0.0f) { const int64_t q = fill[(size_t)(c0 + k)]++; ci[(size_t)q] = (int32_t)(row_part[ib] + i); va[(size_t)q] = a; }
                    }
            }
            jo += nb; mo += nb * h * w;
        }
        // block-rows of A^T = the column blocks of A (w rows each); its column blocks are 32 reordered rows of A wide
        std::vector<int64_t> grouping((size_t)cols);
        for (int64_t j = 0; j < cols; j++) grouping[(size_t)j] = j / w;
        return create_from_csr_impl(out, cols, rows, rp.data(), ci.data(), va.data(), grouping.data(), 32, 0, 0, dtype, device, true);   // rows stay in place: they are the columns of C
    } catch (const std::bad_alloc&) {
        return fail(SPARTA_ERR_ALLOC, "sparta_vbs_create_transposed: out of host memory");
    } catch (const std::exception& e) {
        return fail(SPARTA_ERR_INVALID, std::string("sparta_vbs_create_transposed: ") + e.what());
    }
}

/* C (+)= B * A with the handle of A^T (sparta_vbs_create_transposed): B is M x rows(A), C is M x cols(A), both column-major. */
int sparta_vbs_spmm_ba(sparta_vbs_t* At, const void* B, int64_t ldb, int32_t M, void* C, int64_t ldc, int32_t accumulate, int32_t ptr_space,
                       void* stream, float* dt_ms) {
    using sparta::fail;
    if (!At) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_ba: NULL handle");
    if (M <= 0 || ldb < M || ldc < M) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_ba: need M > 0, ldb >= M, ldc >= M");
    // a column-major M x rows matrix with leading dimension ld is the row-major rows x M matrix with the same ld
    return sparta_vbs_spmm(At, B, ldb, SPARTA_ROW_MAJOR, M, C, ldc, SPARTA_ROW_MAJOR, accumulate, ptr_space, stream, SPARTA_SPMM_MFMA, dt_ms);
}

int sparta_vbs_create(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t w, const int64_t* row_part,
                      const int64_t* nzcount, const int64_t* jab, const float* mab, int32_t dtype, int32_t device) {
    return sparta_vbs_create_range(out, rows, cols, block_rows, w, row_part, nzcount, jab, mab, 0, block_rows, dtype, device);
}

int sparta_vbs_destroy(sparta_vbs_t* A) {
    destroy_impl(A);
    return SPARTA_OK;
}

int sparta_vbs_info(const sparta_vbs_t* A, int64_t* info) {
    if (!A || !info) return sparta::fail(SPARTA_ERR_INVALID, "sparta_vbs_info: NULL argument");
    std::memset(info, 0, 16 * sizeof(int64_t));
    info[0] = A->rows; info[1] = A->cols; info[2] = A->block_rows; info[3] = A->w; info[4] = A->nblocks; info[5] = A->nztot;
    for (int c = 0; c < 3; c++) info[6 + c] = A->n_real_tiles[c];
    info[9] = A->n_sp_rows;                // rows handled by the sparse-row path
    info[10] = A->a_bytes; info[11] = A->exec_area;
    info[12] = A->n_steps[0] + A->n_steps[1]; info[13] = A->n_workers; info[14] = A->n_split; info[15] = A->last_path;
    return SPARTA_OK;
}

int sparta_vbs_hub_info(const sparta_vbs_t* A, int64_t* info) {
    if (!A || !info) return sparta::fail(SPARTA_ERR_INVALID, "sparta_vbs_hub_info: NULL argument");
    info[0] = A->n_hub_steps; info[1] = A->n_hub_tiles; info[2] = A->n_hub_groups; info[3] = A->n_hub_steps > 0 ? A->hub_g : 0;
    info[4] = A->hub_area; info[5] = A->hub_union_area; info[6] = A->n_hub_steps > 0 ? A->hub_workers : 0; info[7] = A->hub_chunks; info[8] = A->hub_segments; info[9] = 0;
    return SPARTA_OK;
}

int sparta_vbs_union_info(const sparta_vbs_t* A, int64_t* info) {
    if (!A || !info) return sparta::fail(SPARTA_ERR_INVALID, "sparta_vbs_union_info: NULL argument");
    info[0] = A->u_tiles_h[0]; info[1] = A->u_tiles_h[1]; info[2] = A->u_steps_h[0]; info[3] = A->u_steps_h[1];
    info[4] = A->u_area; info[5] = A->u_cols; info[6] = A->u_nnz; info[7] = 0;
    for (int ty = 0; ty < kUnionTypes; ty++) info[7] = std::max<int64_t>(info[7], A->u_workers[ty]);
    info[8] = A->u_rows; info[9] = A->u_tail_nnz;
    info[10] = A->u_exec_area; info[11] = A->u_steps_total > 0 ? (A->dtype == SPARTA_F32 ? 16 : 32) : 0;
    return SPARTA_OK;
}

int sparta_vbs_sparse_info(const sparta_vbs_t* A, int64_t* info) {
    if (!A || !info) return sparta::fail(SPARTA_ERR_INVALID, "sparta_vbs_sparse_info: NULL argument");
    info[0] = A->n_sp_rows; info[1] = A->sp_nnz; info[2] = A->n_sp_short; info[3] = A->n_sp_long;
    return SPARTA_OK;
}

// the resident-column image of a CSR matrix, built and walked on the HOST exactly as k_colres.hip walks it (slots in slice order, the sums of a slot in entry order,
// extra cells added to their row in chunk order) for ONE column x of B: what the CPU suite checks the builder with (no GPU involved; not a product path)
int sparta_colres_host_check(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals, const int64_t* crow, const float* x, float* y,
                             int64_t* info) {
    using sparta::fail;
    SPARTA_GUARD_BEGIN
    if (rows <= 0 || cols <= 0 || !rowptr || !colidx || !vals || !x || !y || !info) return fail(SPARTA_ERR_INVALID, "sparta_colres_host_check: bad argument");
    // crow[i] = the row of C of CSR row i (NULL: i); + 2^31: the row ADDS to y (a mixed block-row's sparse part); -1: CSR row i is not a sparse row (its row of C belongs to tiles: y untouched)
    std::vector<int64_t> rp{0};
    std::vector<int32_t> ci, cr;
    std::vector<float> va;
    for (int64_t i = 0; i < rows; i++) {
        if (crow && crow[i] == -1) continue;
        ci.insert(ci.end(), colidx + rowptr[i], colidx + rowptr[i + 1]);
        va.insert(va.end(), vals + rowptr[i], vals + rowptr[i + 1]);
        rp.push_back((int64_t)ci.size());
        cr.push_back(crow ? (int32_t)(uint32_t)(crow[i] & 0xffffffffll) : (int32_t)i);
    }
    ColresHost H;
    for (int k = 0; k < 12; k++) info[k] = 0;
    const int force_parts = [] { const char* e = std::getenv("SPARTA_COLRES_FORCE_PARTS"); return e ? atoi(e) : 0; }();      // (4: the image a handle keeps for products of few column sets)
    if (!build_colres(rows, cols, rp, ci, va, cr, H, force_parts)) return SPARTA_OK;    // info[0] = 0: this matrix gets no image
    const int n_ranges = (int)H.krange.size() - 1;
    for (const ColresPartDev& pd : H.parts) {
        const int32_t* wslice = H.meta.data() + pd.meta;
        const int32_t* woff = wslice + 17;
        const int32_t* bnd = woff + (size_t)n_ranges * 17;
        std::vector<float> acc((size_t)pd.n_slices * 64, 0.0f), cell((size_t)pd.plane, 0.0f);
        for (int r = 0; r < n_ranges; r++) {                                          // the sums of a slot carry over from range to range, as the kernel's registers do
            const int64_t k0 = H.krange[(size_t)r], klen = H.krange[(size_t)r + 1] - k0;
            for (int w = 0; w < kColresWaves; w++) {
                for (int32_t i = wslice[w]; i < wslice[w + 1]; i++) {
                    const int32_t t0 = i == wslice[w] ? 0 : bnd[(size_t)r * pd.n_slices + i - 1], t1 = bnd[(size_t)r * pd.n_slices + i];      // batches of 4 steps
                    if (t1 <= t0) return fail(SPARTA_ERR_INVALID, "sparta_colres_host_check: an empty slice in the image");
                    for (int l = 0; l < 64; l++) {
                        float a = acc[(size_t)i * 64 + (size_t)l];
                        for (int32_t t = t0; t < t1; t++) {
                            for (int u = 0; u < 4; u++) {
                                const size_t at = ((size_t)(woff[(size_t)r * 17 + w] + t) * 64 + (size_t)l) * 4 + (size_t)u;
                                const int64_t c = H.col[at];
                                if (c > klen) return fail(SPARTA_ERR_INVALID, "sparta_colres_host_check: column out of range in the image");
                                const float b = c == klen ? 0.0f : x[k0 + c];           // the zero cell behind the range's last row of B
                                a = H.val.empty() ? a + b : std::fma(H.val[at], b, a);
                            }
                        }
                        acc[(size_t)i * 64 + (size_t)l] = a;
                    }
                }
                if (wslice[w + 1] > wslice[w] && (int64_t)bnd[(size_t)r * pd.n_slices + wslice[w + 1] - 1] != (int64_t)woff[(size_t)r * 17 + w + 1] - woff[(size_t)r * 17 + w])
                    return fail(SPARTA_ERR_INVALID, "sparta_colres_host_check: a wave's stream and its slice boundaries disagree");
            }
        }
        for (int32_t q = 0; q < pd.n_slices * 64; q++) {
            const int32_t d = H.dest[(size_t)pd.dest + (size_t)q];
            if (d >= pd.plane) return fail(SPARTA_ERR_INVALID, "sparta_colres_host_check: cell out of range in the image");
            if (d >= 0) cell[(size_t)d] = acc[(size_t)q];
        }
        for (int32_t q = 0; q < pd.n_long; q++) {
            const ColresLong& lr = H.longs[(size_t)pd.longs + (size_t)q];
            float sum = cell[(size_t)lr.row];
            for (int32_t i = 0; i < lr.n; i++) sum += cell[(size_t)lr.first + (size_t)i];
            cell[(size_t)lr.row] = sum;
        }
        for (int32_t i = 0; i < pd.rows; i++) {
            const uint8_t m = H.mode[(size_t)(pd.r0 + i)];
            if (m == 1) y[pd.r0 + i] = cell[(size_t)i]; else if (m == 2) y[pd.r0 + i] += cell[(size_t)i];
        }
    }
    info[0] = H.max_slices; info[1] = H.entries; info[2] = (int64_t)H.longs.size(); info[3] = H.max_cells; info[4] = H.lmax; info[6] = rp.back(); info[7] = H.val.empty() ? 1 : 0;
    info[8] = (int64_t)H.parts.size(); info[9] = n_ranges;
    return SPARTA_OK;
    SPARTA_GUARD_END("sparta_colres_host_check")
}

// the hybrid image of a CSR matrix under `grouping`, built exactly as sparta_vbs_create_from_csr builds it for an fp32 handle -- w-wide tiles, column-compacted tiles,
// sparse rows -- and multiplied with ONE column x on the HOST: the w-wide tiles from the reference-layout image, the column-compacted tiles by walking their DEVICE
// form (per-worker step records, list entries, fragment-order slices: vbs_union.cpp), the sparse rows entry by entry.  What the CPU suite checks the mode-3 builder
// and the plan with (no GPU involved; not a product path).  y: [padded rows], reordered order, double.
int sparta_union_host_check(int64_t rows, int64_t cols, const int64_t* rowptr, const int32_t* colidx, const float* vals, const int64_t* grouping, int64_t col_block_size,
                            int64_t row_block_size, int32_t force_fixed_size, int32_t max_workers, const float* x, double* y, int64_t* info) {
    using sparta::fail;
    SPARTA_GUARD_BEGIN
    if (!rowptr || !colidx || !grouping || !x || !y || !info || max_workers <= 0) return fail(SPARTA_ERR_INVALID, "sparta_union_host_check: bad argument");
    sparta::CsrView a;
    a.rows = rows; a.cols = cols; a.rowptr = rowptr; a.colidx = colidx; a.vals = vals;
    double K = 24.0;
    if (const char* e = std::getenv("SPARTA_SPARSE_K")) K = atof(e);
    if (!(K > 0.0)) return fail(SPARTA_ERR_INVALID, "sparta_union_host_check: the hybrid builder is switched off (SPARTA_SPARSE_K=0)");
    sparta::HybridSparse sp;
    sp.esz = 4.0; sp.want_union = true; sp.union_gran = 16;
    sparta_vbs_host h;
    std::memset(&h, 0, sizeof(h));
    struct Free { sparta_vbs_host* p; ~Free() { sparta_vbs_host_free(p); } } guard{&h};
    if (int rc = sparta::vbs_build_hybrid(a, grouping, col_block_size, row_block_size, force_fixed_size != 0, K, 32, &h, &sp)) return rc;
    for (int64_t r = 0; r < h.rows; r++) y[r] = 0.0;
    int64_t jo = 0, mo = 0;
    for (int64_t ib = 0; ib < h.block_rows; ib++) {                       // the w-wide tiles
        const int64_t r0 = h.row_part[ib], hh = h.row_part[ib + 1] - r0;
        for (int64_t b = 0; b < h.nzcount[ib]; b++, jo++, mo += hh * col_block_size)
            for (int64_t k = 0; k < col_block_size && h.jab[jo] * col_block_size + k < cols; k++)
                for (int64_t i = 0; i < hh; i++) y[r0 + i] += (double)h.mab[mo + k * hh + i] * (double)x[h.jab[jo] * col_block_size + k];
    }
    UnionDevPlan P;
    if (!sp.uni.empty()) {
        if (int rc = build_union_plan(sp.uni, max_workers, P)) return rc;
        union_plan_host_apply(P, x, y);
    }
    for (size_t t = 0; t < sp.crow.size(); t++)                           // the sparse rows (they own their row or add to it: y started at zero either way)
        for (int64_t k = sp.rowptr[t]; k < sp.rowptr[t + 1]; k++) y[sp.crow[t]] += (double)sp.val[(size_t)k] * (double)x[sp.col[(size_t)k]];
    info[0] = (int64_t)sp.uni.tiles[0].size(); info[1] = (int64_t)sp.uni.tiles[1].size(); info[2] = P.steps_by_height[0]; info[3] = P.steps_by_height[1];
    info[4] = P.area; info[5] = P.cols; info[6] = sp.uni.nnz; info[7] = sp.rowptr.empty() ? 0 : sp.rowptr.back();
    info[8] = h.nztot; info[9] = h.rows; info[10] = std::max(P.n_workers[0], P.n_workers[1]); info[11] = std::max(P.n_workers[2], P.n_workers[3]); info[12] = sp.uni.tail_nnz; info[13] = P.rows;
    return SPARTA_OK;
    SPARTA_GUARD_END("sparta_union_host_check")
}

int sparta_vbs_colres_info(const sparta_vbs_t* A, int64_t* info) {
    if (!A || !info) return sparta::fail(SPARTA_ERR_INVALID, "sparta_vbs_colres_info: NULL argument");
    info[0] = A->cr_slices; info[1] = A->cr_entries; info[2] = A->cr_long; info[3] = A->cr_plane; info[4] = A->cr_lmax; info[5] = A->last_colres_nc;
    info[6] = A->cr_slices > 0 ? A->sp_nnz : 0; info[7] = A->cr_slices > 0 && A->cr_unit ? 1 : 0; info[8] = A->cr_parts; info[9] = A->cr_ranges; info[10] = A->cr_small.n_parts; info[11] = A->last_colres_small ? 1 : 0;
    return SPARTA_OK;
}

#ifdef SPARTA_TIMELINE
// developer build only: the raw timeline words (4 waves x 64 steps x 8)
int sparta_debug_timeline(sparta_vbs_t* A, long long* out) {
    if (!A || !out || !A->d_clk) return -1;
    DeviceGuard guard(A->device);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpy(out, A->d_clk + 16, (4 * 64 * 8 + 2048) * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

int sparta_vbs_clock_mhz(sparta_vbs_t* A, double* mhz_out) {
    using sparta::fail;
    if (!A || !mhz_out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_clock_mhz: NULL argument");
    DeviceGuard guard(A->device);
    for (int c = 0; c < 4; c++) mhz_out[c] = 0.0;
    if (!A->class_timing || !A->d_clk) return SPARTA_OK;
    long long h[16];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h, A->d_clk, sizeof(h), hipMemcpyDeviceToHost));
    for (int c = 0; c < 4; c++) {
        if (!A->class_ran[c]) continue;
        const long long dc = h[4 * c + 2] - h[4 * c], dr = h[4 * c + 3] - h[4 * c + 1];
        if (dr > 0 && dc > 0) mhz_out[c] = (double)dc / (double)dr * 100.0;      // s_memrealtime ticks at 100 MHz
    }
    return SPARTA_OK;
}

}  // extern "C"

namespace {

// do the <= 32-row tiles of this handle go through the C ring (CRing)?  Tiles that do not start on multiples of 32 rows AND are short (the plan kept
// contiguous worker ranges for exactly these: vbs_plan.cpp, ring_plan)
bool ring_tiles(const sparta_vbs_t* A) { return !A->tiles_row_aligned[0] && A->n_steps[0] < 6 * A->n_plan_tiles[0]; }

// cache policy of a stream launch's C stores (vbs_kernel_common.hpp): non-temporal for long tiles; SPARTA_C_NT=0|1 forces one
int32_t c_store_nt(const sparta_vbs_t* A, int ty, const float* C, int64_t ldc, bool c_row_major, bool ring = false) {
    if (const char* e = std::getenv("SPARTA_C_NT")) return atoi(e) != 0;
    if (A->n_plan_tiles[ty] <= 0) return 0;
    if (A->n_steps[ty] >= 6 * A->n_plan_tiles[ty]) return 1;                    // long tiles
    // short tiles: only when a store instruction writes whole, aligned 128-byte lines (32-row tiles starting at multiples of 32 rows of a
    // column-major C whose columns are 128-byte aligned) -- banded 200k in fixed 32-row tiles: 63 us default, 51 non-temporal; the same
    // matrix in tiles of 28 rows on average (misaligned pieces of lines): 67-71 default, 79 non-temporal
    // (`ring`: the no-barrier fp32 kernel parks tiles of arbitrary height in LDS and stores aligned blocks of 32 rows: banded 200k 51 us default, 45 non-temporal)
    return !c_row_major && (A->tiles_row_aligned[ty] || ring) && ldc % 32 == 0 && ((uintptr_t)C % 128) == 0;
}

// long runs of rows without blocks, accumulate = 0: streamed zero fill (vbs_zero_rows_kernel), one launch per run
void launch_zero_ranges(sparta_vbs_t* A, float* C, int64_t ldc, bool c_row_major, int n_cols, hipStream_t st) {
    for (const auto& zr : A->zero_ranges) {
        const int64_t n_lines = c_row_major ? zr.second : (int64_t)n_cols, line_len = c_row_major ? (int64_t)n_cols : zr.second;
        const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (line_len / 4 + kThreads - 1) / kThreads));
        const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_lines, 16384));
        launch_zero_rows(dim3(gx, gy), st, C, ldc, (int)c_row_major, zr.first, zr.second, n_cols);
    }
}

// step lists for a gathered B (slab index + row inside the slab), rebuilt when the slab height changes
int ensure_gathered_steps(sparta_vbs_t* A, int64_t shard_rows, hipStream_t st) {
    if (A->g_shard_rows == shard_rows) return SPARTA_OK;
    if (g_capturing) return capture_refusal("build the step lists of this slab height");
    for (int ty = 0; ty < 2; ty++) {
        if (A->h_steps[ty].empty()) continue;
        std::vector<StepRec> g = A->h_steps[ty];
        for (StepRec& r : g) { r.pad = (int32_t)(r.b_row / shard_rows); r.b_row = (int32_t)(r.b_row % shard_rows); }
        if (!A->d_steps_g[ty]) HIP_TRY(hipMalloc((void**)&A->d_steps_g[ty], g.size() * sizeof(StepRec)));
        HIP_TRY(hipMemcpyAsync(A->d_steps_g[ty], g.data(), g.size() * sizeof(StepRec), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));                       // g goes out of scope
    }
    if (!A->h_hub_steps.empty()) {
        std::vector<HubStep> g = A->h_hub_steps;
        for (HubStep& r : g) if (!(r.flags & STEP_TAIL)) { r.shard = (int32_t)(r.b_row / shard_rows); r.b_row = (int32_t)(r.b_row % shard_rows); }
        if (!A->d_hub_steps_g) HIP_TRY(hipMalloc((void**)&A->d_hub_steps_g, g.size() * sizeof(HubStep)));
        HIP_TRY(hipMemcpyAsync(A->d_hub_steps_g, g.data(), g.size() * sizeof(HubStep), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    A->g_shard_rows = shard_rows;
    return SPARTA_OK;
}


// fp32 handles whose <= 32-row tiles have a fragment image (k_f32_direct.hip) hold A twice at create time.  Once the no-barrier kernel has won the plan-time autotune on a
// handle whose blocks are ALL in that image (no 33..64-row tiles, no sparse rows), the reference-layout copy is dropped (drop_legacy_image: a_bytes = one image); a later call that reads it -- SPARTA_SPMM_EXACT, a
// row-major or gathered B, the per-class kernels -- rebuilds it on the device from the fragment image (an exact copy) and then keeps it.  SPARTA_F32_KEEP_LEGACY=1: never drop.
int ensure_legacy_image(sparta_vbs_t* A, hipStream_t st) {
    if (A->d_A || A->dtype != SPARTA_F32) return SPARTA_OK;
    if (g_capturing) return capture_refusal("rebuild the reference-layout image of A");
    const size_t bytes = (size_t)(A->nztot + 128) * sizeof(float);
    HIP_TRY(hipMalloc((void**)&A->d_A, bytes));
    HIP_TRY(hipMemsetAsync(A->d_A, 0, bytes, st));
    if (A->n_steps[0] > 0) launch_f32_legacy_from_frag(st, A->d_steps[0], A->n_steps[0], A->d_a_frag, A->d_A);
    HIP_TRY(hipGetLastError());
    A->a_bytes += (int64_t)bytes;
    return SPARTA_OK;
}
void drop_legacy_image(sparta_vbs_t* A) {
    static const bool keep = [] { const char* e = std::getenv("SPARTA_F32_KEEP_LEGACY"); return e && atoi(e) != 0; }();
    if (keep || g_capturing || A->legacy_dropped || !A->d_A || !A->d_a_frag || A->dtype != SPARTA_F32 || A->n_steps[1] != 0 || A->n_sp_rows != 0 || A->class_timing) return;
    (void)hipFree(A->d_A);                               // (synchronises: the autotune's launches that read it are done)
    A->d_A = nullptr;
    A->a_bytes -= (int64_t)((A->nztot + 128) * sizeof(float));
    A->legacy_dropped = true;
}

// The ROW-major B of the product in flight (A->brm_ready / brm_ld; cleared when the product returns): B itself when the call's B is row-major, the caller's prepared
// copy (sparta_vbs_prepare_b), or a transposed copy made now -- ONCE per product, whoever asks first (the column-compacted tiles, then the sparse rows).  `aligned16`:
// rows must start on 16-byte boundaries (k_union.hip loads them 16 bytes per lane); the copies made here always do (row stride rounded up to 16 bytes).
int ensure_brm(sparta_vbs_t* A, const void* dB, int64_t ldb, bool b_row_major, int64_t shard_rows, int64_t shard_stride, int bk, int32_t n_cols, hipStream_t st,
               bool aligned16) {
    if (A->brm_ready) return SPARTA_OK;
    const size_t esz = bk == 0 ? 4 : 2;
    const int64_t per16 = 16 / (int64_t)esz;
    if (b_row_major && shard_rows == 0 && (!aligned16 || (ldb % per16 == 0 && ((uintptr_t)dB % 16) == 0))) { A->brm_ready = dB; A->brm_ld = ldb; return SPARTA_OK; }
    if (A->prepared_brm && !b_row_major) { A->brm_ready = A->prepared_brm; A->brm_ld = A->prepared_ld; return SPARTA_OK; }
    if (g_capturing && A->d_Brm_bytes < (size_t)A->cols * (size_t)((n_cols + per16 - 1) / per16 * per16) * esz) return capture_refusal("allocate the row-major copy of B");
    const int64_t ld = (n_cols + per16 - 1) / per16 * per16;
    if (int rc = ensure_scratch(&A->d_Brm, &A->d_Brm_bytes, (size_t)A->cols * (size_t)ld * esz)) return rc;
    if (b_row_major) {                                   // (a row-major B whose rows are not 16-byte aligned: an aligned copy)
        HIP_TRY(hipMemcpy2DAsync(A->d_Brm, (size_t)ld * esz, dB, (size_t)ldb * esz, (size_t)n_cols * esz, (size_t)A->cols, hipMemcpyDeviceToDevice, st));
    } else {
        const int64_t n_wg = ((A->cols + 63) / 64) * (int64_t)((n_cols + 63) / 64);
        if (n_wg > INT32_MAX) return sparta::fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: B too large for the transpose grid");
        launch_b_to_row_major(bk != 0, (unsigned)n_wg, st, dB, ldb, shard_rows, shard_stride, A->cols, (int)n_cols, A->d_Brm, ld);
    }
    A->brm_ready = A->d_Brm; A->brm_ld = ld;
    return SPARTA_OK;
}

// the column-compacted tiles (k_union.hip): C[tile rows, :] (+)= Atile . B[list, :] on the matrix cores, B row-major.  Stores (or adds to) every row of its tiles;
// the sparse rows of those block-rows (their thinly used columns) add behind it on the same stream.
int launch_union_tiles(sparta_vbs_t* A, const void* dB, int64_t ldb, bool b_row_major, int64_t shard_rows, int64_t shard_stride, int32_t n_cols, float* dC,
                       int64_t ldc, bool c_row_major, bool accumulate, hipStream_t st) {
    if (A->u_steps_total == 0) return SPARTA_OK;
    // 32-bit byte offsets inside a tile's 64 rows x 32 columns of C
    if ((c_row_major ? 64 * ldc + 128 : 32 * ldc + 128) * 4 >= ((int64_t)1 << 31) - 65536)
        return sparta::fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: leading dimension of C too large for the column-compacted tile kernel (32-bit offsets inside a tile)");
    const int bk = A->dtype == SPARTA_F32 ? 0 : (A->dtype == SPARTA_BF16 ? 2 : 1);
    if (int rc = ensure_brm(A, dB, ldb, b_row_major, shard_rows, shard_stride, bk, n_cols, st, true)) return rc;
    UnionParams up;
    up.B = (const float*)A->brm_ready; up.ldb = A->brm_ld; up.C = dC; up.ldc = ldc;
    up.n_cols = n_cols; up.accumulate = accumulate ? 1 : 0; up.c_row_major = c_row_major ? 1 : 0;
    up.pad = [] { const char* e = std::getenv("SPARTA_UNION_PROBE"); return e ? atoi(e) : 0; }();      // developer probes (timing only, wrong products; read per call)
    for (int ty = 0; ty < kUnionTypes; ty++) {
        UnionSide& sd = up.side[ty];
        sd.rec = A->d_u_rec[ty]; sd.ids = A->d_u_ids[ty]; sd.A = A->d_u_a[ty]; sd.worker_range = A->d_u_wrange[ty]; sd.tail = (const uint2*)A->d_u_tail[ty];
        sd.n_workers = A->u_steps[ty] > 0 ? A->u_workers[ty] : 0;
        // C is written once and never read back: non-temporal stores where tiles are long (as the stream kernels: vbs_kernel_common.hpp); SPARTA_C_NT=0|1 forces one
        const char* e = std::getenv("SPARTA_C_NT");
        sd.c_nt = e ? (atoi(e) != 0) : (A->u_steps[ty] >= 6 * A->u_tiles[ty] ? 1 : 0);
    }
    if (bk == 0) launch_union_f32((unsigned)((n_cols + kTN - 1) / kTN), st, up);
    else launch_union_h16(bk == 2, (unsigned)((n_cols + kTN - 1) / kTN), st, up);
    HIP_TRY(hipGetLastError());
    return SPARTA_OK;
}

int launch_sparse_rows(sparta_vbs_t* A, const void* dB, int64_t ldb, bool b_row_major, int64_t shard_rows, int64_t shard_stride, int bk,
                       int32_t n_cols, float* dC, int64_t ldc, bool c_row_major, bool accumulate, hipStream_t st) {
    const size_t esz = bk == 0 ? 4 : 2;
    A->last_colres_nc = 0;
    // Small A, the reference's own layouts (column-major B and C): the resident-column product -- NC columns of B in LDS, A streamed past them from L2, one launch
    // (k_colres.hip).  No transpose of B, no partial rows, B and C cross HBM once.
    if (bk == 0 && A->cr_slices > 0 && !b_row_major && !c_row_major && shard_rows == 0) {
        const int nc = colres_columns(A, n_cols);
        if (nc > 0) {
            int nc_used = nc;
            bool used_small = false;
            ColresParams cp;
            cp.col4 = (const uint2*)A->d_cr_col; cp.val4 = (const float4*)A->d_cr_val; cp.parts = (const ColresPartDev*)A->d_cr_parts; cp.meta = A->d_cr_meta;
            cp.dest = A->d_cr_dest; cp.longs = (const ColresLong*)A->d_cr_longs; cp.mode = (const uint8_t*)A->d_cr_mode;
            cp.B = (const float*)dB; cp.ldb = ldb; cp.C = dC; cp.ldc = ldc;
            for (int r = 0; r <= kColresMaxRanges; r++) cp.krange[r] = A->cr_krange[r];
            cp.n_parts = A->cr_parts; cp.n_ranges = A->cr_ranges; cp.N = n_cols; cp.accumulate = accumulate ? 1 : 0;
            cp.vec_out = ldc % 4 == 0 && ((uintptr_t)dC % 16) == 0 ? 1 : 0;
            cp.vec_in = ldb % 4 == 0 && ((uintptr_t)dB % 16) == 0 ? 1 : 0;
            cp.probe = [] { const char* e = std::getenv("SPARTA_COLRES_PROBE"); return e ? atoi(e) : 0; }();        // (read per call: developer A/B)
            size_t lds_bytes = (size_t)A->cr_plane * (size_t)nc * sizeof(float);
            // few column sets: the four-part image, the fewest columns per workgroup that keep the product one round of workgroups (read per call: SPARTA_COLRES_SMALL=0 keeps the whole image)
            if (A->cr_small.slices > 0) {
                const char* es = std::getenv("SPARTA_COLRES_SMALL");
                const char* en = std::getenv("SPARTA_COLRES_NC");
                const int fit = (int)std::min<int64_t>(std::min<int64_t>(4, kColresCells / A->cr_plane), n_cols);
                int ncs = 0;
                for (int c = 1; c <= fit && ncs == 0; c++) if ((int64_t)((n_cols + c - 1) / c) * A->cr_small.n_parts <= 256 && A->cr_small.slices <= colres_max_slices(c)) ncs = c;
                if (ncs > 0 && !en && !(es && atoi(es) == 0)) {
                    nc_used = ncs; used_small = true;
                    cp.col4 = (const uint2*)A->cr_small.col; cp.val4 = (const float4*)A->cr_small.val; cp.parts = (const ColresPartDev*)A->cr_small.parts; cp.meta = A->cr_small.meta;
                    cp.dest = A->cr_small.dest; cp.longs = (const ColresLong*)A->cr_small.longs;
                    cp.n_parts = A->cr_small.n_parts;
                    lds_bytes = (size_t)A->cr_plane * (size_t)ncs * sizeof(float);
                }
            }
            {
                static const int n_cus = [] { hipDeviceProp_t pr; return hipGetDeviceProperties(&pr, 0) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }();
                cp.n_cus = n_cus;
                // start the CUs in three groups, a quarter of a workgroup's time apart (its stream of A: LDS-bound, ~7 / 7 / 14 / 13 cycles per step of 64 slots with 1..4 columns;
                // its columns of B and C at HBM speed).  Measured at N = 8192 (profiles/r4/lab_colres_stagger2.txt): bcsstk18 0.262 -> 0.234 ms, wiki-Vote 0.217 -> 0.189,
                // ca-HepPh 0.326 -> 0.294; two groups or four, or offsets of half a workgroup's time: less or nothing.  SPARTA_COLRES_STAGGER_US / SPARTA_COLRES_GROUPS:
                // developer A/B, read per call; 0 = everybody at once
                static const double step_cycles[5] = {0.0, 7.0, 7.0, 14.0, 13.0};
                const double t_wg = (double)A->cr_entries / 64.0 * step_cycles[nc] / 2.4e9 + (double)(A->rows + A->cols) * nc * 4.0 / (6.0e12 / n_cus) + 3.0e-6;
                const char* e = std::getenv("SPARTA_COLRES_STAGGER_US");
                const char* eg = std::getenv("SPARTA_COLRES_GROUPS");
                cp.share = eg ? std::max(1, atoi(eg)) : 3;
                const double stagger = e ? atof(e) * 1e-6 : t_wg / 4.0;
                cp.stagger_ticks = cp.share > 1 && (int64_t)((n_cols + nc_used - 1) / nc_used) * cp.n_parts > 2 * (int64_t)n_cus ? (int32_t)std::min(stagger * 1e8, 1.0e5) : 0;
            }
            if (int hrc = launch_colres(nc_used, cp, lds_bytes, st)) return sparta::fail(SPARTA_ERR_HIP, "sparta_vbs_spmm: the resident-column kernel could not be launched (hipError_t " + std::to_string(hrc) + ")");
            HIP_TRY(hipGetLastError());
            A->last_colres_nc = nc_used; A->last_colres_small = used_small;
            return SPARTA_OK;
        }
    }
    SparseParams q;
    q.rowptr = A->d_sp_rowptr; q.col = A->d_sp_col; q.val = A->d_sp_val; q.crow = A->d_sp_crow;
    q.list = nullptr; q.n_list = 0;
    q.N = n_cols; q.accumulate = accumulate;
    q.b_col_stride = 0; q.shard_rows = 0; q.shard_stride = 0;
    // A column-major B read in place costs one 64-byte line per ELEMENT (16 x the bytes of a row-major row); transposing costs
    // 2 x |B| once.  In place wins while  nnz * 16 < 2 * cols.
    // SPARTA_SP_INPLACE = 0 | 1 forces the choice (developer A/B)
    static const int inplace_env = [] { const char* e = std::getenv("SPARTA_SP_INPLACE"); return e ? atoi(e) : -1; }();
    // (a row-major copy the column-compacted tiles of this product already made is used, whatever the count)
    const bool in_place = !(b_row_major && shard_rows == 0) && !A->brm_ready && (inplace_env >= 0 ? inplace_env != 0 : A->sp_nnz * 8 < A->cols);
    if (b_row_major && shard_rows == 0) { q.B = dB; q.ldb = ldb; }
    else if (in_place) { q.B = dB; q.ldb = 0; q.b_col_stride = ldb; q.shard_rows = shard_rows; q.shard_stride = shard_stride; }
    else {                                               // sparta_vbs_prepare_b: transposed once; else now, once per product (ensure_brm)
        if (int rc = ensure_brm(A, dB, ldb, b_row_major, shard_rows, shard_stride, bk, n_cols, st, false)) return rc;
        q.B = A->brm_ready; q.ldb = A->brm_ld;
    }
    // round 4: (column, value) pairs through scalar loads and the row of B as an SGPR base (k_sparse.hip: sparse_row_partial_s) -- rows of B shorter than 4 GB (the
    // row offset is a 32 x 32 -> 64 bit scalar multiply); SPARTA_SP_SCALAR=0: the round-3 gather (developer A/B; read per call)
    {
        const char* e = std::getenv("SPARTA_SP_SCALAR");
        q.scalar_gather = (e ? atoi(e) != 0 : true) && q.b_col_stride == 0 && q.ldb * (int64_t)esz < ((int64_t)1 << 32) ? 1 : 0;
    }
    // the kernels write C themselves: rows of a row-major C, or 16-row pieces of the columns of the reference's column-major C
    q.out = dC; q.ldo = ldc; q.out_is_c = c_row_major ? 1 : 2;
    if (A->n_sp_long > 0)
        if (int rc = ensure_scratch(&A->d_sp_part, &A->d_sp_part_bytes, (size_t)A->n_sp_segs * (size_t)n_cols * sizeof(float))) return rc;
    // widest vector the shapes allow: every row start VEC-element aligned, N a multiple of 64 * VEC (no ragged chunk)
    auto aligned = [&](int v) {
        const bool out_ok = q.out_is_c == 2 || (q.ldo % v == 0 && ((uintptr_t)q.out % (4 * v)) == 0);
        return n_cols % (64 * v) == 0 && q.ldb % v == 0 && out_ok && ((uintptr_t)q.B % (esz * v)) == 0;
    };
    // How many columns of C a wave takes = how many bytes of a row of B one gather instruction reads.  A workgroup column (grid y) is walked
    // to the end before the next starts, so the bytes per row set how many rows of B an XCD's 4 MiB L2 holds while the rows that reference
    // them go by: on power-law inputs the hub rows of B stay resident with narrow chunks and are pushed out by wide ones.  Measured
    // (R-MAT, 10 edges per row, fp32 N = 256 | bf16 N = 512; ms per product at 1024 / 512 / 256 bytes per row): 2^20 rows (|B| = 1 GiB)
    // 4.33 / 3.92 / 3.77 | - / 4.54 / 4.27; 2^18 rows (256 MiB) 0.93 / 0.87 / 0.89 | - / 1.02 / 1.03; 2^16 rows and below: widest wins
    // by 3-10 %.  2-byte loads (one bf16 per lane) are slow whatever the size (6.6 ms) and never chosen.  SPARTA_SP_VEC caps the width.
    static const int vec_cap = [] { const char* e = std::getenv("SPARTA_SP_VEC"); return e ? atoi(e) : 4; }();
    const double b_bytes = (double)A->cols * (double)n_cols * (double)esz;
    const int row_bytes_env = [] { const char* e = std::getenv("SPARTA_SP_ROW_BYTES"); return e ? atoi(e) : 0; }();      // (read per call: developer A/B)
    // (what matters is the SLICE of B one column chunk touches, cols x row_bytes: while that fits the Infinity Cache the widest chunk wins -- the reference's real
    // matrices at its operand width N = 8192, 8-58 k columns, |B| = 0.3-1.9 GB: 1.14 -> 0.83 ms (bcsstk18), 3.37 -> 2.81 (social_location) with 1024 instead of 256)
    const bool slice_fits = (double)A->cols * 1024.0 <= 96e6;
    const int row_bytes = row_bytes_env > 0 ? row_bytes_env : (slice_fits ? 1024 : (b_bytes > 384e6 ? 256 : (b_bytes > 96e6 ? 512 : 1024)));
    const int vec_want = std::max(esz == 2 ? 2 : 1, std::min(vec_cap, (int)(row_bytes / (64 * (int)esz))));
    const int vec = in_place ? 1 : (vec_want >= 4 && aligned(4) ? 4 : (vec_want >= 2 && aligned(2) ? 2 : 1));
    const unsigned gy = (unsigned)((n_cols + 64 * vec - 1) / (64 * vec));
    launch_sparse_kernels(vec, bk, q, gy, st, A->d_sp_list, A->n_sp_short, (const SpSegRec*)A->d_sp_segs, A->n_sp_segs, (const SpLongRec*)A->d_sp_long,
                          A->n_sp_long, (float*)A->d_sp_part, A->d_sp_stream_begin, A->sp_max_stream);
    HIP_TRY(hipGetLastError());
    return SPARTA_OK;
}

// 16-bit handles (SPARTA_F16 / SPARTA_BF16): A and B in the 16-bit type, fp32 accumulation, fp32 C.  Device pointers: B is a
// 16-bit column-major matrix (ldb in elements, even).  Host pointers keep the reference's contract (fp32 B in, fp32 C out):
// B is converted on the device (round to nearest even).  This is the product over whole 128-column slabs; spmm16_impl (below) cuts a
// call with any other n_cols into whole slabs + one padded tail slab.
int spmm16_core(sparta_vbs_t* A, const void* B, int64_t ldb, int32_t b_layout, int64_t shard_rows, int64_t shard_stride, int32_t n_cols, void* C,
                int64_t ldc, int32_t c_layout, int32_t accumulate, int32_t ptr_space, hipStream_t st, int32_t algo, float* dt_ms) {
    using sparta::fail;
    if (algo != SPARTA_SPMM_MFMA) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: SPARTA_SPMM_EXACT needs an fp32 handle");
    if (shard_rows != 0 && ptr_space != SPARTA_PTR_DEVICE) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: device pointers only");
    if (b_layout != SPARTA_COL_MAJOR) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: 16-bit handles need a column-major B (k contiguous)");
    if (n_cols % kTN != 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm: internal: the 16-bit core takes whole 128-column slabs");
    const bool bf16 = A->dtype == SPARTA_BF16;
    const size_t c_elems = (size_t)ldc * (size_t)(c_layout == SPARTA_COL_MAJOR ? n_cols : A->rows);
    const uint16_t* dB = (const uint16_t*)B;
    float* dC = (float*)C;
    int64_t ldb16 = ldb;
    if (ptr_space == SPARTA_PTR_HOST) {
        const size_t b_elems = (size_t)ldb * (size_t)n_cols;
        ldb16 = (A->cols + 7) / 8 * 8;
        if (int rc = ensure_scratch(&A->d_B, &A->d_B_bytes, b_elems * sizeof(float))) return rc;
        if (int rc = ensure_scratch(&A->d_B16, &A->d_B16_bytes, (size_t)ldb16 * n_cols * sizeof(uint16_t))) return rc;
        if (int rc = ensure_scratch(&A->d_C, &A->d_C_bytes, c_elems * sizeof(float))) return rc;
        HIP_TRY(hipMemcpyAsync(A->d_B, B, b_elems * sizeof(float), hipMemcpyHostToDevice, st));
        if (accumulate) HIP_TRY(hipMemcpyAsync(A->d_C, C, c_elems * sizeof(float), hipMemcpyHostToDevice, st));
        else if (c_elems > 0) HIP_TRY(hipMemsetAsync(A->d_C, 0, c_elems * sizeof(float), st));
        launch_convert_h16(bf16, st, (const float*)A->d_B, ldb, A->cols, (int64_t)n_cols, (uint16_t*)A->d_B16, ldb16);
        dB = (const uint16_t*)A->d_B16;
        dC = (float*)A->d_C;
    } else if (ldb % 2 != 0) {
        return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: 16-bit B needs an even leading dimension (16-byte loads start on 4-byte boundaries)");
    }
    // 32-bit byte offsets: the no-barrier kernel keeps them inside ONE WAVE's 32 columns (31 x ld x element size < 2^31: ldb < 34 M, ldc < 17 M elements;
    // its 64-column waves span 64 columns of B: ldb < 16 M);
    // the LDS-staged kernel (SPARTA_H16_PATH=lds) inside the 128-column slab (ldb < 8.3 M, ldc < 4.1 M)
    {
        const bool direct = h16_uses_direct_kernel(A->kp16, false) || A->wide16;
        const int64_t span_b = A->wide16 ? 64 : direct ? 32 : 128, span_c = direct ? 32 : 128;       // columns a wave's 32-bit offsets span (64-column waves: B only)
        if (ldb16 * span_b * 2 >= ((int64_t)1 << 31) - 65536 || (c_layout == SPARTA_ROW_MAJOR ? ldc * 32 : ldc * span_c) * 4 >= ((int64_t)1 << 31) - 65536)
            return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: leading dimension too large for the 16-bit stream kernels (ldb < 34 M -- 16 M for one-tile plans of "
                                                "32-wide blocks -- and ldc < 17 M elements; 8.3 M / 4.1 M with SPARTA_H16_PATH=lds)");
    }
    if (dt_ms) HIP_TRY(hipEventRecord(A->ev0, st));
    const int n_nt = n_cols / kTN;
    const bool prof = A->class_timing;
    for (int c = 0; c < 4; c++) A->class_ran[c] = false;
    const size_t slab = (size_t)A->n_slots * SK_SLOT_FLOATS;
    if (A->n_split > 0)
        if (int rc = ensure_scratch(&A->d_ws, &A->d_ws_bytes, slab * n_nt * sizeof(float))) return rc;
    StreamParams sp;
    sp.A = A->d_A; sp.B = (const float*)dB; sp.C = dC; sp.ws = (float*)A->d_ws;
    sp.ldb = ldb16; sp.ldc = ldc; sp.cols = A->cols; sp.shard_rows = shard_rows; sp.shard_stride = shard_stride;
    if (shard_rows > 0)
        if (int rc = ensure_gathered_steps(A, shard_rows, st)) return rc;
    sp.ws_slab_stride = (int64_t)slab; sp.accumulate = accumulate != 0; sp.c_row_major = c_layout == SPARTA_ROW_MAJOR;
    sp.N = n_cols; sp.w = (int32_t)A->w; sp.B_tail = nullptr; sp.clk = nullptr; sp.stagger = 0;
    if (A->n_steps[0] + A->n_steps[1] + A->n_hub_steps > 0) {
        if (prof) HIP_TRY(hipEventRecord(A->cev[0][0], st));
        if (A->has_tail && shard_rows == 0) {
            if (int rc = ensure_scratch(&A->d_btail, &A->d_btail_bytes, (size_t)A->w * n_cols * sizeof(uint16_t))) return rc;
            const int64_t row0 = ((A->cols - 1) / A->w) * A->w;
            launch_tail_copy_h16(st, dB, ldb16, row0, A->cols, (int)A->w, (int)n_cols, (uint16_t*)A->d_btail);
            sp.B_tail = (const float*)A->d_btail;
        }
        const dim3 grid((unsigned)A->n_workers, (unsigned)n_nt);
        const int probe_ty = A->n_steps[1] >= A->n_steps[0] ? 1 : 0;
        for (int ty = 1; ty >= 0; ty--) {
            if (A->n_steps[ty] == 0) continue;
            sp.steps = shard_rows > 0 ? A->d_steps_g[ty] : A->d_steps[ty]; sp.worker_range = A->d_wrange[ty]; sp.c_nt = c_store_nt(A, ty, sp.C, sp.ldc, sp.c_row_major != 0);
            sp.clk = (prof && ty == probe_ty) ? A->d_clk : nullptr;
            const bool gth = shard_rows > 0;
            // tiles of <= 32 rows of arbitrary height + column-major C: finished tiles wait in the LDS ring for whole aligned blocks (CRing)
            const int cst = [] { const char* e = std::getenv("SPARTA_CSTAGE"); return e ? atoi(e) : -1; }();
            const bool c_stage = ty == 0 && !gth && c_layout == SPARTA_COL_MAJOR && (cst >= 0 ? cst != 0 : ring_tiles(A)) && h16_uses_direct_kernel(A->kp16, false);
            if (c_stage) sp.c_nt = c_store_nt(A, ty, sp.C, sp.ldc, false, true);
            // 256-column slabs for the one-tile plan of 32-wide blocks (A read once per 256 columns; flagship matrix, N = 256: 42.2 -> 35.7 us, N = 512: 88.0 -> 77.0):
            // N % 256 == 0, no split tile, not through the ring, the plain plan (on the plan with two sub-worker ranges per workgroup a workgroup walks its tiles in
            // two ascending passes and the gain is gone: 41.1 us; allowed there with SPARTA_H16_SLAB256=2).  SPARTA_H16_SLAB256=0: off.
            const bool slab256 = ty == 0 && A->kp16 == 32 && !gth && !c_stage && A->n_split == 0 && n_cols % 256 == 0 && h16_uses_direct_kernel(32, false) && ldb16 * 64 * 2 < ((int64_t)1 << 31) - 65536 &&
                                 [&] { const char* e = std::getenv("SPARTA_H16_SLAB256"); const int m = e ? atoi(e) : 1; return m == 2 || (m == 1 && !A->wide16); }();
            // 64-row tiles of 64-wide blocks (the dense hub of a power-law matrix under the fixed 64 x 64 grid), N % 256 == 0: four accumulators per wave over
            // 256-column slabs -- A read once per 256 columns, every B fragment used twice (k_h16.hip, QUAD); split tiles allowed.  SPARTA_H16_QUAD=0: off.
            const bool quad = ty == 1 && n_cols % 256 == 0 && h16_uses_direct_kernel(A->kp16, true) && ldb16 * 64 * 2 < ((int64_t)1 << 31) - 65536 &&
                              [] { const char* e = std::getenv("SPARTA_H16_QUAD"); return !e || atoi(e) != 0; }();
            if (quad) {
                sp.sub_ranges = 0;
                launch_h16_quad(A->kp16, bf16, gth, dim3((unsigned)A->n_workers, (unsigned)(n_cols / 256)), st, sp);
            } else if (slab256) {
                sp.sub_ranges = A->wide16 ? 1 : 0;
                launch_h16_slab256(bf16, dim3((unsigned)A->n_workers, (unsigned)(n_cols / 256)), st, sp);
                sp.sub_ranges = 0;
            } else
                launch_h16_stream(A->kp16, ty != 0, bf16, gth, c_stage, ty == 0 && A->wide16, grid, st, sp);
        }
        // the hub plan: group tiles of 2 / 4 long 64-row tiles through the GEMM-shaped kernel (k_hub16.hip), 256-column slabs (the last one may be a half slab)
        if (A->n_hub_steps > 0) {
            HubParams hp;
            hp.steps = shard_rows > 0 ? A->d_hub_steps_g : A->d_hub_steps; hp.worker_range = A->d_hub_wrange; hp.tiles = A->d_hub_tiles; hp.A = A->d_hub_A;
            hp.B = dB; hp.B_tail = (const uint16_t*)sp.B_tail; hp.C = dC; hp.ws = (float*)A->d_ws;
            hp.ldb = ldb16; hp.ldc = ldc; hp.shard_stride = shard_stride; hp.ws_slab_stride = (int64_t)slab;
            hp.accumulate = accumulate != 0; hp.c_row_major = c_layout == SPARTA_ROW_MAJOR; hp.c_nt = 1; hp.w = (int32_t)A->w;
            hp.n_slabs = (n_cols + 255) / 256; hp.n_workers = A->hub_workers; hp.n_cols = n_cols; hp.pad0 = 0;
            static const int hub_variant_env = [] { const char* e = std::getenv("SPARTA_HUB_VARIANT"); return e ? atoi(e) : -1; }();
            int variant = A->hub_g == 4 ? 10 : 0;                        // KP 64: G = 4 two stages (eight waves), G = 2 three stages
            if (hub_variant_env >= 0 && hub_variant_kp(hub_variant_env) == 64 && hub_variant_g(hub_variant_env) == A->hub_g) variant = hub_variant_env;
            // a wave's panel loads span 64 (G = 2) or 32 (G = 4) columns of B in 32-bit buffer offsets: the leading dimension must keep them inside the descriptor's 2 GB
            if (ldb16 * (int64_t)(hub_variant_g(variant) == 2 ? 64 : 32) * 2 >= ((int64_t)1 << 31) - 65536)
                return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: leading dimension of B too large for the 16-bit hub kernel (32-bit panel offsets)");
            launch_h16_hub(variant, bf16, shard_rows > 0, st, hp);
        }
        if (prof) { HIP_TRY(hipEventRecord(A->cev[0][1], st)); A->class_ran[0] = true; }
    }
    const bool fix_launch = A->n_fix > 0 && !(accumulate && A->n_split == 0);   // C += 0 for the block-rows without blocks: nothing to launch
    const bool zero_launch = !A->zero_ranges.empty() && !accumulate;
    if (fix_launch || zero_launch) {
        if (prof) HIP_TRY(hipEventRecord(A->cev[1][0], st));
        if (fix_launch && A->n_big_fix > 0)
            launch_fixup_group(dim3((unsigned)A->n_big_fix, (unsigned)n_nt, (unsigned)((A->max_tile_slots + kFixGroup - 1) / kFixGroup)), st, A->d_fix,
                               A->d_big_fix, A->d_fix_slots, (float*)A->d_ws, (int64_t)slab);
        if (fix_launch)
            launch_fixup(dim3((unsigned)A->n_fix, (unsigned)n_nt), st, A->d_fix, A->d_fix_slots, (const float*)A->d_ws, (int64_t)slab, dC, ldc,
                         (int)(c_layout == SPARTA_ROW_MAJOR), (int)(accumulate != 0));
        if (zero_launch) launch_zero_ranges(A, dC, ldc, c_layout == SPARTA_ROW_MAJOR, n_cols, st);
        if (prof) { HIP_TRY(hipEventRecord(A->cev[1][1], st)); A->class_ran[1] = true; }
    }
    // (a 16-bit product runs this core once per piece of B -- the whole slabs, then the padded tail slab: each piece has its own row-major copy, made by whoever needs it first)
    A->brm_ready = nullptr; A->brm_ld = 0;
    struct BrmPiece { sparta_vbs_t* a; ~BrmPiece() { a->brm_ready = nullptr; a->brm_ld = 0; } } brm_piece{A};
    if (A->u_steps_total > 0) {   // column-compacted tiles (k_union.hip, 16-bit form): they store every row of their block-rows; those block-rows' sparse rows add, below
        const bool rec = prof && !A->class_ran[2];
        if (rec) HIP_TRY(hipEventRecord(A->cev[2][0], st));
        if (int rc = launch_union_tiles(A, dB, ldb16, false, shard_rows, shard_stride, n_cols, dC, ldc, c_layout == SPARTA_ROW_MAJOR, accumulate != 0, st)) return rc;
        if (rec) { HIP_TRY(hipEventRecord(A->cev[2][1], st)); A->class_ran[2] = true; }
    }
    if (A->n_sp_rows > 0) {                  // nearly empty block-rows: sparse rows over a row-major 16-bit copy of B
        if (prof) HIP_TRY(hipEventRecord(A->cev[3][0], st));
        if (int rc = launch_sparse_rows(A, dB, ldb16, false, shard_rows, shard_stride, bf16 ? 2 : 1, n_cols, dC, ldc, c_layout == SPARTA_ROW_MAJOR, accumulate != 0, st)) return rc;
        if (prof) { HIP_TRY(hipEventRecord(A->cev[3][1], st)); A->class_ran[3] = true; }
    }
    A->last_path = 1;
    HIP_TRY(hipGetLastError());
    if (dt_ms) {
        HIP_TRY(hipEventRecord(A->ev1, st));
        HIP_TRY(hipEventSynchronize(A->ev1));
        HIP_TRY(hipEventElapsedTime(dt_ms, A->ev0, A->ev1));
    }
    if (ptr_space == SPARTA_PTR_HOST) {
        HIP_TRY(hipMemcpyAsync(C, A->d_C, c_elems * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SPARTA_OK;
}

// Any n_cols on a 16-bit handle (the reference's `-c` is arbitrary, include/input.h:15-42): the whole 128-column slabs go through the
// kernels as they are; the last n_cols % 128 columns of B are copied (they are contiguous in a column-major B: one copy, per slab of a
// gathered B one row of a 2-D copy) into a zero-padded 128-column slab, multiplied by the same kernels into a scratch slab of C, and their
// n_cols % 128 columns merged into the caller's C (either layout, accumulate or not).  Costs scratch of 128 columns of B and of C, only
// on calls that have such a tail.
int spmm16_impl(sparta_vbs_t* A, const void* B, int64_t ldb, int32_t b_layout, int64_t shard_rows, int64_t shard_stride, int32_t n_cols, void* C,
                int64_t ldc, int32_t c_layout, int32_t accumulate, int32_t ptr_space, hipStream_t st, int32_t algo, float* dt_ms) {
    using sparta::fail;
    if (n_cols % kTN == 0) return spmm16_core(A, B, ldb, b_layout, shard_rows, shard_stride, n_cols, C, ldc, c_layout, accumulate, ptr_space, st, algo, dt_ms);
    if (algo != SPARTA_SPMM_MFMA) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: SPARTA_SPMM_EXACT needs an fp32 handle");
    if (shard_rows != 0 && ptr_space != SPARTA_PTR_DEVICE) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: device pointers only");
    if (b_layout != SPARTA_COL_MAJOR) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: 16-bit handles need a column-major B (k contiguous)");
    const bool bf16 = A->dtype == SPARTA_BF16;
    const int32_t n_main = n_cols / kTN * kTN, n_t = n_cols - n_main;
    const size_t c_elems = (size_t)ldc * (size_t)(c_layout == SPARTA_COL_MAJOR ? n_cols : A->rows);
    const uint16_t* dB = (const uint16_t*)B;
    float* dC = (float*)C;
    int64_t ldb16 = ldb;
    if (ptr_space == SPARTA_PTR_HOST) {                    // stage here (the core's own staging assumes whole slabs)
        const size_t b_elems = (size_t)ldb * (size_t)n_cols;
        ldb16 = (A->cols + 7) / 8 * 8;
        if (int rc = ensure_scratch(&A->d_B, &A->d_B_bytes, b_elems * sizeof(float))) return rc;
        if (int rc = ensure_scratch(&A->d_B16, &A->d_B16_bytes, (size_t)ldb16 * n_cols * sizeof(uint16_t))) return rc;
        if (int rc = ensure_scratch(&A->d_C, &A->d_C_bytes, c_elems * sizeof(float))) return rc;
        HIP_TRY(hipMemcpyAsync(A->d_B, B, b_elems * sizeof(float), hipMemcpyHostToDevice, st));
        if (accumulate) HIP_TRY(hipMemcpyAsync(A->d_C, C, c_elems * sizeof(float), hipMemcpyHostToDevice, st));
        else if (c_elems > 0) HIP_TRY(hipMemsetAsync(A->d_C, 0, c_elems * sizeof(float), st));
        launch_convert_h16(bf16, st, (const float*)A->d_B, ldb, A->cols, (int64_t)n_cols, (uint16_t*)A->d_B16, ldb16);
        dB = (const uint16_t*)A->d_B16;
        dC = (float*)A->d_C;
    } else if (ldb % 2 != 0 && shard_rows == 0) {
        return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: 16-bit B needs an even leading dimension (16-byte loads start on 4-byte boundaries)");
    }
    if (dt_ms) HIP_TRY(hipEventRecord(A->tev0, st));
    // (1) the whole slabs
    if (n_main > 0)
        if (int rc = spmm16_core(A, dB, ldb16, b_layout, shard_rows, shard_stride, n_main, dC, ldc, c_layout, accumulate, SPARTA_PTR_DEVICE, st, algo, nullptr)) return rc;
    // (2) the tail slab: n_t columns of B + zero columns
    const int64_t n_shards = shard_rows > 0 ? A->cols / shard_rows : 1;
    const int64_t ld_t = shard_rows > 0 ? shard_rows : ldb16;                     // rows of a column of the tail slab of B
    const size_t bt_elems = (size_t)ld_t * kTN * (size_t)n_shards;
    if (int rc = ensure_scratch(&A->d_Bt, &A->d_Bt_bytes, bt_elems * sizeof(uint16_t))) return rc;
    if (int rc = ensure_scratch(&A->d_Ct, &A->d_Ct_bytes, (size_t)A->rows * kTN * sizeof(float))) return rc;
    HIP_TRY(hipMemsetAsync(A->d_Bt, 0, bt_elems * sizeof(uint16_t), st));
    if (shard_rows > 0 && ldb16 == shard_rows)
        HIP_TRY(hipMemcpy2DAsync(A->d_Bt, (size_t)shard_rows * kTN * sizeof(uint16_t), dB + (size_t)n_main * shard_rows, (size_t)shard_stride * sizeof(uint16_t),
                                 (size_t)n_t * shard_rows * sizeof(uint16_t), (size_t)n_shards, hipMemcpyDeviceToDevice, st));
    else if (shard_rows > 0)                                  // columns of a slab ldb16 > shard_rows apart (sparta_vbs_spmm_gathered_ld): one 2-D copy per slab
        for (int64_t sh = 0; sh < n_shards; sh++)
            HIP_TRY(hipMemcpy2DAsync((uint16_t*)A->d_Bt + (size_t)sh * shard_rows * kTN, (size_t)shard_rows * sizeof(uint16_t), dB + (size_t)sh * shard_stride + (size_t)n_main * ldb16,
                                     (size_t)ldb16 * sizeof(uint16_t), (size_t)shard_rows * sizeof(uint16_t), (size_t)n_t, hipMemcpyDeviceToDevice, st));
    else
        HIP_TRY(hipMemcpyAsync(A->d_Bt, dB + (size_t)n_main * ldb16, ((size_t)(n_t - 1) * ldb16 + (size_t)A->cols) * sizeof(uint16_t), hipMemcpyDeviceToDevice, st));   // (not past the last column's rows: a caller may have allocated ldb * (n_cols - 1) + cols)
    {
        // the tail slab is a B of its own (zero-padded copy of the last columns): a prepared row-major copy of the caller's B does not describe it
        struct Unprepare { sparta_vbs_t* a; const void* keep; ~Unprepare() { a->prepared_brm = keep; } } unprep{A, A->prepared_brm};
        A->prepared_brm = nullptr;
        if (int rc = spmm16_core(A, A->d_Bt, ld_t, b_layout, shard_rows, shard_rows > 0 ? shard_rows * kTN : 0, kTN, A->d_Ct, A->rows, SPARTA_COL_MAJOR, 0,
                                 SPARTA_PTR_DEVICE, st, algo, nullptr)) return rc;
    }
    launch_col_tail_merge(st, (const float*)A->d_Ct, A->rows, dC, ldc, (int)(c_layout == SPARTA_ROW_MAJOR), n_main, n_t, (int)(accumulate != 0));
    HIP_TRY(hipGetLastError());
    if (dt_ms) {
        HIP_TRY(hipEventRecord(A->tev1, st));
        HIP_TRY(hipEventSynchronize(A->tev1));
        HIP_TRY(hipEventElapsedTime(dt_ms, A->tev0, A->tev1));
    }
    if (ptr_space == SPARTA_PTR_HOST) {
        HIP_TRY(hipMemcpyAsync(C, A->d_C, c_elems * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SPARTA_OK;
}

int spmm_impl(sparta_vbs_t* A, const void* B, int64_t ldb, int32_t b_layout, int64_t shard_rows, int64_t shard_stride, int32_t n_cols,
              void* C, int64_t ldc, int32_t c_layout, int32_t accumulate, int32_t ptr_space, void* stream, int32_t algo,
              float* dt_ms) {
    using sparta::fail;
    if (!A || !B || !C) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm: NULL argument");
    if (n_cols <= 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm: n_cols must be > 0");
    if (b_layout != SPARTA_COL_MAJOR && b_layout != SPARTA_ROW_MAJOR) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm: bad b_layout");
    if (c_layout != SPARTA_COL_MAJOR && c_layout != SPARTA_ROW_MAJOR) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm: bad c_layout");
    if (shard_rows == 0 && ldb < (b_layout == SPARTA_COL_MAJOR ? A->cols : (int64_t)n_cols))
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm: ldb too small");
    if (ldc < (c_layout == SPARTA_COL_MAJOR ? A->rows : (int64_t)n_cols)) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm: ldc too small");
    if (algo != SPARTA_SPMM_MFMA && algo != SPARTA_SPMM_EXACT) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm: bad algo");
    if (ptr_space != SPARTA_PTR_HOST && ptr_space != SPARTA_PTR_DEVICE) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm: bad ptr_space");

    DeviceGuard guard(A->device);
    if (!guard.ok) return fail(SPARTA_ERR_HIP, "sparta_vbs_spmm: hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const CaptureScope capture(st, ptr_space == SPARTA_PTR_DEVICE);
    if (g_capturing && dt_ms) return capture_refusal("time the product (dt_ms != NULL synchronises)");
    if (A->dtype != SPARTA_F32)
        return spmm16_impl(A, B, ldb, b_layout, shard_rows, shard_stride, n_cols, C, ldc, c_layout, accumulate, ptr_space, st, algo, dt_ms);

    const float* dB = (const float*)B;
    float* dC = (float*)C;
    const size_t b_elems = (size_t)ldb * (size_t)(b_layout == SPARTA_COL_MAJOR ? n_cols : A->cols);
    const size_t c_elems = (size_t)ldc * (size_t)(c_layout == SPARTA_COL_MAJOR ? n_cols : A->rows);
    if (ptr_space == SPARTA_PTR_HOST) {
        // the reference's back-end contract: host in, host out, dt excludes the copies
        if (int rc = ensure_scratch(&A->d_B, &A->d_B_bytes, b_elems * sizeof(float))) return rc;
        if (int rc = ensure_scratch(&A->d_C, &A->d_C_bytes, c_elems * sizeof(float))) return rc;
        HIP_TRY(hipMemcpyAsync(A->d_B, B, b_elems * sizeof(float), hipMemcpyHostToDevice, st));
        if (accumulate) HIP_TRY(hipMemcpyAsync(A->d_C, C, c_elems * sizeof(float), hipMemcpyHostToDevice, st));
        else if (c_elems > 0) HIP_TRY(hipMemsetAsync(A->d_C, 0, c_elems * sizeof(float), st));   // ld padding stays defined
        dB = (const float*)A->d_B;
        dC = (float*)A->d_C;
    }

    struct BrmScope { sparta_vbs_t* a; ~BrmScope() { a->brm_ready = nullptr; a->brm_ld = 0; } } brm_scope{A};     // (the row-major B belongs to this product only)
    A->brm_ready = nullptr; A->brm_ld = 0;
    if (algo == SPARTA_SPMM_EXACT && A->ext_sparse && (A->n_sp_rows > 0 || A->u_steps_total > 0))
        return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: SPARTA_SPMM_EXACT needs the dense image of every block-row (handle made by sparta_vbs_create_from_csr)");
    if (dt_ms) HIP_TRY(hipEventRecord(A->ev0, st));
    if (algo == SPARTA_SPMM_EXACT) {
        if (int rc = ensure_legacy_image(A, st)) return rc;
        if (A->n_brows > 0) {
            launch_f32_exact((unsigned)A->n_brows, st, A->d_brows, A->d_jab, A->d_A, dB, dC, ldb, ldc, A->cols, (int)n_cols, (int)A->w,
                             (int)(b_layout == SPARTA_ROW_MAJOR), (int)(c_layout == SPARTA_ROW_MAJOR), (int)accumulate, shard_rows, shard_stride);
        }
    } else {
        const char* nv = std::getenv("SPARTA_NO_VEC");
        const bool novec = nv && nv[0] == '1';
        const int n_nt = (n_cols + kTN - 1) / kTN;
        const bool full_slabs = (n_cols % kTN) == 0;
        // two product paths (both branch-free, both need full panels) + the generic fallback for odd shapes
        // The stream kernels address B and C through buffer descriptors with 32-bit byte offsets inside one 128-column slab
        // (range-checked against 2 GB): 127 columns x leading dimension x 4 bytes must stay below 2^31, i.e. ld < 4.2 M
        // elements for the column-major layouts.  Larger leading dimensions take the per-class / generic kernels (64-bit
        // pointer arithmetic).
        const int64_t ld_lim = ((int64_t)1 << 31) / (128 * 4) - 64;
        const bool ld_ok = (b_layout == SPARTA_ROW_MAJOR ? ldb * 32 : ldb * 128) < ((int64_t)1 << 29) - 4096 &&
                           (c_layout == SPARTA_ROW_MAJOR ? ldc * 64 : ldc * 128) < ((int64_t)1 << 29) - 4096;
        (void)ld_lim;
        const bool can_stream = A->n_workers > 0 && full_slabs && !novec && !force_generic() && ld_ok;
        const bool can_class = (A->w % kKP) == 0 && full_slabs && !novec && !force_generic();

        // persistent stream kernel + fix-up of the split tiles
        auto run_stream = [&](float* Cout, bool prof) -> int {
            const size_t slab = (size_t)A->n_slots * SK_SLOT_FLOATS;
            if (A->n_split > 0)
                if (int rc = ensure_scratch(&A->d_ws, &A->d_ws_bytes, slab * n_nt * sizeof(float))) return rc;
            if (shard_rows > 0)
                if (int rc = ensure_gathered_steps(A, shard_rows, st)) return rc;
            // the LDS-staged kernels (33..64-row tiles; any tile under a row-major or gathered B) read the reference-layout image; the no-barrier kernel the fragment image
            if (A->n_steps[1] > 0 || !(A->d_a_frag && b_layout == SPARTA_COL_MAJOR && shard_rows == 0))
                if (int rc = ensure_legacy_image(A, st)) return rc;
            StreamParams sp;
            sp.A = A->d_A; sp.B = dB; sp.C = Cout; sp.ws = (float*)A->d_ws;
            sp.ldb = ldb; sp.ldc = ldc; sp.cols = A->cols; sp.shard_rows = shard_rows; sp.shard_stride = shard_stride;
            sp.ws_slab_stride = (int64_t)slab; sp.accumulate = accumulate != 0; sp.c_row_major = c_layout == SPARTA_ROW_MAJOR;
            sp.N = n_cols; sp.w = (int32_t)A->w; sp.B_tail = nullptr;
            sp.clk = nullptr;
            sp.stagger = [] { const char* e = std::getenv("SPARTA_STAGGER"); return e ? atoi(e) : 0; }();
            if (A->n_steps[0] + A->n_steps[1] > 0) {
                if (prof) HIP_TRY(hipEventRecord(A->cev[0][0], st));
                if (A->has_tail && shard_rows == 0) {
                    if (int rc = ensure_scratch(&A->d_btail, &A->d_btail_bytes, (size_t)A->w * n_cols * sizeof(float))) return rc;
                    const int64_t row0 = ((A->cols - 1) / A->w) * A->w;
                    launch_tail_copy(st, dB, ldb, (int)(b_layout == SPARTA_ROW_MAJOR), row0, A->cols, (int)A->w, (int)n_cols, (float*)A->d_btail);
                    sp.B_tail = (const float*)A->d_btail;
                }
                const dim3 grid((unsigned)A->n_workers, (unsigned)n_nt);
                // heavier type first; the clock probe rides on the launch with more steps
                const int probe_ty = A->n_steps[1] >= A->n_steps[0] ? 1 : 0;
                for (int ty = 1; ty >= 0; ty--) {
                    if (A->n_steps[ty] == 0) continue;
                    sp.steps = shard_rows > 0 ? A->d_steps_g[ty] : A->d_steps[ty]; sp.worker_range = A->d_wrange[ty]; sp.c_nt = c_store_nt(A, ty, sp.C, sp.ldc, sp.c_row_major != 0);
                    sp.clk = (prof && ty == probe_ty) ? A->d_clk : nullptr;
                    if (ty == 0 && A->d_a_frag && b_layout == SPARTA_COL_MAJOR && shard_rows == 0) {
                        StreamParams sd = sp;                 // the <= 32-row tiles without the workgroup stage
                        sd.A = A->d_a_frag;
                        // tiles of arbitrary height + column-major C: finished tiles wait in an LDS ring for whole aligned blocks of 32 rows (k_f32_direct.hip)
                        const int cst = [] { const char* e = std::getenv("SPARTA_CSTAGE"); return e ? atoi(e) : -1; }();     // (read per launch: tests flip it)
                        const bool c_stage = c_layout == SPARTA_COL_MAJOR && (cst >= 0 ? cst != 0 : ring_tiles(A));
                        sd.c_nt = c_store_nt(A, ty, sd.C, sd.ldc, sd.c_row_major != 0, c_stage);
                        launch_f32_direct(c_stage, grid, st, sd);
                    } else launch_f32_stream(ty != 0, b_layout == SPARTA_ROW_MAJOR, shard_rows > 0, grid, st, sp);
                }
                if (prof) { HIP_TRY(hipEventRecord(A->cev[0][1], st)); A->class_ran[0] = true; }
            }
            const bool fix_launch = A->n_fix > 0 && !(accumulate && A->n_split == 0);   // C += 0 for the block-rows without blocks: nothing to launch
            const bool zero_launch = !A->zero_ranges.empty() && !accumulate;
            if (fix_launch || zero_launch) {
                if (prof) HIP_TRY(hipEventRecord(A->cev[1][0], st));
                if (fix_launch && A->n_big_fix > 0)
                    launch_fixup_group(dim3((unsigned)A->n_big_fix, (unsigned)n_nt, (unsigned)((A->max_tile_slots + kFixGroup - 1) / kFixGroup)), st, A->d_fix,
                                       A->d_big_fix, A->d_fix_slots, (float*)A->d_ws, (int64_t)slab);
                if (fix_launch)
                    launch_fixup(dim3((unsigned)A->n_fix, (unsigned)n_nt), st, A->d_fix, A->d_fix_slots, (const float*)A->d_ws, (int64_t)slab, Cout, ldc,
                                 (int)(c_layout == SPARTA_ROW_MAJOR), (int)(accumulate != 0));
                if (zero_launch) launch_zero_ranges(A, Cout, ldc, c_layout == SPARTA_ROW_MAJOR, n_cols, st);
                if (prof) { HIP_TRY(hipEventRecord(A->cev[1][1], st)); A->class_ran[1] = true; }
            }
            return SPARTA_OK;
        };
        // one launch per tile class (<=16 / <=32 / <=64 rows), one workgroup per tile
        auto run_class = [&](float* Cout, bool generic, bool prof) -> int {
            if (int rc = ensure_legacy_image(A, st)) return rc;
            SpmmParams p;
            p.jab = A->d_jab; p.A = A->d_A; p.B = dB; p.C = Cout; p.ldb = ldb; p.ldc = ldc; p.cols = A->cols;
            p.n_ntiles = n_nt; p.N = n_cols; p.w = (int32_t)A->w;
            p.b_row_major = b_layout == SPARTA_ROW_MAJOR; p.c_row_major = c_layout == SPARTA_ROW_MAJOR;
            p.accumulate = accumulate != 0;
            p.shard_rows = shard_rows; p.shard_stride = shard_stride;
            p.vec_ok = novec ? 0 : 1;
            for (int c = 3; c >= 0; c--) {              // heavy classes first
                if (A->n_tiles[c] == 0) continue;
                if (A->n_tiles[c] * (int64_t)p.n_ntiles > INT32_MAX) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: grid too large");
                p.tiles = A->d_tiles[c]; p.n_tiles = (int32_t)A->n_tiles[c];
                p.clk = prof ? A->d_clk + 4 * c : nullptr;
                if (prof) HIP_TRY(hipEventRecord(A->cev[c][0], st));
                launch_f32_class(c, p.b_row_major != 0, generic, p, st);
                if (prof) { HIP_TRY(hipEventRecord(A->cev[c][1], st)); A->class_ran[c] = true; }
            }
            return SPARTA_OK;
        };

        // ---- choose the path: forced by SPARTA_PATH, else measured once per (n_cols, layouts, gathered) on this handle -------
        int path = 3;                                   // 1 stream, 2 per-class fast, 3 generic
        const char* pe = std::getenv("SPARTA_PATH");
        const std::string forced = pe ? pe : "auto";
        if (forced == "stream") path = can_stream ? 1 : 3;
        else if (forced == "class") path = can_class ? 2 : 3;
        else if (forced == "generic") path = 3;
        else if (can_stream && can_class) {
            const int64_t key = ((int64_t)n_cols << 8) | (b_layout << 2) | (c_layout << 1) | (shard_rows > 0 ? 1 : 0);
            path = 0;
            for (const auto& kv : A->tuned) if (kv.first == key) path = kv.second;
            if (path == 0 && g_capturing) return capture_refusal("time its two product paths");
            if (path == 0) {
                // plan-time autotune: both paths write a scratch C (the caller's C must not be accumulated into twice)
                if (int rc = ensure_scratch(&A->d_tune, &A->d_tune_bytes, c_elems * sizeof(float))) return rc;
                float best[3] = {0.0f, 1e30f, 1e30f};
                for (int cand = 1; cand <= 2; cand++) {
                    for (int rep = 0; rep < 4; rep++) {
                        HIP_TRY(hipEventRecord(A->tev0, st));
                        if (int rc = cand == 1 ? run_stream((float*)A->d_tune, false) : run_class((float*)A->d_tune, false, false)) return rc;
                        HIP_TRY(hipEventRecord(A->tev1, st));
                        HIP_TRY(hipEventSynchronize(A->tev1));
                        float ms = 0.0f;
                        HIP_TRY(hipEventElapsedTime(&ms, A->tev0, A->tev1));
                        if (rep > 0) best[cand] = std::min(best[cand], ms);
                    }
                }
                path = best[1] <= best[2] ? 1 : 2;
                A->tuned.emplace_back(key, path);
                A->tune_ms[0] = best[1]; A->tune_ms[1] = best[2];
            }
        } else if (can_stream) path = 1;
        else if (can_class) path = 2;

        const bool prof = A->class_timing;
        for (int c = 0; c < 4; c++) A->class_ran[c] = false;
        if (path == 1) { if (int rc = run_stream(dC, prof)) return rc; }
        else if (int rc = run_class(dC, path == 3, prof)) return rc;
        A->last_path = path;
        // one image of A, not two: the no-barrier kernel carries this handle's products (every tile is in the fragment image) -- see ensure_legacy_image
        if (path == 1 && b_layout == SPARTA_COL_MAJOR && shard_rows == 0) drop_legacy_image(A);

        // ---- the sparse rows.  Fully sparse block-rows (flag 1) own their rows of C; rows of MIXED block-rows (flag 2, bit 31 of crow) ADD to what
        // the tile and fix-up launches stored: this launch must stay BEHIND those launches on the same stream (moving the sparse leg to a side
        // stream for overlap needs an event wait for the mixed rows) ----
        // ---- the column-compacted tiles (fp32 handles made from a CSR): they store every row of their block-rows; those block-rows' sparse rows add, below ----
        if (A->u_steps_total > 0) {
            const bool rec = prof && !A->class_ran[2];
            if (rec) HIP_TRY(hipEventRecord(A->cev[2][0], st));
            if (int rc = launch_union_tiles(A, dB, ldb, b_layout == SPARTA_ROW_MAJOR, shard_rows, shard_stride, n_cols, dC, ldc, c_layout == SPARTA_ROW_MAJOR,
                                            accumulate != 0, st)) return rc;
            if (rec) { HIP_TRY(hipEventRecord(A->cev[2][1], st)); A->class_ran[2] = true; }
        }
        if (A->n_sp_rows > 0) {
            if (prof) HIP_TRY(hipEventRecord(A->cev[3][0], st));
            if (int rc = launch_sparse_rows(A, dB, ldb, b_layout == SPARTA_ROW_MAJOR, shard_rows, shard_stride, 0, n_cols, dC, ldc,
                                            c_layout == SPARTA_ROW_MAJOR, accumulate != 0, st)) return rc;
            if (prof) { HIP_TRY(hipEventRecord(A->cev[3][1], st)); A->class_ran[3] = true; }
        }
    }
    HIP_TRY(hipGetLastError());
    if (dt_ms) {
        HIP_TRY(hipEventRecord(A->ev1, st));
        HIP_TRY(hipEventSynchronize(A->ev1));
        HIP_TRY(hipEventElapsedTime(dt_ms, A->ev0, A->ev1));
    }
    if (ptr_space == SPARTA_PTR_HOST) {
        HIP_TRY(hipMemcpyAsync(C, A->d_C, c_elems * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SPARTA_OK;
}

}  // namespace

extern "C" {

int sparta_vbs_spmm(sparta_vbs_t* A, const void* B, int64_t ldb, int32_t b_layout, int32_t n_cols, void* C, int64_t ldc,
                    int32_t c_layout, int32_t accumulate, int32_t ptr_space, void* stream, int32_t algo, float* dt_ms) {
    SPARTA_GUARD_BEGIN
    return spmm_impl(A, B, ldb, b_layout, 0, 0, n_cols, C, ldc, c_layout, accumulate, ptr_space, stream, algo, dt_ms);
    SPARTA_GUARD_END("sparta_vbs_spmm")
}

int sparta_vbs_spmm_gathered_ld(sparta_vbs_t* A, const void* B_gathered, int64_t shard_rows, int64_t shard_ld, int64_t shard_stride, int32_t n_cols,
                                void* C, int64_t ldc, int32_t c_layout, int32_t accumulate, void* stream, int32_t algo, float* dt_ms) {
    using sparta::fail;
    if (!A) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: NULL handle");
    if (shard_rows <= 0 || shard_rows % A->w != 0)
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: shard_rows must be a positive multiple of block_col_size");
    if (shard_ld < shard_rows) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: shard_ld must be >= shard_rows");
    if (shard_stride < shard_ld * (int64_t)(n_cols - 1) + shard_rows) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: shard_stride too small");
    if (A->cols % shard_rows != 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: cols must be n_shards * shard_rows");
    SPARTA_GUARD_BEGIN
    return spmm_impl(A, B_gathered, shard_ld, SPARTA_COL_MAJOR, shard_rows, shard_stride, n_cols, C, ldc, c_layout, accumulate,
                     SPARTA_PTR_DEVICE, stream, algo, dt_ms);
    SPARTA_GUARD_END("sparta_vbs_spmm_gathered")
}

int sparta_vbs_spmm_gathered(sparta_vbs_t* A, const void* B_gathered, int64_t shard_rows, int64_t shard_stride, int32_t n_cols,
                             void* C, int64_t ldc, int32_t c_layout, int32_t accumulate, void* stream, int32_t algo, float* dt_ms) {
    if (shard_rows > 0 && shard_stride < shard_rows * (int64_t)n_cols) return sparta::fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: shard_stride too small");
    return sparta_vbs_spmm_gathered_ld(A, B_gathered, shard_rows, shard_rows, shard_stride, n_cols, C, ldc, c_layout, accumulate, stream, algo, dt_ms);
}

// ---- a B that does not change between products, prepared once ------------------------------------------------------------
struct sparta_b {
    const void* B = nullptr;
    int64_t ldb = 0, shard_rows = 0, shard_stride = 0, cols = 0;
    int32_t n_cols = 0, dtype = 0, device = 0;
    void* d_Brm = nullptr;                 // row-major copy for the sparse-row kernels and the column-compacted tiles (nullptr: this handle / shape does not need one)
    int64_t ld_brm = 0;                    // its row stride in elements (n_cols rounded up to 16 bytes)
};

int sparta_vbs_prepare_b(sparta_vbs_t* A, const void* B, int64_t ldb, int64_t shard_rows, int64_t shard_stride, int32_t n_cols, void* stream,
                         sparta_b_t** out) {
    using sparta::fail;
    SPARTA_GUARD_BEGIN
    if (!A || !B || !out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_prepare_b: NULL argument");
    *out = nullptr;
    if (n_cols <= 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_prepare_b: n_cols must be > 0");
    if (shard_rows == 0 && ldb < A->cols) return fail(SPARTA_ERR_INVALID, "sparta_vbs_prepare_b: ldb too small");
    // a gathered B: `ldb` is the column stride inside a slab (sparta_vbs_spmm_gathered_ld's shard_ld; 0 = unpadded, shard_rows)
    const int64_t shard_ld = shard_rows != 0 ? (ldb > 0 ? ldb : shard_rows) : 0;
    if (shard_rows != 0 && (shard_rows < 0 || shard_rows % A->w != 0 || A->cols % shard_rows != 0 || shard_ld < shard_rows ||
                            shard_stride < shard_ld * (int64_t)(n_cols - 1) + shard_rows))
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_prepare_b: bad gathered layout");
    if (g_capturing) return capture_refusal("prepare a copy of B");
    DeviceGuard guard(A->device);
    sparta_b* p = new (std::nothrow) sparta_b;
    if (!p) return fail(SPARTA_ERR_ALLOC, "sparta_vbs_prepare_b: out of host memory");
    p->B = B; p->ldb = shard_rows ? shard_ld : ldb; p->shard_rows = shard_rows; p->shard_stride = shard_stride; p->cols = A->cols;
    p->n_cols = n_cols; p->dtype = A->dtype; p->device = A->device;
    const size_t esz = A->dtype == SPARTA_F32 ? 4 : 2;
    // the sparse-row kernels of this handle would transpose this B per product (launch_sparse_rows: not when a handful of rows reads it in place)
    // (nor when the resident-column kernel carries the products of a plain column-major B: it reads the columns where they are; a later product into a ROW-major C
    // then transposes per call as sparta_vbs_spmm does)
    const bool needs_copy = (A->n_sp_rows > 0 && !(A->sp_nnz * 8 < A->cols) && !(A->cr_slices > 0 && shard_rows == 0)) || A->u_steps_total > 0;
    if (needs_copy) {
        hipStream_t st = (hipStream_t)stream;
        const int64_t n_wg = ((A->cols + 63) / 64) * (int64_t)((n_cols + 63) / 64);
        if (n_wg > INT32_MAX) { delete p; return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_prepare_b: B too large for the transpose grid"); }
        const int64_t per16 = 16 / (int64_t)esz;
        p->ld_brm = (n_cols + per16 - 1) / per16 * per16;
        if (hipMalloc(&p->d_Brm, (size_t)A->cols * (size_t)p->ld_brm * esz) != hipSuccess) { delete p; return fail(SPARTA_ERR_ALLOC, "sparta_vbs_prepare_b: out of device memory"); }
        launch_b_to_row_major(A->dtype != SPARTA_F32, (unsigned)n_wg, st, B, p->ldb, shard_rows, shard_stride, A->cols, (int)n_cols, p->d_Brm, p->ld_brm);
        if (hipGetLastError() != hipSuccess) { (void)hipFree(p->d_Brm); delete p; return fail(SPARTA_ERR_HIP, "sparta_vbs_prepare_b: launch failed"); }
    }
    *out = p;
    return SPARTA_OK;
    SPARTA_GUARD_END("sparta_vbs_prepare_b")
}

int sparta_vbs_spmm_prepared(sparta_vbs_t* A, const sparta_b_t* Bp, void* C, int64_t ldc, int32_t c_layout, int32_t accumulate, void* stream,
                             float* dt_ms) {
    using sparta::fail;
    if (!A || !Bp || !C) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_prepared: NULL argument");
    if (Bp->cols != A->cols || Bp->dtype != A->dtype || Bp->device != A->device)
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_prepared: this B was prepared for another handle shape / type / device");
    SPARTA_GUARD_BEGIN
    // (cleared on every way out, exceptions included: a stale pointer would make a later plain product gather from this B)
    struct Prepared { sparta_vbs_t* a; ~Prepared() { a->prepared_brm = nullptr; a->prepared_ld = 0; } } guard{A};
    A->prepared_brm = Bp->d_Brm; A->prepared_ld = Bp->ld_brm;
    return spmm_impl(A, Bp->B, Bp->ldb, SPARTA_COL_MAJOR, Bp->shard_rows, Bp->shard_stride, Bp->n_cols, C, ldc, c_layout, accumulate,
                     SPARTA_PTR_DEVICE, stream, SPARTA_SPMM_MFMA, dt_ms);
    SPARTA_GUARD_END("sparta_vbs_spmm_prepared")
}

int sparta_b_destroy(sparta_b_t* Bp) {
    if (!Bp) return SPARTA_OK;
    DeviceGuard guard(Bp->device);
    if (Bp->d_Brm) (void)hipFree(Bp->d_Brm);
    delete Bp;
    return SPARTA_OK;
}

int sparta_vbs_set_class_timing(sparta_vbs_t* A, int32_t enable) {
    using sparta::fail;
    if (!A) return fail(SPARTA_ERR_INVALID, "sparta_vbs_set_class_timing: NULL handle");
    DeviceGuard guard(A->device);
    if (enable && !A->cev[0][0]) {
        for (int c = 0; c < 4; c++)
            for (int e = 0; e < 2; e++) HIP_TRY(hipEventCreate(&A->cev[c][e]));
    }
    if (enable && !A->d_clk) {
        HIP_TRY(hipMalloc((void**)&A->d_clk, (16 + 4 * 64 * 8 + 2048) * sizeof(long long)));
        HIP_TRY(hipMemset(A->d_clk, 0, (16 + 4 * 64 * 8 + 2048) * sizeof(long long)));
    }
    A->class_timing = enable != 0;
    return SPARTA_OK;
}

int sparta_vbs_class_times(sparta_vbs_t* A, float* ms_out) {
    using sparta::fail;
    if (!A || !ms_out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_class_times: NULL argument");
    DeviceGuard guard(A->device);
    for (int c = 0; c < 4; c++) {
        ms_out[c] = 0.0f;
        if (!A->class_timing || !A->class_ran[c]) continue;
        HIP_TRY(hipEventSynchronize(A->cev[c][1]));
        HIP_TRY(hipEventElapsedTime(&ms_out[c], A->cev[c][0], A->cev[c][1]));
    }
    return SPARTA_OK;
}

int sparta_pack_blocks(const void* src, int64_t block_bytes, const int32_t* ids_dev, int64_t n_blocks, void* dst, void* stream) {
    using sparta::fail;
    if (n_blocks < 0 || block_bytes <= 0 || block_bytes % 16 != 0)
        return fail(SPARTA_ERR_INVALID, "sparta_pack_blocks: block_bytes must be a positive multiple of 16, n_blocks >= 0");
    if (n_blocks == 0) return SPARTA_OK;
    if (!src || !ids_dev || !dst) return fail(SPARTA_ERR_INVALID, "sparta_pack_blocks: NULL argument");
    if (((uintptr_t)src | (uintptr_t)dst) % 16 != 0) return fail(SPARTA_ERR_INVALID, "sparta_pack_blocks: src and dst must be 16-byte aligned");
    if (n_blocks > INT32_MAX) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_pack_blocks: too many blocks for one launch");
    const int64_t vec = block_bytes / 16;
    const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8, vec / (4 * kThreads)));
    launch_pack_blocks(dim3((unsigned)n_blocks, gy), (hipStream_t)stream, src, ids_dev, dst, vec);
    HIP_TRY(hipGetLastError());
    return SPARTA_OK;
}

}  // extern "C"
