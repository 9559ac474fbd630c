// vbs_device.hpp -- declarations shared by the translation units of the device side of libsparta_amd.so:
//   k_f32_class.hip   per-class fp32 MFMA kernels + the exact-order parity kernel
//   k_f32_stream.hip  persistent fp32 stream kernels, fix-up, B-tail copy, zero fill
//   k_h16.hip         fp16 / bf16 storage: LDS-staged and direct stream kernels, conversions
//   k_sparse.hip      sparse-row kernels, layout transposes, row-block pack
//   vbs_plan.cpp      host: stream plans (step lists, worker ranges, split tiles)
//   vbs_capi.cpp      host: device image (sparta_vbs), sparta_vbs_create* / sparta_vbs_spmm* (include/sparta_amd.h)
// Every kernel TU exports plain launch functions (namespace sparta_dev); the host TUs never see a __global__ symbol.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <memory>
#include <vector>

#include "host_core.hpp"

namespace sparta_dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned: global_load_dwordx4 on any float address
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kTN = 128;   // columns of C per workgroup
constexpr int kKP = 64;    // k-depth of one panel step

// one row tile of one block-row
struct TileDesc {
    int64_t a_off;     // element offset into A of (first block of the block-row) + r0
    int64_t jab_off;   // offset into jab of the block-row's first block-column id
    int32_t nb;        // nonzero blocks in the block-row
    int32_t h;         // block-row height = leading dimension of each of its blocks
    int32_t c_row;     // first row of C written by this tile
    int32_t mt_flags;  // low 16 bits: rows in this tile (<= class height); TILE_* flags above
};
constexpr int32_t TILE_TAIL = 1 << 16;   // the block-row's last block lies in the zero-padded last block column (cols % w != 0)
static_assert(sizeof(TileDesc) == 32, "TileDesc must stay 32 bytes");

struct SpmmParams {
    const TileDesc* tiles;
    const int32_t* jab;
    const float* A;
    const float* B;
    float* C;
    int64_t ldb, ldc;
    int64_t cols;      // valid rows of B
    int32_t n_tiles, n_ntiles;
    int32_t N, w;
    int32_t b_row_major, c_row_major;
    int32_t accumulate, vec_ok;
    int64_t shard_stride; // elements between consecutive slabs of a gathered B
    int64_t shard_rows;   // 0: B is one matrix; >0: B is an all-gather result of column-major shard_rows x N slabs
    long long* clk;       // clock probe (NULL = off): workgroup 0 writes {s_memtime, s_memrealtime} at entry and exit
};

constexpr int SK_KP = 32;                 // k depth of a step
constexpr int SK_TM = 64;                 // max rows of a tile
constexpr int32_t STEP_FIRST = 1 << 16;   // first step of a (segment of a) tile: accumulators start at zero
constexpr int32_t STEP_LAST = 1 << 17;    // last step of a (segment of a) tile: run the epilogue
constexpr int32_t STEP_SPLIT = 1 << 18;   // the tile is shared with another worker: epilogue goes to the workspace
constexpr int32_t STEP_LO_ABSENT = 1 << 27;  // 16-bit pair tiles (two vertically adjacent 32-row block-rows walked as ONE 64-row tile over the union of their block
constexpr int32_t STEP_HI_ABSENT = 1 << 28;  // columns, vbs_plan.cpp): the lower / upper block-row has no block in this step's column -- that half of the slice is zeros and is not fetched
constexpr int32_t STEP_KPAIRS_SHIFT = 24;  // bits 24..26: (MFMA pairs this step needs) - 1, fp32 one-tile plans with a fragment image (k-compaction: the
                                           // non-empty columns of the step's slice of A come first, vbs_plan.cpp); read by vbs_spmm_f32_direct_kernel only
constexpr int64_t kAFragSlice = 1040;      // floats per step of the fragment image: 16 (the step's LDS position table: 32 bytes + padding) + 1024 (fragments)
constexpr int32_t STEP_TAIL = 1 << 19;    // panel of the zero-padded last block column: read from StreamParams::B_tail, b_row = k offset in it
constexpr int SK_SLOT_FLOATS = 32 * kThreads;   // one partial accumulator image: 32 registers x 256 threads

struct StepRec {                          // 32 bytes, one per (block, 32-deep k slice), in execution order
    int64_t a_off;                        // element offset into A of (tile row 0, first k of this step)
    int32_t b_row;                        // first row of B of this step's panel (jb * w + ks)
    int32_t h;                            // leading dimension of the A block (block-row height)
    int32_t c_row;                        // first row of C of the tile
    int32_t mt_flags;                     // rows of the tile (low 16 bits) | STEP_* flags
    int32_t slot;                         // workspace slot for STEP_LAST|STEP_SPLIT, else -1
    int32_t pad;                          // gathered-B step lists: index of the slab that holds b_row (b_row is then slab-local); else 0
};
static_assert(sizeof(StepRec) == 32, "StepRec must stay 32 bytes");

struct FixRec {                           // one per split tile
    int32_t c_row, mt;
    int32_t slot_begin, n_slots;          // its partial images: fix_slots[slot_begin .. slot_begin + n_slots)
};

struct StreamParams {
    const StepRec* steps;
    const int32_t* worker_range;          // [2 * P]: begin, end step of every worker
    const float* A;
    const float* B;
    const float* B_tail;                  // zero-padded copy of B's last (partial) block row: w x N, ld = w (col-major) / N (row-major)
    float* C;
    float* ws;                            // partial images: [n_ntiles][n_slots][SK_SLOT_FLOATS], one per segment of a split tile
    int64_t ldb, ldc, cols;
    int64_t shard_rows, shard_stride;
    int64_t ws_slab_stride;               // floats between the workspaces of consecutive 128-column slabs
    int32_t accumulate, c_row_major;
    int32_t N, w;
    long long* clk;                       // clock probe, see clock_probe()
    int32_t c_nt;                         // 1: non-temporal C stores (long tiles), 0: default cache policy (short tiles); see vbs_kernel_common.hpp
    int32_t stagger;                      // developer knob (SPARTA_STAGGER): workgroups of the second half of the grid start this many x 64 cycles late
    int32_t sub_ranges = 0;               // 1: worker_range holds two adjacent sub-worker ranges per workgroup (16-bit `wide16` plans)
};

// ---- the GEMM-shaped hub kernel of 16-bit handles (k_hub16.hip): group tiles of G = 2 or 4 sub-tiles of <= 64 rows (block-rows of the fixed 64 x 64 grid that
// share most of their block columns -- not necessarily neighbours), walked over the UNION of their block columns; a workgroup owns the group's rows x one
// 256-column slab of C and stages A and B once through LDS for all its waves
constexpr int kHubGMax = 4;               // most 64-row sub-tiles per group tile (the kernel has a two- and a four-sub-tile form)
struct HubStep {                          // 32 bytes, one per (group tile, KP-deep k slice), in execution order
    uint32_t a_lo, a_hi;                  // element offset into the 16-bit image of A of this step's first PRESENT slice (present slices back to back)
    int32_t b_row;                        // first row of B of the step's panel (row inside its slab for a gathered B; k offset into B_tail for STEP_TAIL)
    int32_t shard;                        // gathered B: index of the slab that holds b_row; else 0
    int32_t flags;                        // bits 0..3: which sub-tiles have a block in this block column; STEP_LAST / STEP_SPLIT / STEP_TAIL
    int32_t slot;                         // STEP_LAST | STEP_SPLIT: first of the segment's G consecutive workspace images (one per sub-tile), else -1
    int32_t tile;                         // index into HubTile
    int32_t pad;
};
static_assert(sizeof(HubStep) == 32, "HubStep must stay 32 bytes");
struct HubTile { int32_t c_row[4]; int32_t mt[4]; };      // per sub-tile: first row of C, rows (<= 64; 0: no such sub-tile)
struct HubParams {
    const HubStep* steps;
    const int32_t* worker_range;          // [2 * n_workers]: begin, end step of every worker
    const HubTile* tiles;
    const uint16_t* A;                    // slices of 64 rows x KP, [row][16-byte chunk c ^ swizzle(row)] (the LDS image, see k_hub16.hip)
    const uint16_t* B;
    const uint16_t* B_tail;               // zero-padded copy of B's last (partial) block row: w x N, ld = w; or nullptr
    float* C;
    float* ws;                            // partial images, as StreamParams::ws
    int64_t ldb, ldc;
    int64_t shard_stride;
    int64_t ws_slab_stride;               // floats between the workspaces of consecutive 128-column slabs
    int32_t accumulate, c_row_major, c_nt, w;
    int32_t n_slabs;                      // 256-column slabs of this launch: grid = n_workers x n_slabs workgroups (the slabs of a worker adjacent on one XCD)
    int32_t n_workers;
    int32_t n_cols;                       // columns of B / C (a multiple of 128): the last slab may be a half slab
    int32_t pad0;
};

constexpr int kFixGroup = 16;   // partial images per group of the fix-up group stage (k_f32_stream.hip)

struct SparseParams {
    const int64_t* rowptr;     // [n_rows + 1] into col / val
    const int32_t* col;
    const float* val;
    const int32_t* crow;       // C row of every sparse row
    const int32_t* list;       // the rows this launch handles (ordinals)
    int32_t n_list;
    const void* B;             // row-major, ld = ldb elements; fp32 (BK = 0), fp16 (1) or bf16 (2)
    int64_t b_col_stride;      // 0: row-major B as above.  > 0: B is COLUMN-major (element (k, n) at slab(k) + k % shard_rows + n * b_col_stride),
    int64_t shard_rows, shard_stride;   //      read in place, one 4-byte gather per element: only worth it for a handful of sparse rows
    int64_t ldb;
    float* out;                // out_is_c 1: row-major C (ld = ldc, row = crow); 2: COLUMN-major C (ld = ldc)
    int64_t ldo;
    int32_t out_is_c, accumulate, N;
    int32_t scalar_gather;     // 1: (column, value) pairs through scalar loads, row base in SGPRs (k_sparse.hip: sparse_row_partial_s); 0: the round-3 gather (SPARTA_SP_SCALAR=0)
};

// resident-column product (k_colres.hip): A as slots (rows, the long ones cut into chunks) sorted by length, 64 to a slice, entry k of a slice's slots contiguous
constexpr int kColresWaves = 16;        // waves of a workgroup of k_colres.hip (host layout and kernel agree on it)
struct ColresLong { int32_t row, first, n, pad; };   // a row cut into chunks: its cell in the staging image, the first of its extra cells, how many
// A matrix whose columns of B or of C do not fit LDS whole is cut: the rows of C into PARTS (a workgroup owns NC columns of one part: grid y), the columns of A into K RANGES (a workgroup
// walks them one after the other, its sums staying in registers: the rows of B of a range in LDS at a time).
constexpr int kColresMaxParts = 4, kColresMaxRanges = 4;
constexpr int kColresCellsPerPlane = 160 * 1024 / 4;      // floats of LDS: the most cells a column's staging image can have
struct ColresPartDev {
    int32_t r0, rows;          // the part's rows of C: [r0, r0 + rows), r0 a multiple of 4
    int32_t n_slices, n_long;
    int32_t plane;             // cells per column of the staging image (rows + extra cells, a multiple of 4)
    int32_t meta;              // offset (int32) of the part's block in `meta`: wslice[17], woff[n_ranges][17], bnd[n_ranges][n_slices]
    int32_t dest;              // offset (int32) of its dest[64 n_slices] in `dest`
    int32_t longs;             // offset (records) of its list in `longs`
    int32_t all_store;         // 1: every row of the part is stored by this kernel (no row of tiles, no mixed row: `mode` is not read)
    int32_t pad;
};
struct ColresParams {
    // Per part and K range: the slices of wave w (w, w + 16, ... of the part's length-sorted slots; per range a multiple of 4 steps wide, at least 4) back to back, in batches of 4 steps:
    // batch t of wave w, lane l at (woff[range][w] + t) * 64 + l.  Columns are relative to the range's first; a slot shorter than its slice (in that range) ends in (the range's length, 0.0f):
    // the cell behind the range's last row of B in LDS, which the kernel clears.
    const uint2* col4;         // four 16-bit columns per batch and lane
    const float4* val4;        // four values per batch and lane; nullptr: every stored value is 1.0f (unit image)
    const ColresPartDev* parts;
    const int32_t* meta;       // per part: wslice[17] (first slice of every wave in the wave-major order of bnd / dest), woff[range][17] (batches), bnd[range][slice] (the batch of
                               // its wave's stream of that range behind the slice's last one)
    const int32_t* dest;       // per part [64 n_slices] (wave-major): cell of the staging image the slot's sum goes to (row of C - r0, or an extra cell >= rows); -1: padding slot
    const ColresLong* longs;
    const uint8_t* mode;       // per row of C (padded to a multiple of 4): 0 the row is not this kernel's (a block-row of tiles), 1 store its sum, 2 add it to what the tile launches stored
    const float* B;            // column-major, ld = ldb
    int64_t ldb;
    float* C;                  // column-major, ld = ldc
    int64_t ldc;
    int32_t krange[kColresMaxRanges + 1];      // first column of every K range (multiples of 4), then cols
    int32_t n_parts, n_ranges, N, accumulate, vec_out, vec_in;
    int32_t n_cus, share, stagger_ticks;      // CUs of the device, groups the CUs of the first dispatch round start in, and the offset between the groups in 10 ns ticks (0: none) -- see the kernel
    int32_t probe;             // developer probe (SPARTA_COLRES_PROBE, wrong products): 1 skip the loads of B, 2 skip the stream of A, 4 skip the stores of C
};

// ---- column-compacted ("union-pattern") tiles of fp32 handles (k_union.hip, vbs_union.cpp; host form: sparta::UnionPlanHost) ----
struct UnionRec { int32_t c_row, info, tail_off, pad; };  // per step: first row of C of its tile; info = rows of the tile (bits 0..6) | valid list positions of this step, 0..32 (bits 8..13) | UREC_LAST
                                                         // | tail entries per row (bits 17..21); tail_off: first (column, value) pair of the tile's tail
static_assert(sizeof(UnionRec) == 16, "UnionRec must stay 16 bytes");
constexpr int32_t UREC_LAST = 1 << 16;                   // the tile's last step: add the tail, store the tile's rows of C
constexpr int UREC_TAIL_SHIFT = 17;
constexpr int kUnionPadSteps = 4;                        // records / list entries / slices behind the last step (the pipeline requests up to three steps past a worker's range)
constexpr int kUnionPairFloats = 256;    // fp32 plans: floats (1 KB) behind a step's slice of A that hold the (column, value) pairs of the tail entry the step requests
constexpr int kUnionTypes = 4;    // tile types of one handle.  fp32: type t = tiles of 16 t + 1 .. 16 (t + 1) rows (t + 1 MFMA row tiles of 16); 16-bit: types 0, 1 = tiles of <= 32 / 33..64 rows
struct UnionSide {                // one tile type
    const UnionRec* rec;          // per step, in execution order (worker after worker)
    const int32_t* ids;           // [step][32]: the rows of B (= columns of A) of the step's list positions; behind the valid ones the tile's first column (fetched: A holds zeros there)
    const void* A;                // the steps' slices as the LDS image the kernel wants.  fp32, type t (R = 16 (t + 1) rows): [step]{[rt][h][lane][4] floats, then kUnionPairFloats: the step's tail pairs [row] uint2}, lane = 16 kq + i:
                                  // A[16 rt + i][k = 4 (4 h + e) + kq] (one ds_read_b128 per lane = the row's values of four consecutive 16x16x4 MFMAs);
                                  // 16-bit, type t (32 (t + 1) rows): [step][rt][m][kg][row][8] = A[32 rt + row][k = 16 m + 8 kg + e]
    const int32_t* worker_range;  // [2 x workers]: begin, end step
    const uint2* tail;            // per tile (execution order) [entry][R rows]: (column, value bits) added in the tile's epilogue
    int32_t n_workers, c_nt;
};
struct UnionParams {              // ONE launch: workgroup b walks its range of every type (side[t].worker_range[2 b ..]), tallest type first
    UnionSide side[kUnionTypes];
    const float* B;               // ROW-major cols x n_cols, ld = ldb (a multiple of 4 elements, 16-byte aligned base)
    int64_t ldb;
    float* C;
    int64_t ldc;
    int32_t n_cols, accumulate, c_row_major, pad;
};

struct SpSegRec { int64_t p0; int32_t cnt, pad; };
struct SpLongRec { int32_t ord, seg_begin, n_seg, pad; };

struct BlockRowDesc {
    int64_t a_off, jab_off;
    int32_t nb, h, c_row, pad;
};

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return sparta::fail(SPARTA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

}  // namespace sparta_dev

struct sparta_vbs {
    int device = 0, dtype = SPARTA_F32;
    int64_t rows = 0, cols = 0, block_rows = 0, w = 0, nblocks = 0, nztot = 0;
    float* d_A = nullptr;                    // fp32: the reference's mab; 16-bit handles: packed slices (see sparta_vbs_create)
    int kp16 = 0;                            // 16-bit handles: k depth of a step (32 or 64)
    int32_t* d_jab = nullptr;
    sparta_dev::TileDesc* d_tiles[4] = {nullptr, nullptr, nullptr, nullptr};   // classes 16, 32, 64, 128
    int64_t n_tiles[4] = {0, 0, 0, 0};       // launch entries (real tiles + padding)
    int64_t n_real_tiles[4] = {0, 0, 0, 0};
    sparta_dev::BlockRowDesc* d_brows = nullptr;
    int64_t n_brows = 0;
    int64_t exec_area = 0;
    int64_t a_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t cev[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    bool class_timing = false;
    bool class_ran[4] = {false, false, false, false};
    // stream plan (w % 32 == 0): see vbs_spmm_f32_stream_kernel
    sparta_dev::StepRec* d_steps[2] = {nullptr, nullptr};      // per tile type: [0] <= 32 rows, [1] 33..64 rows
    std::vector<sparta_dev::StepRec> h_steps[2];               // host copies (padded), source of the gathered-B variants
    sparta_dev::StepRec* d_steps_g[2] = {nullptr, nullptr};    // step lists for sparta_vbs_spmm_gathered with shard_rows == g_shard_rows
    int64_t g_shard_rows = 0;
    int32_t* d_wrange[2] = {nullptr, nullptr};
    sparta_dev::FixRec* d_fix = nullptr;
    std::vector<std::pair<int64_t, int64_t>> zero_ranges;   // long runs of rows without blocks (local C rows), see vbs_zero_rows_kernel
    int32_t* d_fix_slots = nullptr;
    int64_t n_steps[2] = {0, 0};
    int32_t n_workers = 0, n_fix = 0, n_split = 0, n_slots = 0;
    int32_t max_tile_slots = 0;            // most partial images of one split tile
    int32_t* d_big_fix = nullptr;          // fix records with more than 2 * kFixGroup images (group stage)
    int32_t n_big_fix = 0;
    void* d_ws = nullptr;
    size_t d_ws_bytes = 0;
    int64_t n_plan_tiles[2] = {0, 0};             // see StreamPlanHost
    bool tiles_row_aligned[2] = {true, true};
    bool wide16 = false;                      // the 16-bit one-tile plan holds two sub-worker ranges per workgroup (vbs_spmm_h16_direct_kernel, WC = 64)
    float* d_a_frag = nullptr;                    // A of the one-tile plan in fragment order (k_f32_direct.hip) or nullptr
    bool legacy_dropped = false;                  // fp32: the reference-layout image d_A was freed once the fragment image had won (rebuilt on demand, then kept)
    // hub plan (16-bit handles of 64-wide blocks; vbs_plan.cpp, k_hub16.hip)
    sparta_dev::HubStep* d_hub_steps = nullptr;
    sparta_dev::HubStep* d_hub_steps_g = nullptr; // step list for sparta_vbs_spmm_gathered with shard_rows == g_shard_rows
    std::vector<sparta_dev::HubStep> h_hub_steps; // host copy (padded), source of the gathered variant
    sparta_dev::HubTile* d_hub_tiles = nullptr;
    int32_t* d_hub_wrange = nullptr;
    uint16_t* d_hub_A = nullptr;
    int hub_g = 0, hub_workers = 0;
    int64_t n_hub_steps = 0, hub_area = 0, hub_union_area = 0, n_hub_tiles = 0, n_hub_groups = 0, hub_chunks = 0, hub_segments = 0;
    bool has_tail = false;                 // cols % w != 0: the stream path needs B_tail
    void* d_btail = nullptr;
    size_t d_btail_bytes = 0;
    int last_path = 0;                     // 1: stream kernel, 2: per-class branch-free kernels, 3: per-class generic kernels
    std::vector<std::pair<int64_t, int>> tuned;   // (n_cols/layout key) -> measured best path
    float tune_ms[2] = {0.0f, 0.0f};
    void* d_tune = nullptr;
    size_t d_tune_bytes = 0;
    void* d_B16 = nullptr;                 // 16-bit handles, host-pointer calls: B converted on the device
    size_t d_B16_bytes = 0;
    void* d_Bt = nullptr;                  // 16-bit handles, n_cols % 128 != 0: the last n_cols % 128 columns of B, zero-padded to a 128-column slab
    size_t d_Bt_bytes = 0;
    void* d_Ct = nullptr;                  // ... and the 128-column slab of C they produce (column-major, ld = rows), merged into C afterwards
    size_t d_Ct_bytes = 0;
    long long* d_clk = nullptr;           // clock probe: [4 launches][4] = {s_memtime, s_memrealtime} at entry, at exit
    hipEvent_t tev0 = nullptr, tev1 = nullptr;
    // sparse-row path (fp32 handles): the block-rows taken out of the MFMA plans, as rows of (column, value)
    int64_t n_sp_rows = 0, n_sp_short = 0, n_sp_long = 0, sp_nnz = 0;
    bool ext_sparse = false;               // created from CSR: the sparse rows have no dense image (no exact-order kernel for them)
    int64_t* d_sp_rowptr = nullptr;
    int32_t* d_sp_col = nullptr;
    float* d_sp_val = nullptr;
    int32_t* d_sp_crow = nullptr;
    int32_t* d_sp_list = nullptr;          // the short rows (one wave each)
    void* d_sp_segs = nullptr;             // SpSegRec[n_sp_segs]: segments of the long rows
    void* d_sp_long = nullptr;             // SpLongRec[n_sp_long]
    int64_t n_sp_segs = 0;
    int32_t* d_sp_stream_begin = nullptr;  // XCD-affine order of the segments: eight streams, [9] offsets into d_sp_segs (nullptr: one list, window-major)
    int64_t sp_max_stream = 0;
    void* d_sp_part = nullptr;             // partial rows of the segments
    size_t d_sp_part_bytes = 0;
    // resident-column product (k_colres.hip): fp32 handles whose rows are ALL sparse rows and whose columns of B fit LDS
    void* d_cr_col = nullptr;               // 16-bit columns
    void* d_cr_val = nullptr;               // values (nullptr: unit image)
    int32_t* d_cr_meta = nullptr;          // woff[17], wslice[17], bnd[n_slices]
    int32_t* d_cr_dest = nullptr;
    void* d_cr_longs = nullptr;
    int32_t cr_slices = 0, cr_long = 0, cr_plane = 0, cr_lmax = 0;       // slices of the part with most; rows cut into chunks; cells of the largest staging image / range of B; longest slot
    int32_t cr_parts = 0, cr_ranges = 0, cr_span = 0;
    int32_t cr_krange[sparta_dev::kColresMaxRanges + 1] = {0, 0, 0, 0, 0};
    void* d_cr_parts = nullptr;
    void* d_cr_mode = nullptr;
    bool cr_unit = false;                  // every value 1.0f: no value array
    // ... and the same matrix once more in FOUR parts of the rows of C, for products of few column sets (N <= 128 with two columns per workgroup): 4 x the workgroups, each with a
    // quarter of the stream of A -- with fewer workgroups than CUs one workgroup's stream IS the product's time
    struct ColresSmall {
        void* col = nullptr; void* val = nullptr; int32_t* meta = nullptr; int32_t* dest = nullptr; void* longs = nullptr; void* parts = nullptr;
        int32_t slices = 0, plane = 0, n_parts = 0;
    } cr_small;
    bool last_colres_small = false;        // the last product on this path used the four-part image
    int64_t cr_entries = 0;                // stored entries, padding included
    int last_colres_nc = 0;                // columns per workgroup of the last product on this path (0: the product took another path)
    // column-compacted tiles (fp32 handles made from a CSR: vbs_build.cpp mode 3): per tile type [0] <= 32 rows, [1] 33..64 rows
    sparta_dev::UnionRec* d_u_rec[sparta_dev::kUnionTypes] = {nullptr, nullptr, nullptr, nullptr};
    int32_t* d_u_ids[sparta_dev::kUnionTypes] = {nullptr, nullptr, nullptr, nullptr};
    void* d_u_a[sparta_dev::kUnionTypes] = {nullptr, nullptr, nullptr, nullptr};
    int32_t* d_u_wrange[sparta_dev::kUnionTypes] = {nullptr, nullptr, nullptr, nullptr};
    void* d_u_tail[sparta_dev::kUnionTypes] = {nullptr, nullptr, nullptr, nullptr};
    int32_t u_workers[sparta_dev::kUnionTypes] = {0, 0, 0, 0};
    int64_t u_steps[sparta_dev::kUnionTypes] = {0, 0, 0, 0}, u_tiles[sparta_dev::kUnionTypes] = {0, 0, 0, 0};   // per tile type of the DEVICE plan
    int64_t u_tiles_h[2] = {0, 0}, u_steps_h[2] = {0, 0}, u_steps_total = 0;   // tiles / steps of tiles of <= 32 / 33..64 rows; all steps
    int64_t u_area = 0, u_cols = 0, u_nnz = 0;      // stored elements (rows x list entries), list entries, nonzeros held (lists + tails)
    int64_t u_tail_nnz = 0, u_rows = 0;             // nonzeros in the tiles' tails; rows of C the tiles own
    int64_t u_exec_area = 0;                        // elements the kernel multiplies: steps x 32 x rows of the tile's type
    const void* brm_ready = nullptr;       // the row-major B of the product in flight (set by the first launch that needs it, cleared when the product returns)
    int64_t brm_ld = 0;
    void* d_Brm = nullptr;                 // row-major copy of a column-major / gathered B
    size_t d_Brm_bytes = 0;
    const void* prepared_brm = nullptr;    // set for the duration of a sparta_vbs_spmm_prepared call: the caller's row-major copy, made once
    int64_t prepared_ld = 0;               // ... and its row stride (the n_cols it was prepared for: a 16-bit call with n_cols % 128 != 0 is cut into sub-calls)
    void* d_B = nullptr;
    size_t d_B_bytes = 0;
    void* d_C = nullptr;
    size_t d_C_bytes = 0;
};

namespace sparta_dev {

// ---- launch functions exported by the kernel translation units ------------------------------------------------------
// k_f32_class.hip
void launch_f32_class(int cls, bool b_row_major, bool generic, const SpmmParams& p, hipStream_t st);
void launch_f32_exact(unsigned n_brows, hipStream_t st, const BlockRowDesc* rows, const int32_t* jab, const float* A, const float* B, float* C,
                      int64_t ldb, int64_t ldc, int64_t cols, int N, int w, int b_row_major, int c_row_major, int accumulate, int64_t shard_rows,
                      int64_t shard_stride);
// k_f32_stream.hip
void launch_f32_stream(bool mi2, bool b_row_major, bool gathered, dim3 grid, hipStream_t st, const StreamParams& sp);
void launch_f32_direct(bool c_stage, dim3 grid, hipStream_t st, const StreamParams& sp);
void launch_f32_legacy_from_frag(hipStream_t st, const StepRec* steps, int64_t n_steps, const float* a_frag, float* A);
void launch_fixup_group(dim3 grid, hipStream_t st, const FixRec* fix, const int32_t* big, const int32_t* fix_slots, float* ws_all, int64_t ws_slab_stride);
void launch_fixup(dim3 grid, hipStream_t st, const FixRec* fix, const int32_t* fix_slots, const float* ws_all, int64_t ws_slab_stride, float* C, int64_t ldc,
                  int c_row_major, int accumulate);
void launch_tail_copy(hipStream_t st, const float* B, int64_t ldb, int b_row_major, int64_t row0, int64_t cols, int w, int N, float* B_tail);
void launch_zero_rows(dim3 grid, hipStream_t st, float* C, int64_t ldc, int c_row_major, int64_t row0, int64_t nrows, int N);
// k_h16.hip
void launch_h16_stream(int kp, bool mi2, bool bf16, bool gathered, bool c_stage, bool wide, dim3 grid, hipStream_t st, const StreamParams& sp);
void launch_h16_slab256(bool bf16, dim3 grid, hipStream_t st, const StreamParams& sp);   // one-tile plans of 32-wide blocks, 256-column slabs (grid.y = N / 256), no split tile
bool h16_uses_direct_kernel(int kp, bool mi2);
void launch_h16_quad(int kp, bool bf16, bool gathered, dim3 grid, hipStream_t st, const StreamParams& sp);
// k_hub16.hip
void launch_h16_hub(int variant, bool bf16, bool gathered, hipStream_t st, const HubParams& p);
int hub_variant_kp(int variant);
int hub_variant_g(int variant);
void launch_tail_copy_h16(hipStream_t st, const uint16_t* B, int64_t ldb, int64_t row0, int64_t cols, int w, int N, uint16_t* B_tail);
void launch_convert_h16(bool bf16, hipStream_t st, const float* src, int64_t ld_in, int64_t rows, int64_t n_cols, uint16_t* dst, int64_t ld_out);
// C[:, col0 + j] (+)= Ct[:, j] for j < n_t: the column tail of a 16-bit product (Ct column-major, ld = rows)
void launch_col_tail_merge(hipStream_t st, const float* Ct, int64_t rows, float* C, int64_t ldc, int c_row_major, int col0, int n_t, int accumulate);
// k_sparse.hip
void launch_sparse_kernels(int vec, int bk, const SparseParams& q, unsigned gy, hipStream_t st, const int32_t* list, int64_t n_short, const SpSegRec* segs,
                           int64_t n_segs, const SpLongRec* longs, int64_t n_long, float* part, const int32_t* stream_begin = nullptr, int64_t max_stream = 0);
void launch_b_to_row_major(bool is16, unsigned grid, hipStream_t st, const void* B, int64_t ldb, int64_t shard_rows, int64_t shard_stride, int64_t rows, int N,
                           void* out, int64_t ld_out);
// k_union.hip
void launch_union_f32(unsigned n_slabs, hipStream_t st, const UnionParams& p);
void launch_union_h16(bool bf16, unsigned n_slabs, hipStream_t st, const UnionParams& p);   // 16-bit handles: p.B = the row-major 16-bit B, side.A = 16-bit slices
// k_colres.hip
int launch_colres(int nc, const ColresParams& p, size_t lds_bytes, hipStream_t st);     // nc = 1..4 columns per workgroup; 0 or a hipError_t
int colres_max_slices(int nc);                                                          // slices the nc-column kernel holds sums for
void launch_pack_blocks(dim3 grid, hipStream_t st, const void* src, const int32_t* ids, void* dst, int64_t block_vec);

// vbs_plan.cpp
struct StreamPlanIn {
    int64_t cols, w, br0, br1, jab_lo, mab_lo;
    const int64_t* row_part; const int64_t* nzcount; const int64_t* jab; const float* mab;
    int32_t dtype, device;
    const uint8_t* skip;                      // [br1 - br0] block-rows handled by the sparse-row path (no tiles), or nullptr
};
struct StreamPlanHost {
    std::vector<StepRec> steps[2];            // per tile type: [0] <= 32 rows, [1] 33..64 rows
    std::vector<int32_t> wrange[2];           // [2 * n_workers] begin / end step of every worker
    std::vector<FixRec> fix;                  // split tiles + tiles of block-rows without blocks (zero fill)
    std::vector<std::pair<int64_t, int64_t>> zero_ranges;   // (first row, rows) of long block-rows without blocks: vbs_zero_rows_kernel instead of fix-up tiles
    std::vector<int32_t> fix_slots;
    std::vector<uint16_t> a16_steps[2];       // 16-bit handles: A as TM x kp slices ([k chunk of 8][row][8]), one per step, in STEP order; device image = [0] then [1]
    int n_workers = 0, n_split = 0;
    int plan_aligned[2] = {0, 0};
    bool wide16 = false;                      // 16-bit one-tile plan with two sub-workers per workgroup (2 x n_workers ranges in wrange[0])
    bool window_plan = false;                 // 64-row tiles dealt as (window, tile) pieces (SPARTA_TILE_WINDOW_COLS, experiment)
    int64_t kp = SK_KP;
    int64_t n_plan_tiles[2] = {0, 0};         // tiles (with at least one block) per plan: steps per tile decides the cache policy of the C stores
    bool tiles_row_aligned[2] = {true, true}; // every tile of the plan starts at a multiple of 32 rows of C (whole 128-byte lines of a column-major C)
    std::vector<float> a_frag;                // fp32 one-tile plan: A per step in MFMA fragment order (vbs_spmm_f32_direct_kernel), or empty
    // hub plan (16-bit handles of 64-wide blocks): the long tiles of 33..64 rows, grouped by the similarity of their block columns into group tiles for
    // vbs_spmm_h16_hub_kernel (k_hub16.hip); their block-rows are in none of the two plans above
    std::vector<HubStep> hub_steps;           // in execution order, padded by 32 harmless copies
    std::vector<HubTile> hub_tiles;
    std::vector<int32_t> hub_wrange;          // [2 * hub_workers]
    std::unique_ptr<uint16_t[]> hub_a16;      // slices of 64 rows x 64 k in the kernel's LDS image, present sub-tiles of a step back to back, steps in execution order
    size_t hub_a16_elems = 0;                 // (uninitialised storage: 10^10 elements on a dense hub part -- the packing loop writes every element, zeros included)
    int hub_g = 0;                            // sub-tiles per group tile (2 or 4); 0: no hub plan
    int hub_workers = 0;
    int64_t n_hub_steps = 0, hub_area = 0, hub_union_area = 0;    // steps; stored elements of the hub tiles; elements the kernel multiplies (absent sub-tiles included)
    int64_t n_hub_tiles = 0, n_hub_groups = 0, hub_chunks = 0, hub_segments = 0;    // K chunks of the step order; segments (runs of one group on one worker)
};
// vbs_union.cpp: the device form of the column-compacted tiles -- tiles dealt to workers longest first, a worker's steps back to back
struct UnionDevPlan {
    std::vector<UnionRec> rec[kUnionTypes];
    std::vector<int32_t> ids[kUnionTypes];
    std::vector<float> a[kUnionTypes];        // fp32 handles: slices [step][R x 32 + kUnionPairFloats] floats (R = 16 (type + 1)), the layout of UnionSide::A
    std::vector<uint16_t> a16[kUnionTypes];   // 16-bit handles: slices [step][R x 32] 16-bit elements (R = 32 (type + 1)), rounded to the storage type
    std::vector<int32_t> wrange[kUnionTypes];
    std::vector<uint32_t> tail[kUnionTypes];  // (column, value bits) pairs, tile after tile in execution order
    int32_t n_workers[kUnionTypes] = {0, 0, 0, 0};
    int64_t n_steps[kUnionTypes] = {0, 0, 0, 0}, n_tiles[kUnionTypes] = {0, 0, 0, 0};
    int32_t type_rows[kUnionTypes] = {0, 0, 0, 0};   // R of each type
    int64_t tiles_by_height[2] = {0, 0}, steps_by_height[2] = {0, 0};   // tiles / steps of tiles of <= 32 rows, of 33..64 rows (what sparta_vbs_union_info reports)
    int64_t area = 0, cols = 0, rows = 0;     // stored elements (tile rows x list entries); list entries; rows of C the tiles own
};
int build_union_plan(const sparta::UnionPlanHost& U, int max_workers, UnionDevPlan& P, int dtype = SPARTA_F32, int n_cus = 0);
// y (+)= the tiles' part of A . x, walked on the HOST from the device form (test aid for the CPU suite: the layout of plan and slices without a GPU)
void union_plan_host_apply(const UnionDevPlan& P, const float* x, double* y);

constexpr int64_t kZeroRangeRows = 2048;    // block-rows without blocks at least this tall are zero-filled by vbs_zero_rows_kernel
int build_stream_plans(const StreamPlanIn& in, StreamPlanHost& P);
uint16_t to_h16(float v, bool bf16);   // fp32 -> fp16 / bf16 bits, round to nearest even (what the device conversion kernel does too)

}  // namespace sparta_dev
