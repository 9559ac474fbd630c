// vbs_spmm.hip -- VBS A x dense B on MI355X (gfx950 / CDNA4): device image, tile plan and the
// hand-written kernels behind sparta_vbs_create / sparta_vbs_spmm (include/sparta_amd.h).
//
// What it replaces in the reference: the CPU triple loop VBR::multiply (src/general/vbr.cpp:323-372)
// and the "one library GEMM per nonzero block" GPU back-ends (src/cuda/cuda_utilities.cpp:39-887,
// src/cuda/cutlass_bellpack_lib.cu:380-1019).  Not a translation of either: there is ONE fused kernel
// family, no vendor BLAS/SPARSE call, no per-block launch.
//
// Design (see DESIGN.md §3 for the long form):
//   * A keeps the reference's VBS layout in HBM (column-major h x w blocks, blocks of a block-row
//     back to back), so a block-row IS a dense column-major h x (nb*w) matrix with lda = h.
//   * The host cuts every block-row into ROW TILES of <=128 / <=64 / <=32 / <=16 rows ("classes");
//     one 256-thread workgroup owns one (row tile, 128-column slab of C) and walks the block-row's
//     nonzero blocks.  Per block ("panel step") the workgroup stages the w x 128 panel of B
//     (gathered through jab) and the tile's slice of the A block through LDS, then every wave runs
//     fp32 MFMAs on it: v_mfma_f32_32x32x2_f32 for tiles >16 rows, v_mfma_f32_16x16x4_f32 for thin
//     (<=16 row) tiles so ragged clusters do not pay 32-row padding.
//   * The product is computed TRANSPOSED inside the MFMA (D = Bpanel^T * Atile^T): the accumulator
//     then holds, per register, 32 (16) consecutive ROWS of one column of C, so the column-major C
//     of the reference is written in 128-byte (64-byte) contiguous runs.
//   * Global->LDS staging is register-prefetched one panel ahead (loads for step s+1 are issued
//     before the MFMAs of step s), so HBM/L2 latency hides under the fp32 MFMAs (64 cycles each).
//   * The workgroup->tile map is XCD-aware: the 8 XCDs each get a contiguous range of the tile list,
//     so neighbouring block-rows (which gather the same B panels) share one L2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "host_core.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned: global_load_dwordx4 on any float address
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
};

template <int MF>
struct Acc;
template <>
struct Acc<32> { typedef f32x16 type; };
template <>
struct Acc<16> { typedef f32x4 type; };

// MF: MFMA tile (32 -> 32x32x2, 16 -> 16x16x4).  WM x WN waves, each MI x NI MFMA tiles.  KP: k-depth of a
// panel step.  BRM: B is row-major.  GENERIC: any w / N / alignment (slow, branchy staging); the non-GENERIC
// instantiation requires w % KP == 0, N % 128 == 0 (checked on the host) and keeps its main loop free of
// any data-dependent branch: every load of a step is an unconditional 16-byte load, so the loads of step
// s+1 stay in flight under the MFMAs of step s.  The one irregular case it still meets -- the zero-padded
// LAST block column when cols % w != 0, which can only be the last block of a block-row because jab is
// ascending -- is peeled out of the loop into a single slow step (TILE_TAIL flag of the tile).
template <int MF, int WM, int WN, int MI, int NI, int KP, bool BRM, bool GENERIC>
__global__ __launch_bounds__(kThreads) void vbs_spmm_f32_kernel(const SpmmParams p) {
    constexpr int TM = WM * MI * MF;
    constexpr int TN = WN * NI * MF;
    static_assert(TN == kTN, "workgroup covers 128 columns");
    static_assert(WM * WN == 4, "4 waves");
    // LDS image of the B panel: column-major B -> Bs[j][k] (k contiguous, +4 pad: conflict-free ds_read_b128);
    //                           row-major B    -> Bs[k][j] (j contiguous: conflict-free ds_read_b32)
    constexpr int LDBS = BRM ? TN : KP + 4;
    constexpr int BS_FLOATS = BRM ? KP * TN : TN * (KP + 4);
    constexpr int LDAS = (TM == 16) ? 20 : TM;      // As[k][i]; 16-row tiles: shift the upper k-quarters onto the other banks
    constexpr int NBV = TN * KP / 4 / kThreads;     // 16-byte B chunks staged per thread
    constexpr int NAV = TM * KP / 4 / kThreads;     // 16-byte A chunks staged per thread
    constexpr int KC = KP / 4;                      // col-major B: chunks per panel column
    constexpr int BJ_STEP = kThreads / KC;          // col-major B: columns between a thread's consecutive chunks
    constexpr int BK_STEP = kThreads / (TN / 4);    // row-major B: k rows between a thread's consecutive chunks
    constexpr int AC = TM / 4;                      // A: chunks per k column
    constexpr int AK_STEP = kThreads / AC;          // A: k between a thread's consecutive chunks
    constexpr int KG = (MF == 32) ? 8 : 16;         // k consumed per fragment round
    constexpr int NACC = (MF == 32) ? 16 : 4;
    static_assert(KP % KG == 0 && NBV >= 1 && NAV >= 1 && kThreads % AC == 0, "bad tile configuration");
    typedef typename Acc<MF>::type acc_t;

    __shared__ __attribute__((aligned(16))) float lds[BS_FLOATS + KP * LDAS];
    float* Bs = lds;
    float* As = lds + BS_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lm = lane & (MF - 1);                 // row/col inside the MFMA tile
    const int g = lane / MF;                        // k lane-group (half for 32x32x2, quarter for 16x16x4)

    // tile list is pre-arranged on the host (see sparta_vbs_create): entry t belongs to XCD t % 8 (workgroups are
    // dealt round-robin over the 8 XCDs), each XCD owning a contiguous range of block-rows sorted by descending cost
    const int tile_id = blockIdx.x % p.n_tiles;
    const int n0 = (blockIdx.x / p.n_tiles) * TN;
    const TileDesc td = p.tiles[tile_id];
    const int mt = td.mt_flags & 0xffff;
    const bool tail_partial = !GENERIC && (td.mt_flags & TILE_TAIL) != 0 && p.shard_rows == 0;
    const int w = p.w, N = p.N;
    const int spb = (w + KP - 1) / KP;              // panel steps per block
    const int nsteps = (td.nb - (tail_partial ? 1 : 0)) * spb;   // steps of the regular loop

    acc_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int r = 0; r < NACC; r++) acc[mi][ni][r] = 0.0f;

    f32x4 breg[NBV];
    f32x4 areg[NAV];

    // per-thread staging coordinates (constant over the whole tile)
    const int bj0 = tid / KC, bk = (tid % KC) * 4;                  // col-major B: column bj0 + BJ_STEP*q, k = bk..bk+3
    const int rj = (tid % (TN / 4)) * 4, rk0 = tid / (TN / 4);      // row-major B: k = rk0 + BK_STEP*q, columns rj..rj+3
    const int ai = (tid % AC) * 4, ak0 = tid / AC;                  // A: rows ai..ai+3, k = ak0 + AK_STEP*q

    // where panel step (b, ks) finds its B rows: base pointer, first row, number of rows that exist
    struct PanelSrc { const float* base; int64_t gk0; int64_t bvalid; };
    auto panel_src = [&](int b, int ks) {
        PanelSrc ps;
        const int64_t jb = p.jab[td.jab_off + b];
        ps.gk0 = jb * (int64_t)w + ks;
        ps.base = p.B;
        ps.bvalid = p.cols;
        if (p.shard_rows > 0) {
            // gathered B: rank s contributed rows [s*shard_rows, (s+1)*shard_rows) as its own column-major slab;
            // shard_rows is a multiple of w, so a panel never straddles two slabs (wave-uniform arithmetic)
            const int64_t sh = ps.gk0 / p.shard_rows;
            ps.base += sh * p.shard_stride;
            ps.gk0 -= sh * p.shard_rows;
            ps.bvalid = p.shard_rows;
        }
        return ps;
    };

    // A slice of block b, k in [ks, ks+KP): 16-byte loads along the rows of a column.  Rows past the tile (mt < TM)
    // read the following rows / the next column / the 128-float pad behind A: finite garbage that only reaches
    // accumulator rows which are never stored.  Needs no mask when the k range is full (w % KP == 0).
    auto load_a_fast = [&](int b, int ks) {
        const float* asrc = p.A + td.a_off + ((int64_t)b * w + ks + ak0) * td.h + ai;
#pragma unroll
        for (int q = 0; q < NAV; q++) {
            const f32x4u t = *reinterpret_cast<const f32x4u*>(asrc + (int64_t)(q * AK_STEP) * td.h);
            areg[q] = (f32x4){t.x, t.y, t.z, t.w};
        }
    };

    // ---- stage loader: global -> registers ------------------------------------------------------
    auto load_step = [&](int s) {
        const int b = s / spb;
        const int ks = (s - b * spb) * KP;
        const PanelSrc ps = panel_src(b, ks);
        if constexpr (!GENERIC) {
            if constexpr (!BRM) {
                const float* src = ps.base + ps.gk0 + bk + (int64_t)(n0 + bj0) * p.ldb;
#pragma unroll
                for (int q = 0; q < NBV; q++) {
                    const f32x4u t = *reinterpret_cast<const f32x4u*>(src + (int64_t)(q * BJ_STEP) * p.ldb);
                    breg[q] = (f32x4){t.x, t.y, t.z, t.w};
                }
            } else {
                const float* src = ps.base + (ps.gk0 + rk0) * p.ldb + n0 + rj;
#pragma unroll
                for (int q = 0; q < NBV; q++) {
                    const f32x4u t = *reinterpret_cast<const f32x4u*>(src + (int64_t)(q * BK_STEP) * p.ldb);
                    breg[q] = (f32x4){t.x, t.y, t.z, t.w};
                }
            }
            load_a_fast(b, ks);
        } else {
            const int kp_a = min(KP, w - ks);                                         // k that exist in the block
            const int kp_b = (int)min((int64_t)kp_a, ps.bvalid - ps.gk0);             // ... and in B
            if constexpr (!BRM) {
#pragma unroll
                for (int q = 0; q < NBV; q++) {
                    const int col = n0 + bj0 + BJ_STEP * q;
                    const float* src = ps.base + ps.gk0 + bk + (int64_t)col * p.ldb;
                    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (col < N) {
                        if (p.vec_ok && bk + 3 < kp_b) {
                            const f32x4u t = *reinterpret_cast<const f32x4u*>(src);
                            v = (f32x4){t.x, t.y, t.z, t.w};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (bk + e < kp_b) v[e] = src[e];
                        }
                    }
                    breg[q] = v;
                }
            } else {
#pragma unroll
                for (int q = 0; q < NBV; q++) {
                    const int k = rk0 + BK_STEP * q;
                    const int col = n0 + rj;
                    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (k < kp_b) {
                        const float* src = ps.base + (ps.gk0 + k) * p.ldb + col;
                        if (p.vec_ok && col + 3 < N) {
                            const f32x4u t = *reinterpret_cast<const f32x4u*>(src);
                            v = (f32x4){t.x, t.y, t.z, t.w};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (col + e < N) v[e] = src[e];
                        }
                    }
                    breg[q] = v;
                }
            }
            const float* asrc = p.A + td.a_off + ((int64_t)b * w + ks) * td.h;
#pragma unroll
            for (int q = 0; q < NAV; q++) {
                const int k = ak0 + AK_STEP * q;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (k < kp_a) {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (ai + e < mt) v[e] = asrc[(int64_t)k * td.h + ai + e];
                }
                areg[q] = v;
            }
        }
    };

    // ---- registers -> LDS -----------------------------------------------------------------------
    auto store_a = [&]() {
#pragma unroll
        for (int q = 0; q < NAV; q++) *reinterpret_cast<f32x4*>(&As[(ak0 + AK_STEP * q) * LDAS + ai]) = areg[q];
    };
    auto store_step = [&]() {
        if constexpr (!BRM) {
#pragma unroll
            for (int q = 0; q < NBV; q++) *reinterpret_cast<f32x4*>(&Bs[(bj0 + BJ_STEP * q) * LDBS + bk]) = breg[q];
        } else {
#pragma unroll
            for (int q = 0; q < NBV; q++) *reinterpret_cast<f32x4*>(&Bs[(rk0 + BK_STEP * q) * LDBS + rj]) = breg[q];
        }
        store_a();
    };

    // ---- MFMA over one staged panel -----------------------------------------------------------------
    // Fragment k-mapping: MFMA number m of a round takes, from lane-group g, k = kb + 4g + m for BOTH operands
    // (any bijection works: an MFMA just sums over its k slots).  Column-major B: one ds_read_b128 gives a lane the
    // 4 consecutive k of its column; row-major B: four ds_read_b32, each conflict-free across the 32 columns.
    const float* a_frag = As + (4 * g) * LDAS + wm * MI * MF + lm;
    const float* b_frag = BRM ? Bs + (4 * g) * LDBS + wn * NI * MF + lm : Bs + (wn * NI * MF + lm) * LDBS + 4 * g;
    auto mfma_round = [&](int kb) {
        float a[MI][4];
        f32x4 bf[NI];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int m = 0; m < 4; m++) a[mi][m] = a_frag[(kb + m) * LDAS + mi * MF];
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            if constexpr (!BRM) {
                bf[ni] = *reinterpret_cast<const f32x4*>(&b_frag[ni * MF * LDBS + kb]);
            } else {
#pragma unroll
                for (int m = 0; m < 4; m++) bf[ni][m] = b_frag[(kb + m) * LDBS + ni * MF];
            }
        }
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++) {
                    if constexpr (MF == 32)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[ni][m], a[mi][m], acc[mi][ni], 0, 0, 0);
                    else
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[ni][m], a[mi][m], acc[mi][ni], 0, 0, 0);
                }
    };

    if (nsteps > 0) load_step(0);
    for (int s = 0; s < nsteps; s++) {
        __syncthreads();                            // everyone finished reading the previous panel
        store_step();
        __syncthreads();
        if (s + 1 < nsteps) load_step(s + 1);       // in flight while the MFMAs below run
        if constexpr (!GENERIC) {
#pragma unroll
            for (int kb = 0; kb < KP; kb += KG) mfma_round(kb);
        } else {
            const int kp = min(KP, w - (s % spb) * KP);
            for (int kb = 0; kb < kp; kb += KG) mfma_round(kb);   // short block: the staged tail is zero-filled
        }
    }

    if constexpr (!GENERIC) {
        if (tail_partial) {
            // peeled step: the block in the zero-padded last block column.  B rows >= cols do not exist: staged as
            // zeros by a plain bounds-checked loop (slow, once per block-row at most); A is full width as stored.
            const int b = td.nb - 1;
            for (int ks = 0; ks < w; ks += KP) {
                const PanelSrc ps = panel_src(b, ks);
                load_a_fast(b, ks);
                __syncthreads();
#pragma unroll 1
                for (int idx = tid; idx < TN * KP; idx += kThreads) {
                    float v = 0.0f;
                    if constexpr (!BRM) {
                        const int j = idx / KP, k = idx % KP;
                        if (ps.gk0 + k < ps.bvalid) v = ps.base[ps.gk0 + k + (int64_t)(n0 + j) * p.ldb];
                        Bs[j * LDBS + k] = v;
                    } else {
                        const int k = idx / TN, j = idx % TN;
                        if (ps.gk0 + k < ps.bvalid) v = ps.base[(ps.gk0 + k) * p.ldb + n0 + j];
                        Bs[k * LDBS + j] = v;
                    }
                }
                store_a();
                __syncthreads();
#pragma unroll
                for (int kb = 0; kb < KP; kb += KG) mfma_round(kb);
            }
        }
    }

    // ---- epilogue: D[j][i] -> C[i][j] -----------------------------------------------------------------
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        const int row = (wm * MI + mi) * MF + lm;
        if (row >= mt) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
#pragma unroll
            for (int r = 0; r < NACC; r++) {
                const int j = (MF == 32) ? ((r & 3) + 8 * (r >> 2) + 4 * g) : (4 * g + r);
                const int col = n0 + (wn * NI + ni) * MF + j;
                if (GENERIC && col >= N) continue;
                float* dst = p.c_row_major ? p.C + (int64_t)(td.c_row + row) * p.ldc + col
                                           : p.C + (int64_t)(td.c_row + row) + (int64_t)col * p.ldc;
                float v = acc[mi][ni][r];
                if (p.accumulate) v += *dst;
                *dst = v;
            }
        }
    }
}

// ---- exact-order kernel (parity aid) -----------------------------------------------------------------
// One thread per element of C; the sum runs block by block, k ascending, with an UNFUSED multiply and
// add -- the operation order and rounding of the reference's loop nest (src/general/vbr.cpp:358-363,
// compiled for baseline x86-64: no FMA).  Bit-identical to VBR::multiply for finite inputs.
struct BlockRowDesc {
    int64_t a_off, jab_off;
    int32_t nb, h, c_row, pad;
};

__global__ __launch_bounds__(kThreads) void vbs_spmm_f32_exact_kernel(const BlockRowDesc* rows, const int32_t* jab, const float* A,
                                                                      const float* B, float* C, int64_t ldb, int64_t ldc,
                                                                      int64_t cols, int N, int w, int b_row_major, int c_row_major,
                                                                      int accumulate, int64_t shard_rows, int64_t shard_stride) {
#pragma clang fp contract(off)
    const BlockRowDesc br = rows[blockIdx.x];
    const int64_t total = (int64_t)br.h * N;
    for (int64_t idx = threadIdx.x; idx < total; idx += kThreads) {
        const int i = (int)(idx % br.h);
        const int j = (int)(idx / br.h);
        float* dst = c_row_major ? C + (int64_t)(br.c_row + i) * ldc + j : C + (int64_t)(br.c_row + i) + (int64_t)j * ldc;
        float c = accumulate ? *dst : 0.0f;
        for (int b = 0; b < br.nb; b++) {
            const int64_t gk0 = (int64_t)jab[br.jab_off + b] * w;
            const float* ablk = A + br.a_off + (int64_t)b * w * br.h + i;
            for (int k = 0; k < w; k++) {
                const int64_t gk = gk0 + k;
                // the reference reads B out of bounds here when cols % w != 0 (vbr.cpp:351,362) and relies on
                // the matching A entry being a stored zero; we define that product as 0 * 0.
                float bv = 0.0f;
                if (gk < cols) {
                    if (shard_rows > 0) bv = B[(gk / shard_rows) * shard_stride + (gk % shard_rows) + (int64_t)j * ldb];
                    else bv = b_row_major ? B[gk * ldb + j] : B[gk + (int64_t)j * ldb];
                }
                const float prod = ablk[(int64_t)k * br.h] * bv;
                c = c + prod;
            }
        }
        *dst = c;
    }
}

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

}  // namespace

struct sparta_vbs {
    int device = 0, dtype = SPARTA_F32;
    int64_t rows = 0, cols = 0, block_rows = 0, w = 0, nblocks = 0, nztot = 0;
    float* d_A = nullptr;
    int32_t* d_jab = nullptr;
    TileDesc* d_tiles[4] = {nullptr, nullptr, nullptr, nullptr};   // classes 16, 32, 64, 128
    int64_t n_tiles[4] = {0, 0, 0, 0};       // launch entries (real tiles + padding)
    int64_t n_real_tiles[4] = {0, 0, 0, 0};
    BlockRowDesc* d_brows = nullptr;
    int64_t n_brows = 0;
    int64_t exec_area = 0;
    int64_t a_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t cev[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    bool class_timing = false;
    bool class_ran[4] = {false, false, false, false};
    void* d_B = nullptr;
    size_t d_B_bytes = 0;
    void* d_C = nullptr;
    size_t d_C_bytes = 0;
};

namespace {

template <int MF, int WM, int WN, int MI, int NI, int KP, bool BRM, bool GENERIC>
void launch_class(const SpmmParams& p, hipStream_t st) {
    if (p.n_tiles == 0) return;
    const int64_t grid = (int64_t)p.n_tiles * p.n_ntiles;
    hipLaunchKernelGGL((vbs_spmm_f32_kernel<MF, WM, WN, MI, NI, KP, BRM, GENERIC>), dim3((unsigned)grid), dim3(kThreads), 0, st, p);
}

bool force_generic() { const char* e = std::getenv("SPARTA_FORCE_GENERIC"); return e && e[0] == '1'; }

template <bool BRM, bool GENERIC>
void launch_tile_class(int c, const SpmmParams& p, hipStream_t st) {
    switch (c) {
        case 0: launch_class<16, 1, 4, 1, 2, kKP, BRM, GENERIC>(p, st); break;   // <=16 x 128, 16x16x4 MFMA
        case 1: launch_class<32, 1, 4, 1, 1, kKP, BRM, GENERIC>(p, st); break;   // <=32 x 128
        default: launch_class<32, 2, 2, 1, 2, kKP, BRM, GENERIC>(p, st); break;  // <=64 x 128
    }
}

int ensure_scratch(void** ptr, size_t* have, size_t need) {
    if (*have >= need) return SPARTA_OK;
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
    if (v->d_B) (void)hipFree(v->d_B);
    if (v->d_C) (void)hipFree(v->d_C);
    if (v->ev0) (void)hipEventDestroy(v->ev0);
    if (v->ev1) (void)hipEventDestroy(v->ev1);
    for (int c = 0; c < 4; c++)
        for (int e = 0; e < 2; e++)
            if (v->cev[c][e]) (void)hipEventDestroy(v->cev[c][e]);
    delete v;
}

}  // namespace

extern "C" {

int sparta_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sparta_vbs_create_range(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t w, const int64_t* row_part,
                            const int64_t* nzcount, const int64_t* jab, const float* mab, int64_t br0, int64_t br1, int32_t dtype,
                            int32_t device) {
    using sparta::fail;
    if (!out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: out is NULL");
    *out = nullptr;
    if (rows <= 0 || cols <= 0 || block_rows <= 0 || w <= 0 || !row_part || !nzcount)
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: bad dimensions or NULL index array");
    if (br0 < 0 || br1 > block_rows || br0 >= br1) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: bad block-row range");
    if (dtype != SPARTA_F32) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: only SPARTA_F32 is implemented in this build");
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
    if (ndev <= 0) return fail(SPARTA_ERR_NO_DEVICE, "sparta_vbs_create: no HIP device visible (this path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: device index out of range");

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
                int64_t r0 = 0;
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

    sparta_vbs* v = new (std::nothrow) sparta_vbs;
    if (!v) return fail(SPARTA_ERR_ALLOC, "sparta_vbs_create: out of host memory");
    v->device = device; v->dtype = dtype;
    v->rows = row_part[br1] - row_part[br0]; v->cols = cols; v->block_rows = br1 - br0; v->w = w;
    v->nblocks = nblocks; v->nztot = nztot; v->exec_area = exec_area;

    DeviceGuard guard(device);
    if (!guard.ok) { delete v; return fail(SPARTA_ERR_HIP, "sparta_vbs_create: hipSetDevice failed"); }
#define CREATE_TRY(expr)                                                                                     \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
            destroy_impl(v);                                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? SPARTA_ERR_ALLOC : SPARTA_ERR_HIP, m_);                  \
        }                                                                                                    \
    } while (0)
    // A is padded by 128 floats so that no (masked-off) lane ever forms an address past the allocation
    v->a_bytes = (nztot + 128) * (int64_t)sizeof(float);
    CREATE_TRY(hipMalloc((void**)&v->d_A, (size_t)v->a_bytes));
    CREATE_TRY(hipMemset(v->d_A, 0, (size_t)v->a_bytes));
    if (nztot > 0) CREATE_TRY(hipMemcpy(v->d_A, mab + mab_lo, (size_t)nztot * sizeof(float), hipMemcpyHostToDevice));
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
    CREATE_TRY(hipEventCreate(&v->ev0));
    CREATE_TRY(hipEventCreate(&v->ev1));
#undef CREATE_TRY
    *out = v;
    return SPARTA_OK;
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
    for (int c = 0; c < 4; c++) info[6 + c] = A->n_real_tiles[c];
    info[10] = A->a_bytes; info[11] = A->exec_area;
    return SPARTA_OK;
}

}  // extern "C"

namespace {

// shared implementation of sparta_vbs_spmm / sparta_vbs_spmm_gathered
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

    if (dt_ms) HIP_TRY(hipEventRecord(A->ev0, st));
    if (algo == SPARTA_SPMM_EXACT) {
        if (A->n_brows > 0) {
            hipLaunchKernelGGL(vbs_spmm_f32_exact_kernel, dim3((unsigned)A->n_brows), dim3(kThreads), 0, st, A->d_brows, A->d_jab,
                               A->d_A, dB, dC, ldb, ldc, A->cols, (int)n_cols, (int)A->w, (int)(b_layout == SPARTA_ROW_MAJOR),
                               (int)(c_layout == SPARTA_ROW_MAJOR), (int)accumulate, shard_rows, shard_stride);
        }
    } else {
        SpmmParams p;
        p.jab = A->d_jab; p.A = A->d_A; p.B = dB; p.C = dC; p.ldb = ldb; p.ldc = ldc; p.cols = A->cols;
        p.n_ntiles = (n_cols + kTN - 1) / kTN; p.N = n_cols; p.w = (int32_t)A->w;
        p.b_row_major = b_layout == SPARTA_ROW_MAJOR; p.c_row_major = c_layout == SPARTA_ROW_MAJOR;
        p.accumulate = accumulate != 0;
        p.shard_rows = shard_rows; p.shard_stride = shard_stride;
        const char* nv = std::getenv("SPARTA_NO_VEC");
        p.vec_ok = (nv && nv[0] == '1') ? 0 : 1;
        const bool prof = A->class_timing;
        // the branch-free kernels need full panels: w a multiple of the panel depth, N a multiple of the slab width
        const bool generic = !p.vec_ok || (A->w % kKP) != 0 || (n_cols % kTN) != 0 || force_generic();
        for (int c = 3; c >= 0; c--) {              // heavy classes first
            A->class_ran[c] = false;
            if (A->n_tiles[c] == 0) continue;
            if (A->n_tiles[c] * (int64_t)p.n_ntiles > INT32_MAX) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: grid too large");
            p.tiles = A->d_tiles[c]; p.n_tiles = (int32_t)A->n_tiles[c];
            if (prof) HIP_TRY(hipEventRecord(A->cev[c][0], st));
            if (generic) { if (p.b_row_major) launch_tile_class<true, true>(c, p, st); else launch_tile_class<false, true>(c, p, st); }
            else { if (p.b_row_major) launch_tile_class<true, false>(c, p, st); else launch_tile_class<false, false>(c, p, st); }
            if (prof) { HIP_TRY(hipEventRecord(A->cev[c][1], st)); A->class_ran[c] = true; }
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
    return spmm_impl(A, B, ldb, b_layout, 0, 0, n_cols, C, ldc, c_layout, accumulate, ptr_space, stream, algo, dt_ms);
}

int sparta_vbs_spmm_gathered(sparta_vbs_t* A, const void* B_gathered, int64_t shard_rows, int64_t shard_stride, int32_t n_cols,
                             void* C, int64_t ldc, int32_t c_layout, int32_t accumulate, void* stream, int32_t algo, float* dt_ms) {
    using sparta::fail;
    if (!A) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: NULL handle");
    if (shard_rows <= 0 || shard_rows % A->w != 0)
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: shard_rows must be a positive multiple of block_col_size");
    if (shard_stride < shard_rows * (int64_t)n_cols) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: shard_stride too small");
    if (A->cols % shard_rows != 0) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: cols must be n_shards * shard_rows");
    return spmm_impl(A, B_gathered, shard_rows, SPARTA_COL_MAJOR, shard_rows, shard_stride, n_cols, C, ldc, c_layout, accumulate,
                     SPARTA_PTR_DEVICE, stream, algo, dt_ms);
}

int sparta_vbs_set_class_timing(sparta_vbs_t* A, int32_t enable) {
    using sparta::fail;
    if (!A) return fail(SPARTA_ERR_INVALID, "sparta_vbs_set_class_timing: NULL handle");
    DeviceGuard guard(A->device);
    if (enable && !A->cev[0][0]) {
        for (int c = 0; c < 4; c++)
            for (int e = 0; e < 2; e++) HIP_TRY(hipEventCreate(&A->cev[c][e]));
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

}  // extern "C"
