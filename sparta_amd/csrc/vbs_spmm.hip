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
#include <type_traits>
#include <utility>
#include <vector>

#include "host_core.hpp"

namespace {

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

// Clock probe: s_memtime counts shader-clock cycles, s_memrealtime a constant 100 MHz; the ratio over a kernel's lifetime is
// the clock the MFMA pipes actually ran at (the board drops it under a dense fp32 MFMA load: DESIGN.md, "clock").
// Developer instrumentation (make TIMELINE=1 -> libsparta_amd_tl.so; never in the product build): s_memtime stamps inside the steps
// of one workgroup of the fp32 stream kernel, written behind the 16 clock-probe words.  scripts/timeline.py reads them.
#ifdef SPARTA_TIMELINE
#define TL_STEPS 64
#define TL_FIRST 16
#define TL_STAMP(k) do { if (tl_on) asm volatile("s_memtime %0" : "=s"(tl[k])); } while (0)
#else
#define TL_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void clock_probe(long long* clk, int slot) {
    if (clk != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        clk[slot] = (long long)__builtin_readcyclecounter();
        clk[slot + 1] = (long long)wall_clock64();
    }
}

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

    clock_probe(p.clk, 0);
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
    clock_probe(p.clk, 2);
}

// =====================================================================================================
// Persistent "stream" kernel -- the product path for w % 32 == 0, N % 128 == 0.
//
// Why: a VBS multiply on one MI355X is a few thousand short tiles (a block-row tile has ~10 nonzero blocks).
// Launching one workgroup per tile loses a third of the machine to (i) the exposed descriptor -> jab ->
// first-panel latency chain at the start of every tile and (ii) quantisation (4 tiles of uneven length per CU).
// Here the host flattens all tiles into ONE sequence of 32-deep "steps" (tile after tile, block after
// block) and cuts it into P = 2 x #CU contiguous ranges of equal modelled cost; worker p (a persistent
// 256-thread workgroup) streams through its range with a software pipeline that never drains at a tile
// boundary:
//        G(i+3): global -> registers (2 register sets, ~2 steps of latency budget)
//        W(i+1): registers -> LDS stage (i+1)&1        (interleaved between the MFMAs of step i)
//        C(i)  : ds_read fragments (one round ahead) + v_mfma_f32_32x32x2_f32 from LDS stage i&1
//        one s_barrier per step.
// A tile whose steps straddle a range boundary is "split": each worker stores its partial accumulator to a
// workspace slot and a small fix-up kernel adds the (<= P-1) split tiles' slots in a fixed order -- no
// atomics, bit-reproducible.  Workers of one XCD get a contiguous range, so neighbouring block-rows (which
// gather the same B panels) share an L2; A is streamed with non-temporal loads so it does not evict B.
// Wave v owns columns [32v, 32v+32) of the 128-column slab and one or two 32-row MFMA tiles (rows 0-31,
// 32-63 when the tile has more than 32 rows).
// =====================================================================================================
constexpr int SK_KP = 32;                 // k depth of a step
constexpr int SK_TM = 64;                 // max rows of a tile
constexpr int32_t STEP_FIRST = 1 << 16;   // first step of a (segment of a) tile: accumulators start at zero
constexpr int32_t STEP_LAST = 1 << 17;    // last step of a (segment of a) tile: run the epilogue
constexpr int32_t STEP_SPLIT = 1 << 18;   // the tile is shared with another worker: epilogue goes to the workspace
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

__device__ __forceinline__ int32_t sk_field(int vrec0, int vrec1, int s, int f) {   // field f of step s's record (see the kernels)
    const int ln = ((s & 7) << 3) + f;
    const int32_t x0 = __builtin_amdgcn_readlane(vrec0, ln), x1 = __builtin_amdgcn_readlane(vrec1, ln);
    return ((s >> 3) & 1) ? x1 : x0;
}

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
};

// D[j][i] register image -> C.  lane: i = lane & 31 (row), g = lane >> 5; register r: j = (r&3) + 8(r>>2) + 4g.
__device__ __forceinline__ void sk_store_tile(const f32x16& acc0, const f32x16& acc1, int mt, int c_row, int col0, float* C,
                                              int64_t ldc, int c_row_major, int accumulate, int lm, int g) {
#pragma unroll
    for (int mi = 0; mi < 2; mi++) {
        const int row = mi * 32 + lm;
        if (row >= mt) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int col = col0 + (r & 3) + 8 * (r >> 2) + 4 * g;
            float* dst = c_row_major ? C + (int64_t)(c_row + row) * ldc + col : C + (int64_t)(c_row + row) + (int64_t)col * ldc;
            float v = mi == 0 ? acc0[r] : acc1[r];
            if (accumulate) v += *dst;
            *dst = v;
        }
    }
}

// Instruction budget of the loop.  On gfx950 v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 vector datapath (it has
// exactly the fp32 VALU rate): LDS and vector-memory instructions issue underneath a running MFMA, ordinary VALU
// instructions do NOT -- every v_add/v_cndmask/v_readlane of either co-resident wave takes the pipe away from the
// MFMAs (measured: scripts/ubench/mfma_overlap.hip, 8 VALU ops per 2 MFMAs = +23 %).  So the steady state keeps
// VALU work near zero: all per-step quantities live in SGPRs (records arrive through v_readlane, cursors advance on
// the scalar unit), global loads are buffer loads (per-thread byte offset computed once, per-step base in the
// scalar descriptor / soffset), LDS addresses are per-thread constants plus immediates (stage parity is a template
// argument), and the B-tail / gathered-B variations are scalar selects.
template <bool BRM, bool GATHERED, bool MI2>
__global__ __launch_bounds__(kThreads, 2) void vbs_spmm_f32_stream_kernel(const StreamParams p) {
    constexpr int KP = SK_KP, TN = kTN;
    // rows of the A slice staged per step: a one-MFMA-tile launch stages 32 rows, not 64.  Its tiles are bound by the
    // L2 -> CU load path, not by MFMA (16 KB of B panel per 2 x 32 x 128 x 32 flop): every byte not loaded counts.
    constexpr int TM = MI2 ? SK_TM : 32;
    constexpr int LDB = BRM ? TN : KP + 4;          // col-major B: Bs[j][k] (+4: conflict-free ds_read_b128); row-major: Bs[k][j]
    constexpr int BSZ = BRM ? KP * TN : TN * (KP + 4);
    constexpr int STAGE = BSZ + KP * SK_TM;         // floats per LDS stage (B panel + A slice As[k][i]; sized for 64 rows in both instantiations)
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 31, g = lane >> 5;
    const int n0 = blockIdx.y * TN;
    const int s_begin = p.worker_range[2 * blockIdx.x];
    const int n = p.worker_range[2 * blockIdx.x + 1] - s_begin;
    if (n <= 0) return;
    clock_probe(p.clk, 0);
#ifdef SPARTA_TIMELINE
    long long tl_t0 = 0;
    if (p.clk != nullptr && blockIdx.y == 0 && tid == 0 && blockIdx.x < 1024) tl_t0 = wall_clock64();
#endif
    float* ws = p.ws + (int64_t)blockIdx.y * p.ws_slab_stride;

    // ---- step records, read COALESCED and kept in registers -----------------------------------------------
    // A record is 8 dwords; one 256-byte wave load brings 8 consecutive records into one VGPR (lane = 8*rec + field)
    // and v_readlane hands a field to the scalar unit when its step comes up.  Two such VGPRs (batches b, b+1) cover
    // the pipeline's look-ahead of 3 steps; a batch is requested 5+ steps before its first use, so the control stream
    // never waits on memory and issues no scalar load (SMEM shares lgkmcnt with LDS and returns out of order: one
    // pending s_load turns every fragment wait into lgkmcnt(0)).
    const int32_t* srec = reinterpret_cast<const int32_t*>(p.steps + s_begin);
    int vrec0 = srec[lane];
    int vrec1 = srec[64 + lane];
    int vnext = 0;
    // a macro over a free function, not a lambda: every closure between the loop body and vrec0 / vrec1 is one more level of
    // pointer indirection the optimiser has to peel before it can keep them in registers (three levels deep it gave up and
    // left one of them in memory: an LDS / scratch read behind a full wait in every step)
#define field(s, f) sk_field(vrec0, vrec1, (s), (f))
    enum { F_AOFF_LO = 0, F_AOFF_HI = 1, F_BROW = 2, F_H = 3, F_CROW = 4, F_FLAGS = 5, F_SLOT = 6, F_SHARD = 7 };

    // per-thread constant byte offsets (the only vector part of any address in the loop)
    const int bj0 = tid >> 3, bk = (tid & 7) * 4;       // col-major B: column bj0 + 32q, k = bk..bk+3   (q = 0..3)
    const int rk0 = tid >> 5, rj = (tid & 31) * 4;      // row-major B: k = rk0 + 8q, columns rj..rj+3
    // A: MI2: k = ak0 + 16q (q = 0..1), rows ai..ai+3 of 64;  else: k = ak0 (0..31), rows ai..ai+3 of 32 -- one load per lane
    const int ak0 = MI2 ? tid >> 4 : tid >> 3, ai = MI2 ? (tid & 15) * 4 : (tid & 7) * 4;
    const int64_t ld_t = BRM ? (int64_t)p.N : (int64_t)p.w;                      // leading dimension of B_tail
    const uint32_t voffB = BRM ? (uint32_t)((rk0 * p.ldb + rj) * 4) : (uint32_t)((bk + bj0 * p.ldb) * 4);
    const uint32_t voffBt = BRM ? (uint32_t)((rk0 * ld_t + n0 + rj) * 4) : (uint32_t)((bk + (n0 + bj0) * ld_t) * 4);
    const uint32_t qstepB = (uint32_t)((BRM ? 8 : 32) * p.ldb * 4), qstepBt = (uint32_t)((BRM ? 8 : 32) * ld_t * 4);
    const int64_t n0off = BRM ? (int64_t)n0 : (int64_t)n0 * p.ldb;               // slab offset folded into the scalar base
    const uint32_t lwB = BRM ? (uint32_t)((rk0 * LDB + rj) * 4) : (uint32_t)((bj0 * LDB + bk) * 4);   // LDS write offsets (bytes)
    const uint32_t lwA = (uint32_t)((BSZ + ak0 * TM + ai) * 4);
    const uint32_t lrA = (uint32_t)((BSZ + 4 * g * TM + lm) * 4);                                    // LDS read offsets
    const uint32_t lrB = BRM ? (uint32_t)((4 * g * LDB + 32 * wave + lm) * 4) : (uint32_t)(((32 * wave + lm) * LDB + 4 * g) * 4);
    const uint32_t voffC = p.c_row_major ? (uint32_t)((lm * p.ldc + 32 * wave + 4 * g) * 4) : (uint32_t)((lm + (32 * wave + 4 * g) * p.ldc) * 4);
    char* const ldsb = reinterpret_cast<char*>(lds);

    u32x4 b0[4], a0[2], b1[4], a1[2];                   // register sets 0 / 1 of the staging pipeline (raw bits)

    // ---- G: global -> registers; steps are requested strictly in order s = 0, 1, 2, ... ----------------------
    int64_t g_aoff = 0;                                  // scalar cursor of the G stage
    int32_t g_h = 1;
    uint32_t voA_cur = 0, vo_cur = voffB;
    int32_t tail_prev = 0;
    auto issue_loads = [&](int s, u32x4 (&rb)[4], u32x4 (&ra)[2]) __attribute__((always_inline)) -> int32_t {
        const int32_t flags = field(s, F_FLAGS);
        if (flags & STEP_FIRST) {                        // tile (segment) start: re-seat the cursor, else it just advances
            g_aoff = (int64_t)(uint32_t)field(s, F_AOFF_LO) | ((int64_t)field(s, F_AOFF_HI) << 32);
            g_h = field(s, F_H);
            voA_cur = (uint32_t)(ak0 * g_h + ai) * 4u;
        } else {
            g_aoff += (int64_t)KP * g_h;                 // consecutive steps of a block-row are contiguous in A (column-major blocks back to back)
        }
        const int32_t tail = (flags & STEP_TAIL) != 0;
        if (tail != tail_prev) {
            vo_cur = tail ? voffBt : voffB;
            asm volatile("" : "+v"(vo_cur));
            tail_prev = tail;
        }
        int64_t gk0 = field(s, F_BROW);
        const float* Bbase = tail ? p.B_tail : p.B;
        if constexpr (GATHERED) {                        // a panel never straddles slabs (shard_rows % w == 0); the host split b_row into
            Bbase += (int64_t)field(s, F_SHARD) * p.shard_stride;   // (slab, row inside the slab) when it built the gathered step list:
        }                                                // a 64-bit division here costs ~40 instructions per step, 10 of them VALU
        const float* bptr = tail ? Bbase + (BRM ? gk0 * ld_t : gk0) : Bbase + (BRM ? gk0 * p.ldb : gk0) + n0off;
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bptr), 0, 0x7ffffff0, 0x00020000);
        const uint32_t qs = tail ? qstepBt : qstepB;
#pragma unroll
        for (int q = 0; q < 4; q++) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo_cur, qs * q, 0);
        // A slice: 16-byte loads along the rows of a column, streamed (nt: read exactly once).  Rows past the tile read
        // what follows in memory (next rows / next column / the pad behind A): never stored.
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + g_aoff), 0, 0x7ffffff0, 0x00020000);
        ra[0] = __builtin_amdgcn_raw_buffer_load_b128(rA, voA_cur, 0, 2);
        if constexpr (MI2) ra[1] = __builtin_amdgcn_raw_buffer_load_b128(rA, voA_cur, (uint32_t)(16 * g_h) * 4u, 2);
        return flags;
    };
    int32_t fq0 = 0, fq1 = 0, fq2 = 0, fq_new = 0;

    // ---- W: registers -> LDS stage (compile-time stage => immediate offsets) -----------------------------------
    auto write_b = [&](auto stage_tag, const u32x4 (&rb)[4], int q) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_tag)::value;
        *reinterpret_cast<u32x4*>(ldsb + lwB + (ST * STAGE + (BRM ? 8 * q * LDB : 32 * q * LDB)) * 4) = rb[q];
    };
    auto write_a = [&](auto stage_tag, const u32x4 (&ra)[2], int q) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_tag)::value;
        if (MI2 || q == 0) *reinterpret_cast<u32x4*>(ldsb + lwA + (ST * STAGE + 16 * q * TM) * 4) = ra[q];
    };

    // ---- C: fragments + MFMA ------------------------------------------------------------------------------
    struct Frag { float a[2][4]; f32x4 b; };
    uint32_t a_addr[2][4];
#pragma unroll
    for (int st = 0; st < 2; st++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            a_addr[st][r] = lrA + (uint32_t)((st * STAGE + 8 * r * TM) * 4);
            asm volatile("" : "+v"(a_addr[st][r]));
        }
    auto read_frag = [&](auto stage_tag, auto kb_tag, const bool mi2) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_tag)::value;
        constexpr int kb = decltype(kb_tag)::value;
        Frag f;
        const float* as = reinterpret_cast<const float*>(ldsb + a_addr[ST][kb / 8]);
#pragma unroll
        for (int m = 0; m < 4; m++) f.a[0][m] = as[m * TM];
        if (mi2) {
#pragma unroll
            for (int m = 0; m < 4; m++) f.a[1][m] = as[m * TM + 32];
        }
        if constexpr (!BRM) {
            f.b = *reinterpret_cast<const f32x4*>(ldsb + lrB + (ST * STAGE + kb) * 4);
        } else {
            const float* bs = reinterpret_cast<const float*>(ldsb + lrB + (ST * STAGE + kb * LDB) * 4);
#pragma unroll
            for (int m = 0; m < 4; m++) f.b[m] = bs[m * LDB];
        }
        return f;
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    auto mfma4 = [&](const Frag& f, const bool mi2) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[m], f.a[0][m], acc0, 0, 0, 0);
            if (mi2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[m], f.a[1][m], acc1, 0, 0, 0);
        }
    };

    // one pipeline iteration: compute step i from stage PAR; write step i+1 (register set wb/wa) into stage 1-PAR
    // between the MFMA rounds; refill that register set with step i+3.  The step list is padded on the host, so steps
    // i+1 .. i+3 always exist (at a range end they are the next worker's: loaded, never multiplied).
    auto iteration_t = [&](int i, int32_t flags, u32x4 (&wb)[4], u32x4 (&wa)[2], auto par_tag, auto mi2_tag) __attribute__((always_inline)) {
        constexpr bool mi2 = decltype(mi2_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;
        using cur_t = std::integral_constant<int, PAR>;
        using nxt_t = std::integral_constant<int, 1 - PAR>;
#ifdef SPARTA_TIMELINE
        unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool tl_on = p.clk != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && i >= TL_FIRST && i < TL_FIRST + TL_STEPS;
#endif
        TL_STAMP(0);
        // Straight-line rounds: fragments of round r, the LDS writes / global loads that ride along, MFMAs of round r; the
        // instruction scheduler interleaves across rounds.  Two hand-pinned orders were measured and lost: fragments one
        // round ahead inside the step (+3..8 % time) and one round ahead ACROSS the step boundary with the barrier moved
        // to the middle of the step (+3.5 %): the kernel runs against the board's power limit ("clock"), where
        // extra LDS traffic and issue slots cost more than the LDS latency they hide.
        {
            const Frag f = read_frag(cur_t{}, std::integral_constant<int, 0>{}, mi2);
            write_b(nxt_t{}, wb, 0); write_b(nxt_t{}, wb, 1);
            mfma4(f, mi2);
        }
        TL_STAMP(1);
        {
            const Frag f = read_frag(cur_t{}, std::integral_constant<int, 8>{}, mi2);
            write_b(nxt_t{}, wb, 2); write_b(nxt_t{}, wb, 3);
            mfma4(f, mi2);
        }
        TL_STAMP(2);
        {
            const Frag f = read_frag(cur_t{}, std::integral_constant<int, 16>{}, mi2);
            write_a(nxt_t{}, wa, 0); write_a(nxt_t{}, wa, 1);
            mfma4(f, mi2);
        }
        TL_STAMP(3);
        {
            const Frag f = read_frag(cur_t{}, std::integral_constant<int, 24>{}, mi2);
            fq_new = issue_loads(i + 3, wb, wa);
            mfma4(f, mi2);
        }
        TL_STAMP(4);
        if (flags & STEP_LAST) {
            // epilogue: scalar descriptor + scalar per-register offsets, the per-thread part is a kernel-lifetime constant.
            // The accumulators are cleared HERE (every segment start follows a segment end), not at STEP_FIRST: a
            // conditional clear at the top of the step is if-converted into 32 v_cndmask per step.
            if (flags & STEP_SPLIT) {
                const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(ws + (int64_t)field(i, F_SLOT) * SK_SLOT_FLOATS, 0, SK_SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc0[q]), rW, (uint32_t)tid * 4u, (uint32_t)(q * kThreads * 4), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc1[q]), rW, (uint32_t)tid * 4u, (uint32_t)((16 + q) * kThreads * 4), 0);
                }
            } else {
                const int mt = flags & 0xffff;
                const int64_t c_row = field(i, F_CROW);
                float* cbase = p.c_row_major ? p.C + c_row * p.ldc + n0 : p.C + c_row + (int64_t)n0 * p.ldc;
                const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, 0x7ffffff0, 0x00020000);
                const uint32_t jstep = p.c_row_major ? 4u : (uint32_t)p.ldc * 4u;          // bytes per output column
                const uint32_t mistep = p.c_row_major ? (uint32_t)p.ldc * 128u : 128u;     // bytes per 32 rows
#pragma unroll
                for (int mi = 0; mi < (mi2 ? 2 : 1); mi++) {
                    if (mi * 32 + lm < mt) {
                        float v[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] = mi == 0 ? acc0[q] : acc1[q];
                        if (p.accumulate) {                       // all 16 loads in flight before the first add (one wait, not 16)
                            uint32_t old[16];
#pragma unroll
                            for (int q = 0; q < 16; q++) old[q] = __builtin_amdgcn_raw_buffer_load_b32(rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 0);
#pragma unroll
                            for (int q = 0; q < 16; q++) v[q] += __uint_as_float(old[q]);
                        }
#pragma unroll
                        for (int q = 0; q < 16; q++)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; q++) { acc0[q] = 0.0f; acc1[q] = 0.0f; }
        }
        TL_STAMP(5);
        __syncthreads();
        TL_STAMP(6);
#ifdef SPARTA_TIMELINE
        if (tl_on) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) {
                long long* o = p.clk + 16 + ((int64_t)wave * TL_STEPS + (i - TL_FIRST)) * 8;
#pragma unroll
                for (int k = 0; k < 7; k++) o[k] = (long long)tl[k];
                o[7] = flags;
            }
        }
#endif
    };
    // ---- prologue: G(0) G(1) | W(0) | G(2) ------------------------------------------------------------------
    using st0 = std::integral_constant<int, 0>;
    using st1 = std::integral_constant<int, 1>;
    fq0 = issue_loads(0, b0, a0);
    fq1 = issue_loads(1, b1, a1);
#pragma unroll
    for (int q = 0; q < 4; q++) write_b(st0{}, b0, q);
#pragma unroll
    for (int q = 0; q < 2; q++) write_a(st0{}, a0, q);
    fq2 = issue_loads(2, b0, a0);
    __syncthreads();
    // Batch k+1 of the step records is requested at step 8k and only TOUCHED at step 8k+4 (first needed at 8k+5 by the
    // look-ahead of 3); the register it replaces (batch k-1) is dead by then.  Both sides are inline asm on purpose: with a
    // plain load the compiler if-converts the touch into a v_cndmask that runs EVERY step behind an s_waitcnt vmcnt(0),
    // which also drains the A/B loads issued a moment earlier.  The load is invisible to the compiler's counter
    // bookkeeping (its own waits only get stricter by it); 24 loads are issued between request and touch and memory
    // returns in order, so vmcnt(6) at the touch is a safe, free wait.
    auto batch_upkeep = [&](int i) __attribute__((always_inline)) {
        if ((i & 7) == 0 && i > 0) {
            const int32_t* nb = srec + (int64_t)((i >> 3) + 1) * 64 + lane;
            asm volatile("global_load_dword %0, %1, off" : "=&v"(vnext) : "v"(nb) : "memory");
        }
        if ((i & 7) == 4 && i > 4) {
            asm volatile("s_waitcnt vmcnt(6)" : "+v"(vnext) : : "memory");
            // two selects, not "if (odd) vrec1 = vnext; else vrec0 = vnext;": the compiler merges the two stores of that form into one store
                // through a selected POINTER, which pins vrec0 / vrec1 in scratch memory -- every step then reloads them behind vmcnt(0)
                const bool odd = (((i >> 3) + 1) & 1) != 0;
                vrec1 = odd ? vnext : vrec1;
                vrec0 = odd ? vrec0 : vnext;
        }
    };

    // Step i computes from stage i&1; the register set that holds step i+1 is set (i+1)&1.  All steps of a launch are of
    // one tile type (MI2: two 32-row MFMA tiles per wave and step, else one): no dispatch in the loop.
    using mi2_t = std::integral_constant<bool, MI2>;
    // An odd last step is peeled behind the loop rather than left as a break inside it: the compiler folds such a break into
    // the latch, its vmcnt bookkeeping then sees an edge "even step -> loop header" on which register set 1 has just been
    // refilled, and the even step's LDS writes wait with vmcnt(5..0) instead of vmcnt(11..6).
    const int n_even = n & ~1;
    for (int i = 0; i < n_even; i += 2) {
        batch_upkeep(i);
        iteration_t(i, fq0, b1, a1, st0{}, mi2_t{});
        fq0 = fq1; fq1 = fq2; fq2 = fq_new;
        iteration_t(i + 1, fq0, b0, a0, st1{}, mi2_t{});
        fq0 = fq1; fq1 = fq2; fq2 = fq_new;
    }
    if (n & 1) {
        batch_upkeep(n_even);
        iteration_t(n_even, fq0, b1, a1, st0{}, mi2_t{});
    }
    clock_probe(p.clk, 2);
#ifdef SPARTA_TIMELINE
    if (p.clk != nullptr && blockIdx.y == 0 && tid == 0 && blockIdx.x < 1024) {
        p.clk[16 + 4 * 64 * 8 + 2 * blockIdx.x] = tl_t0;
        p.clk[16 + 4 * 64 * 8 + 2 * blockIdx.x + 1] = wall_clock64();
    }
#endif
}

// =====================================================================================================
// 16-bit (fp16 / bf16) storage, fp32 accumulation: vbs_spmm_h16_stream_kernel<KP, MI2, BF16>
// Same persistent design, step records, plans and epilogue as the fp32 stream kernel; what changes is the operand path.
// * A is re-laid-out ONCE at sparta_vbs_create: every step's slice is a dense row-major TM x KP chunk (k contiguous, rows
//   past the tile zero), the chunks of a tile back to back.  The 16-bit MFMA wants 8 consecutive k of one row per lane; the
//   reference's column-major blocks have k strided, and transposing 16-bit data on the way into LDS costs 8 ds_write_b16 per
//   16-byte load.  The host-side VBS (the boundary) keeps the reference's layout; only the device copy differs.
// * B must be column-major (k contiguous per column) with an even leading dimension: Bs[j][k], As[i][k], rows padded by
//   8 elements; a fragment is one ds_read_b128 (8 k) per operand and feeds v_mfma_f32_32x32x16_{f16,bf16}.
// * D = Bpanel^T . Atile^T as in the fp32 kernels (accumulator layout and epilogue are the same); an MFMA sums its k slots,
//   so any lane -> k assignment works as long as both operands use the same one (lane group g takes k = kb + 8g .. +7).
// A step moves half the bytes of the fp32 kernel and its MFMAs take 1/8 of the time: this kernel is bound by the load path
// (L2 / Infinity Cache -> LDS), not by MFMA.
// =====================================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// DEEP: four register sets instead of two -- loads run 5 steps ahead of the MFMAs (4 steps of panels in flight per workgroup
// instead of 2).  A 16-bit step has 64-128 cycles of MFMA per wave: what a step costs is how long its panel takes to arrive
// divided by the number of panels in flight, and the registers are there (2 waves per SIMD: 256 VGPRs each).
template <int KP, bool MI2, bool BF16, bool GATHERED, bool DEEP>
__global__ __launch_bounds__(kThreads, 2) void vbs_spmm_h16_stream_kernel(const StreamParams p) {
    constexpr int TN = kTN, TM = MI2 ? 64 : 32;
    constexpr int LDK = KP + 8;                          // elements per LDS row
    constexpr int BSZ = TN * LDK;                        // elements of the B panel image
    constexpr int STAGE = BSZ + 64 * LDK;                // elements per LDS stage
    constexpr int CPC = KP / 8;                          // 16-byte chunks per row / column
    constexpr int CPP = kThreads / CPC;                  // B columns covered by one load pass
    constexpr int NBL = TN / CPP;                        // B loads per thread and step (4 for KP 64, 2 for KP 32)
    constexpr int NAL = (TM * CPC + kThreads - 1) / kThreads;   // A loads per thread and step (2, 1, 1, 1)
    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 31, g = lane >> 5;
    const int n0 = blockIdx.y * TN;
    const int s_begin = p.worker_range[2 * blockIdx.x];
    const int n = p.worker_range[2 * blockIdx.x + 1] - s_begin;
    if (n <= 0) return;
    clock_probe(p.clk, 0);
#ifdef SPARTA_TIMELINE
    long long tl_t0 = 0;
    if (p.clk != nullptr && blockIdx.y == 0 && tid == 0 && blockIdx.x < 1024) tl_t0 = wall_clock64();
#endif
    float* ws = p.ws + (int64_t)blockIdx.y * p.ws_slab_stride;
    const uint16_t* A16 = reinterpret_cast<const uint16_t*>(p.A);
    const uint16_t* B16 = reinterpret_cast<const uint16_t*>(p.B);
    const uint16_t* Bt16 = reinterpret_cast<const uint16_t*>(p.B_tail);

    const int32_t* srec = reinterpret_cast<const int32_t*>(p.steps + s_begin);
    int vrec0 = srec[lane];
    int vrec1 = srec[64 + lane];
    int vnext = 0;
    // a macro over a free function, not a lambda: every closure between the loop body and vrec0 / vrec1 is one more level of
    // pointer indirection the optimiser has to peel before it can keep them in registers (three levels deep it gave up and
    // left one of them in memory: an LDS / scratch read behind a full wait in every step)
#define field(s, f) sk_field(vrec0, vrec1, (s), (f))
    enum { F_AOFF_LO = 0, F_AOFF_HI = 1, F_BROW = 2, F_H = 3, F_CROW = 4, F_FLAGS = 5, F_SLOT = 6, F_SHARD = 7 };

    const int bj0 = tid / CPC, bc = tid % CPC;           // B: column bj0 + CPP q, chunk bc (k = 8 bc .. 8 bc + 7)
    const int ac = tid % (TM * CPC);                     // A: chunk ac (+ 256 q) of the contiguous TM x KP slice (clamped: duplicates are harmless)
    const int64_t ld_t = (int64_t)p.w;                   // leading dimension of B_tail
    const uint32_t voffB = (uint32_t)((bc * 8 + bj0 * p.ldb) * 2), voffBt = (uint32_t)((bc * 8 + (n0 + bj0) * ld_t) * 2);
    const uint32_t qstepB = (uint32_t)(CPP * p.ldb * 2), qstepBt = (uint32_t)(CPP * ld_t * 2);
    const int64_t n0off = (int64_t)n0 * p.ldb;
    const uint32_t lwB = (uint32_t)((bj0 * LDK + bc * 8) * 2);
    const uint32_t lwA = (uint32_t)((BSZ + (ac / CPC) * LDK + (ac % CPC) * 8) * 2);
    const uint32_t lrB = (uint32_t)(((32 * wave + lm) * LDK + 8 * g) * 2);
    const uint32_t lrA = (uint32_t)((BSZ + lm * LDK + 8 * g) * 2);
    const uint32_t voffC = p.c_row_major ? (uint32_t)((lm * p.ldc + 32 * wave + 4 * g) * 4) : (uint32_t)((lm + (32 * wave + 4 * g) * p.ldc) * 4);
    char* const ldsb = reinterpret_cast<char*>(lds);

    u32x4 b0[NBL], a0[NAL], b1[NBL], a1[NAL];
    u32x4 b2[DEEP ? NBL : 1], a2[DEEP ? NAL : 1], b3[DEEP ? NBL : 1], a3[DEEP ? NAL : 1];     // DEEP: register sets 2 / 3
    constexpr int AHEAD = DEEP ? 5 : 3;                  // a register set written to LDS at step i is refilled with step i + AHEAD

    int64_t g_aoff = 0;
    uint32_t vo_cur = voffB;                             // see the fp32 kernel: every VALU instruction in the steady state costs an MFMA slot
    int32_t tail_prev = 0;
    auto issue_loads = [&](int s, u32x4 (&rb)[NBL], u32x4 (&ra)[NAL]) __attribute__((always_inline)) -> int32_t {
        const int32_t flags = field(s, F_FLAGS);
        if (flags & STEP_FIRST) g_aoff = (int64_t)(uint32_t)field(s, F_AOFF_LO) | ((int64_t)field(s, F_AOFF_HI) << 32);
        else g_aoff += (int64_t)TM * KP;                 // the slices of a tile are back to back
        const int32_t tail = (flags & STEP_TAIL) != 0;
        if (tail != tail_prev) {
            vo_cur = tail ? voffBt : voffB;
            asm volatile("" : "+v"(vo_cur));
            tail_prev = tail;
        }
        const int64_t gk0 = field(s, F_BROW);
        const uint16_t* bptr = tail ? Bt16 + gk0 : B16 + (GATHERED ? (int64_t)field(s, F_SHARD) * p.shard_stride : (int64_t)0) + gk0 + n0off;   // gathered: (slab, row inside the slab), see the fp32 kernel
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(bptr), 0, 0x7ffffff0, 0x00020000);
        const uint32_t qs = tail ? qstepBt : qstepB;
#pragma unroll
        for (int q = 0; q < NBL; q++) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo_cur, qs * q, 0);
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A16 + g_aoff), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int q = 0; q < NAL; q++) ra[q] = __builtin_amdgcn_raw_buffer_load_b128(rA, (uint32_t)ac * 16u, (uint32_t)(q * kThreads * 16), 2);
        return flags;
    };
    int32_t fq0 = 0, fq1 = 0, fq2 = 0, fq_new = 0;
    auto write_stage = [&](auto stage_tag, const u32x4 (&rb)[NBL], const u32x4 (&ra)[NAL]) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_tag)::value;
#pragma unroll
        for (int q = 0; q < NBL; q++) *reinterpret_cast<u32x4*>(ldsb + lwB + (ST * STAGE + q * CPP * LDK) * 2) = rb[q];
#pragma unroll
        for (int q = 0; q < NAL; q++) *reinterpret_cast<u32x4*>(ldsb + lwA + (ST * STAGE + q * (kThreads / CPC) * LDK) * 2) = ra[q];
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    auto mfma = [&](const u32x4& bf, const u32x4& af, f32x16& acc) __attribute__((always_inline)) {
        if constexpr (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf), __builtin_bit_cast(bf16x8, af), acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf), __builtin_bit_cast(f16x8, af), acc, 0, 0, 0);
    };

    auto iteration_t = [&](int i, int32_t flags, u32x4 (&wb)[NBL], u32x4 (&wa)[NAL], auto par_tag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        using nxt_t = std::integral_constant<int, 1 - PAR>;
#ifdef SPARTA_TIMELINE
        unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool tl_on = p.clk != nullptr && blockIdx.x == 8 && blockIdx.y == 0 && i >= TL_FIRST && i < TL_FIRST + TL_STEPS;
#endif
        TL_STAMP(0);
        write_stage(nxt_t{}, wb, wa);
        TL_STAMP(1);
        fq_new = issue_loads(i + AHEAD, wb, wa);
        TL_STAMP(2);
#pragma unroll
        for (int kb = 0; kb < KP; kb += 16) {
            const u32x4 bf = *reinterpret_cast<const u32x4*>(ldsb + lrB + (PAR * STAGE + kb) * 2);
            const u32x4 af0 = *reinterpret_cast<const u32x4*>(ldsb + lrA + (PAR * STAGE + kb) * 2);
            mfma(bf, af0, acc0);
            if constexpr (MI2) {
                const u32x4 af1 = *reinterpret_cast<const u32x4*>(ldsb + lrA + (PAR * STAGE + 32 * LDK + kb) * 2);
                mfma(bf, af1, acc1);
            }
        }
        TL_STAMP(3);
        TL_STAMP(4);
        if (flags & STEP_LAST) {
            if (flags & STEP_SPLIT) {
                const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(ws + (int64_t)field(i, F_SLOT) * SK_SLOT_FLOATS, 0, SK_SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc0[q]), rW, (uint32_t)tid * 4u, (uint32_t)(q * kThreads * 4), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc1[q]), rW, (uint32_t)tid * 4u, (uint32_t)((16 + q) * kThreads * 4), 0);
                }
            } else {
                const int mt = flags & 0xffff;
                const int64_t c_row = field(i, F_CROW);
                float* cbase = p.c_row_major ? p.C + c_row * p.ldc + n0 : p.C + c_row + (int64_t)n0 * p.ldc;
                const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, 0x7ffffff0, 0x00020000);
                const uint32_t jstep = p.c_row_major ? 4u : (uint32_t)p.ldc * 4u;
                const uint32_t mistep = p.c_row_major ? (uint32_t)p.ldc * 128u : 128u;
#pragma unroll
                for (int mi = 0; mi < (MI2 ? 2 : 1); mi++) {
                    if (mi * 32 + lm < mt) {
                        float v[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] = mi == 0 ? acc0[q] : acc1[q];
                        if (p.accumulate) {
                            uint32_t old[16];
#pragma unroll
                            for (int q = 0; q < 16; q++) old[q] = __builtin_amdgcn_raw_buffer_load_b32(rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 0);
#pragma unroll
                            for (int q = 0; q < 16; q++) v[q] += __uint_as_float(old[q]);
                        }
#pragma unroll
                        for (int q = 0; q < 16; q++)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; q++) { acc0[q] = 0.0f; acc1[q] = 0.0f; }
        }
        TL_STAMP(5);
        __syncthreads();
        TL_STAMP(6);
#ifdef SPARTA_TIMELINE
        if (tl_on) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) {
                long long* o = p.clk + 16 + ((int64_t)wave * TL_STEPS + (i - TL_FIRST)) * 8;
#pragma unroll
                for (int k = 0; k < 7; k++) o[k] = (long long)tl[k];
                o[7] = flags;
            }
        }
#endif
    };

    using st0 = std::integral_constant<int, 0>;
    using st1 = std::integral_constant<int, 1>;
    if constexpr (!DEEP) {
        fq0 = issue_loads(0, b0, a0);
        fq1 = issue_loads(1, b1, a1);
        write_stage(st0{}, b0, a0);
        fq2 = issue_loads(2, b0, a0);
        __syncthreads();
        // record batches: see the fp32 kernel.  At least 4 steps x (NBL + NAL >= 3) = 12 loads are issued between request and touch.
        auto batch_upkeep = [&](int i) __attribute__((always_inline)) {
            if ((i & 7) == 0 && i > 0) {
                const int32_t* nb = srec + (int64_t)((i >> 3) + 1) * 64 + lane;
                asm volatile("global_load_dword %0, %1, off" : "=&v"(vnext) : "v"(nb) : "memory");
            }
            if ((i & 7) == 4 && i > 4) {
                asm volatile("s_waitcnt vmcnt(3)" : "+v"(vnext) : : "memory");
                // two selects, not "if (odd) vrec1 = vnext; else vrec0 = vnext;": the compiler merges the two stores of that form into one store
                // through a selected POINTER, which pins vrec0 / vrec1 in scratch memory -- every step then reloads them behind vmcnt(0)
                const bool odd = (((i >> 3) + 1) & 1) != 0;
                vrec1 = odd ? vnext : vrec1;
                vrec0 = odd ? vrec0 : vnext;
            }
        };
        const int n_even = n & ~1;
        for (int i = 0; i < n_even; i += 2) {
            batch_upkeep(i);
            iteration_t(i, fq0, b1, a1, st0{});
            fq0 = fq1; fq1 = fq2; fq2 = fq_new;
            iteration_t(i + 1, fq0, b0, a0, st1{});
            fq0 = fq1; fq1 = fq2; fq2 = fq_new;
        }
        if (n & 1) {
            batch_upkeep(n_even);
            iteration_t(n_even, fq0, b1, a1, st0{});
        }
    } else {
        // Step s lives in register set s & 3 and LDS stage s & 1.  Iteration i computes step i, writes step i + 1 (set (i + 1) & 3)
        // into the other stage and refills that set with step i + 5: steps i + 2 .. i + 5 are in flight under step i.
        int32_t fq3 = 0, fq4 = 0;
        fq0 = issue_loads(0, b0, a0);
        fq1 = issue_loads(1, b1, a1);
        write_stage(st0{}, b0, a0);
        fq2 = issue_loads(2, b2, a2);
        fq3 = issue_loads(3, b3, a3);
        fq4 = issue_loads(4, b0, a0);
        __syncthreads();
        // Record batches (8 steps each; vrec0 / vrec1 hold batches k, k + 1): step 8k + 3 is the first to look into batch k + 1
        // (its refill is step 8k + 8), so batch k + 1 is requested at step 8k - 2 and touched at step 8k + 2; the register it
        // replaces (batch k - 1) is dead from step 8k on.  The four steps in between issue 4 x (NBL + NAL >= 3) >= 12 loads and
        // memory returns in order: vmcnt(12) at the touch is a wait the pipeline has already paid.
        auto upkeep_request = [&](int i) __attribute__((always_inline)) {
            if ((i & 7) == 6) {
                const int32_t* nb = srec + (int64_t)((i >> 3) + 2) * 64 + lane;
                asm volatile("global_load_dword %0, %1, off" : "=&v"(vnext) : "v"(nb) : "memory");
            }
        };
        auto upkeep_touch = [&](int i) __attribute__((always_inline)) {
            if ((i & 7) == 2 && i > 2) {
                asm volatile("s_waitcnt vmcnt(12)" : "+v"(vnext) : : "memory");
                // two selects, not "if (odd) vrec1 = vnext; else vrec0 = vnext;": the compiler merges the two stores of that form into one store
                // through a selected POINTER, which pins vrec0 / vrec1 in scratch memory -- every step then reloads them behind vmcnt(0)
                const bool odd = (((i >> 3) + 1) & 1) != 0;
                vrec1 = odd ? vnext : vrec1;
                vrec0 = odd ? vrec0 : vnext;
            }
        };
        auto rotate = [&]() __attribute__((always_inline)) { fq0 = fq1; fq1 = fq2; fq2 = fq3; fq3 = fq4; fq4 = fq_new; };
        const int n4 = n & ~3;
        for (int i = 0; i < n4; i += 4) {
            iteration_t(i, fq0, b1, a1, st0{});
            rotate();
            iteration_t(i + 1, fq0, b2, a2, st1{});
            rotate();
            upkeep_request(i + 2);
            upkeep_touch(i + 2);
            iteration_t(i + 2, fq0, b3, a3, st0{});
            rotate();
            iteration_t(i + 3, fq0, b0, a0, st1{});
            rotate();
        }
        // the last n & 3 steps, peeled (see the fp32 kernel on why not a break inside the loop)
        if (n & 3) {
            iteration_t(n4, fq0, b1, a1, st0{});
            rotate();
            if ((n & 3) > 1) {
                iteration_t(n4 + 1, fq0, b2, a2, st1{});
                rotate();
                if ((n & 3) > 2) {
                    upkeep_touch(n4 + 2);
                    iteration_t(n4 + 2, fq0, b3, a3, st0{});
                }
            }
        }
    }
    clock_probe(p.clk, 2);
#ifdef SPARTA_TIMELINE
    if (p.clk != nullptr && blockIdx.y == 0 && tid == 0 && blockIdx.x < 1024) {
        p.clk[16 + 4 * 64 * 8 + 2 * blockIdx.x] = tl_t0;
        p.clk[16 + 4 * 64 * 8 + 2 * blockIdx.x + 1] = wall_clock64();
    }
#endif
}

#undef field

// =====================================================================================================
// 16-bit storage, operands straight from global memory into the MFMA registers: vbs_spmm_h16_direct_kernel<KP, MI2, BF16, GATHERED>
// A 16-bit step has 64-256 cycles of MFMA per wave; in the LDS-staged kernel above what a step costs is its bookkeeping: the
// register -> LDS -> register round trip, one workgroup barrier and the four waves waiting for each other (measured with the
// in-kernel timeline: ~1300 cycles per step with every byte cache-hot).  The packed A slices (row-major, k contiguous) and a
// column-major B (k contiguous) already ARE the per-lane operand layout of v_mfma_f32_32x32x16: lane (lm, g) needs 8 consecutive
// k of row / column lm, one 16-byte load.  So here nothing is staged: every wave loads its own fragments (its 32 columns of B,
// and the A slice, which the four waves of a workgroup share through the L1), keeps LOOK steps of them in flight in registers and
// never synchronises with the other waves -- no LDS, no barrier.  Same plans, step records, accumulator layout, epilogue and
// fix-up as the other stream kernels (a wave's accumulator image is identical, so the split-tile workspace is too).
// =====================================================================================================
template <int KP, bool MI2, bool BF16, bool GATHERED>
__global__ __launch_bounds__(kThreads, 2) void vbs_spmm_h16_direct_kernel(const StreamParams p) {
    constexpr int TN = kTN, TM = MI2 ? 64 : 32;
    constexpr int NK = KP / 16;                          // MFMAs (k groups of 16) per step and 32-row tile
    constexpr int NA = MI2 ? 2 : 1;
    constexpr int LOOK = 4;                              // register sets = steps in flight

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 31, g = lane >> 5;
    const int n0 = blockIdx.y * TN;
    const int s_begin = p.worker_range[2 * blockIdx.x];
    const int n = p.worker_range[2 * blockIdx.x + 1] - s_begin;
    if (n <= 0) return;
    clock_probe(p.clk, 0);
    float* ws = p.ws + (int64_t)blockIdx.y * p.ws_slab_stride;
    const uint16_t* A16 = reinterpret_cast<const uint16_t*>(p.A);
    const uint16_t* B16 = reinterpret_cast<const uint16_t*>(p.B);
    const uint16_t* Bt16 = reinterpret_cast<const uint16_t*>(p.B_tail);

    const int32_t* srec = reinterpret_cast<const int32_t*>(p.steps + s_begin);
    int vrec0 = srec[lane];
    int vrec1 = srec[64 + lane];
    int vnext = 0;
#define field(s, f) sk_field(vrec0, vrec1, (s), (f))
    enum { F_AOFF_LO = 0, F_AOFF_HI = 1, F_BROW = 2, F_H = 3, F_CROW = 4, F_FLAGS = 5, F_SLOT = 6, F_SHARD = 7 };

    const int64_t ld_t = (int64_t)p.w;                   // leading dimension of B_tail
    const uint32_t voffB = (uint32_t)(((32 * wave + lm) * p.ldb + 8 * g) * 2);
    const uint32_t voffBt = (uint32_t)(((n0 + 32 * wave + lm) * ld_t + 8 * g) * 2);
    const uint32_t voffA = (uint32_t)((lm * KP + 8 * g) * 2);
    const int64_t n0off = (int64_t)n0 * p.ldb;
    const uint32_t voffC = p.c_row_major ? (uint32_t)((lm * p.ldc + 32 * wave + 4 * g) * 4) : (uint32_t)((lm + (32 * wave + 4 * g) * p.ldc) * 4);

    struct Set { u32x4 b[NK]; u32x4 a[NA][NK]; };
    Set r0, r1, r2, r3;

    int64_t g_aoff = 0;
    uint32_t vo_cur = voffB;
    int32_t tail_prev = 0;
    auto issue_loads = [&](int s, Set& r) __attribute__((always_inline)) -> int32_t {
        const int32_t flags = field(s, F_FLAGS);
        if (flags & STEP_FIRST) g_aoff = (int64_t)(uint32_t)field(s, F_AOFF_LO) | ((int64_t)field(s, F_AOFF_HI) << 32);
        else g_aoff += (int64_t)TM * KP;                 // the slices of a tile are back to back
        const int32_t tail = (flags & STEP_TAIL) != 0;
        if (tail != tail_prev) {
            vo_cur = tail ? voffBt : voffB;
            asm volatile("" : "+v"(vo_cur));
            tail_prev = tail;
        }
        const int64_t gk0 = field(s, F_BROW);
        const uint16_t* bptr = tail ? Bt16 + gk0 : B16 + (GATHERED ? (int64_t)field(s, F_SHARD) * p.shard_stride : (int64_t)0) + gk0 + n0off;
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(bptr), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int q = 0; q < NK; q++) r.b[q] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo_cur, (uint32_t)(q * 32), 0);
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A16 + g_aoff), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int mi = 0; mi < NA; mi++)
#pragma unroll
            for (int q = 0; q < NK; q++) r.a[mi][q] = __builtin_amdgcn_raw_buffer_load_b128(rA, voffA, (uint32_t)(mi * 32 * KP * 2 + q * 32), 0);
        return flags;
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    auto mfma = [&](const u32x4& bf, const u32x4& af, f32x16& acc) __attribute__((always_inline)) {
        if constexpr (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf), __builtin_bit_cast(bf16x8, af), acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf), __builtin_bit_cast(f16x8, af), acc, 0, 0, 0);
    };

    int32_t fq0 = 0, fq1 = 0, fq2 = 0, fq3 = 0, fq_new = 0;
    // one step: multiply from register set r (step i), run the tile epilogue if it ends here, refill r with step i + LOOK
    auto iteration_t = [&](int i, int32_t flags, Set& r) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NK; q++) {
            mfma(r.b[q], r.a[0][q], acc0);
            if constexpr (MI2) mfma(r.b[q], r.a[1][q], acc1);
        }
        if (flags & STEP_LAST) {
            if (flags & STEP_SPLIT) {
                const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(ws + (int64_t)field(i, F_SLOT) * SK_SLOT_FLOATS, 0, SK_SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc0[q]), rW, (uint32_t)tid * 4u, (uint32_t)(q * kThreads * 4), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc1[q]), rW, (uint32_t)tid * 4u, (uint32_t)((16 + q) * kThreads * 4), 0);
                }
            } else {
                const int mt = flags & 0xffff;
                const int64_t c_row = field(i, F_CROW);
                float* cbase = p.c_row_major ? p.C + c_row * p.ldc + n0 : p.C + c_row + (int64_t)n0 * p.ldc;
                const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, 0x7ffffff0, 0x00020000);
                const uint32_t jstep = p.c_row_major ? 4u : (uint32_t)p.ldc * 4u;
                const uint32_t mistep = p.c_row_major ? (uint32_t)p.ldc * 128u : 128u;
#pragma unroll
                for (int mi = 0; mi < (MI2 ? 2 : 1); mi++) {
                    if (mi * 32 + lm < mt) {
                        float v[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] = mi == 0 ? acc0[q] : acc1[q];
                        if (p.accumulate) {
                            uint32_t old[16];
#pragma unroll
                            for (int q = 0; q < 16; q++) old[q] = __builtin_amdgcn_raw_buffer_load_b32(rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 0);
#pragma unroll
                            for (int q = 0; q < 16; q++) v[q] += __uint_as_float(old[q]);
                        }
#pragma unroll
                        for (int q = 0; q < 16; q++)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; q++) { acc0[q] = 0.0f; acc1[q] = 0.0f; }
        }
        fq_new = issue_loads(i + LOOK, r);
    };

    fq0 = issue_loads(0, r0);
    fq1 = issue_loads(1, r1);
    fq2 = issue_loads(2, r2);
    fq3 = issue_loads(3, r3);
    // Record batches (8 steps each; vrec0 / vrec1 hold batches k, k + 1): step 8k + 4 is the first to look into batch k + 1 (its
    // refill is step 8k + 8).  Batch k + 1 is requested at step 8k and touched at step 8k + 4; the register it replaces (batch
    // k - 1) is dead from step 8k on.  The four steps in between issue 4 x (>= 4) loads and memory returns in order: vmcnt(12)
    // at the touch is a wait the pipeline pays anyway.
    auto batch_upkeep = [&](int i) __attribute__((always_inline)) {
        if ((i & 7) == 0 && i > 0) {
            const int32_t* nb = srec + (int64_t)((i >> 3) + 1) * 64 + lane;
            asm volatile("global_load_dword %0, %1, off" : "=&v"(vnext) : "v"(nb) : "memory");
        }
        if ((i & 7) == 4 && i > 4) {
            asm volatile("s_waitcnt vmcnt(12)" : "+v"(vnext) : : "memory");
            const bool odd = (((i >> 3) + 1) & 1) != 0;
            vrec1 = odd ? vnext : vrec1;
            vrec0 = odd ? vrec0 : vnext;
        }
    };
    auto rotate = [&]() __attribute__((always_inline)) { fq0 = fq1; fq1 = fq2; fq2 = fq3; fq3 = fq_new; };
    const int n4 = n & ~3;
    for (int i = 0; i < n4; i += 4) {
        batch_upkeep(i);
        iteration_t(i, fq0, r0);
        rotate();
        iteration_t(i + 1, fq0, r1);
        rotate();
        iteration_t(i + 2, fq0, r2);
        rotate();
        iteration_t(i + 3, fq0, r3);
        rotate();
    }
    if (n & 3) {                                         // the last n & 3 steps, peeled
        batch_upkeep(n4);
        iteration_t(n4, fq0, r0);
        rotate();
        if ((n & 3) > 1) {
            iteration_t(n4 + 1, fq0, r1);
            rotate();
            if ((n & 3) > 2) iteration_t(n4 + 2, fq0, r2);
        }
    }
    clock_probe(p.clk, 2);
#undef field
}

// zero-padded copy of the rows of a 16-bit column-major B that face the last (partial) block column: B_tail[k + w j], k < w
__global__ __launch_bounds__(kThreads) void vbs_tail_copy_h16_kernel(const uint16_t* B, int64_t ldb, int64_t row0, int64_t cols, int w, int N,
                                                                     uint16_t* B_tail) {
    const int64_t total = (int64_t)w * N;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t k = e % w, j = e / w;
        B_tail[e] = (row0 + k < cols) ? B[row0 + k + j * ldb] : (uint16_t)0;
    }
}

// fp32 -> fp16 / bf16 (round to nearest even), column by column: src ld_in, dst ld_out (host-pointer convenience path)
template <bool BF16>
__global__ __launch_bounds__(kThreads) void vbs_convert_h16_kernel(const float* src, int64_t ld_in, int64_t rows, int64_t n_cols, uint16_t* dst,
                                                                   int64_t ld_out) {
    const int64_t total = rows * n_cols;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t k = e % rows, j = e / rows;
        const float v = src[k + j * ld_in];
        uint16_t o;
        if constexpr (BF16) {
            uint32_t u = __float_as_uint(v);
            if ((u & 0x7fffffffu) > 0x7f800000u) o = (uint16_t)((u >> 16) | 0x40);          // NaN stays NaN
            else o = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        } else {
            const _Float16 h = (_Float16)v;
            o = __builtin_bit_cast(uint16_t, h);
        }
        dst[k + j * ld_out] = o;
    }
}

// A tile that dominates the plan (a hub block-row) is split over hundreds of workers; adding its partial images one after the
// other in ONE workgroup is a latency-bound chain (measured: 512 images, 350 us).  First stage for such plans: blockIdx.z = group
// of kFixGroup consecutive images, summed in order into the group's first image; the fix-up kernel then adds the group leaders
// (stride kFixGroup).  Fixed grouping -> the result is reproducible run to run.
constexpr int kFixGroup = 16;
__global__ __launch_bounds__(kThreads) void vbs_spmm_f32_fixup_group_kernel(const FixRec* fix, const int32_t* big, const int32_t* fix_slots,
                                                                            float* ws_all, int64_t ws_slab_stride) {
    const FixRec fr = fix[big[blockIdx.x]];          // only the tiles with more than 2 * kFixGroup images come here
    const int s0 = blockIdx.z * kFixGroup;
    if (s0 + 1 >= fr.n_slots) return;                 // no such group, or a group of one image
    float* ws = ws_all + (int64_t)blockIdx.y * ws_slab_stride;
    const int tid = threadIdx.x;
    const int s1 = s0 + kFixGroup < fr.n_slots ? s0 + kFixGroup : fr.n_slots;
    float* lead = ws + (int64_t)fix_slots[fr.slot_begin + s0] * SK_SLOT_FLOATS + tid;
    float acc[32];
#pragma unroll
    for (int q = 0; q < 32; q++) acc[q] = lead[q * kThreads];
    int s = s0 + 1;
    for (; s + 2 <= s1; s += 2) {
        const float* i0 = ws + (int64_t)fix_slots[fr.slot_begin + s] * SK_SLOT_FLOATS + tid;
        const float* i1 = ws + (int64_t)fix_slots[fr.slot_begin + s + 1] * SK_SLOT_FLOATS + tid;
#pragma unroll
        for (int q = 0; q < 32; q++) { const float a0 = i0[q * kThreads], a1 = i1[q * kThreads]; acc[q] += a0; acc[q] += a1; }
    }
    for (; s < s1; s++) {
        const float* i0 = ws + (int64_t)fix_slots[fr.slot_begin + s] * SK_SLOT_FLOATS + tid;
#pragma unroll
        for (int q = 0; q < 32; q++) acc[q] += i0[q * kThreads];
    }
#pragma unroll
    for (int q = 0; q < 32; q++) lead[q * kThreads] = acc[q];
}

// adds the partial images of every split tile (fixed order: worker order; `stride` > 1 after the group stage) and writes the tile
__global__ __launch_bounds__(kThreads) void vbs_spmm_f32_fixup_kernel(const FixRec* fix, const int32_t* fix_slots, const float* ws_all,
                                                                      int64_t ws_slab_stride, float* C, int64_t ldc, int c_row_major,
                                                                      int accumulate) {
    const FixRec fr = fix[blockIdx.x];
    if (accumulate && fr.n_slots == 0) return;   // a block-row without blocks adds nothing to C (vbr.cpp:340-368 never touches its rows)
    const float* ws = ws_all + (int64_t)blockIdx.y * ws_slab_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 31, g = lane >> 5;
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; q++) { acc0[q] = 0.0f; acc1[q] = 0.0f; }
    // the partial images are added in slot order, four at a time (four independent loads in flight per element: a tile that
    // dominates the plan is split over hundreds of workers, and one dependent chain of loads per element is latency-bound)
    const int stride = fr.n_slots > 2 * kFixGroup ? kFixGroup : 1;   // such tiles went through the group stage: add the group leaders
    int s = 0;
    for (; s + 3 * stride < fr.n_slots; s += 4 * stride) {
        const float* i0 = ws + (int64_t)fix_slots[fr.slot_begin + s] * SK_SLOT_FLOATS + tid;
        const float* i1 = ws + (int64_t)fix_slots[fr.slot_begin + s + stride] * SK_SLOT_FLOATS + tid;
        const float* i2 = ws + (int64_t)fix_slots[fr.slot_begin + s + 2 * stride] * SK_SLOT_FLOATS + tid;
        const float* i3 = ws + (int64_t)fix_slots[fr.slot_begin + s + 3 * stride] * SK_SLOT_FLOATS + tid;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const float a0 = i0[q * kThreads], a1 = i1[q * kThreads], a2 = i2[q * kThreads], a3 = i3[q * kThreads];
            const float b0 = i0[(16 + q) * kThreads], b1 = i1[(16 + q) * kThreads], b2 = i2[(16 + q) * kThreads], b3 = i3[(16 + q) * kThreads];
            acc0[q] += a0; acc0[q] += a1; acc0[q] += a2; acc0[q] += a3;
            acc1[q] += b0; acc1[q] += b1; acc1[q] += b2; acc1[q] += b3;
        }
    }
    for (; s < fr.n_slots; s += stride) {
        const float* img = ws + (int64_t)fix_slots[fr.slot_begin + s] * SK_SLOT_FLOATS + tid;
#pragma unroll
        for (int q = 0; q < 16; q++) { acc0[q] += img[q * kThreads]; acc1[q] += img[(16 + q) * kThreads]; }
    }
    sk_store_tile(acc0, acc1, fr.mt, fr.c_row, blockIdx.y * kTN + 32 * wave, C, ldc, c_row_major, accumulate, lm, g);
}

// zero-padded copy of the rows of B that face the last (partial) block column: B_tail[k][j], k < w
__global__ __launch_bounds__(kThreads) void vbs_tail_copy_kernel(const float* B, int64_t ldb, int b_row_major, int64_t row0, int64_t cols,
                                                                 int w, int N, float* B_tail) {
    const int64_t total = (int64_t)w * N;
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
        if (b_row_major) {
            const int64_t k = idx / N, j = idx % N;
            B_tail[idx] = row0 + k < cols ? B[(row0 + k) * ldb + j] : 0.0f;
        } else {
            const int64_t j = idx / w, k = idx % w;
            B_tail[idx] = row0 + k < cols ? B[row0 + k + j * ldb] : 0.0f;
        }
    }
}

// ---- sparse-row path: block-rows whose blocks are almost empty -----------------------------------------------------
// A block-row of a clustered power-law matrix is typically 1-4 rows tall with 1-2 nonzeros per w-wide block: as a dense
// tile it executes 32 x w x N multiply-adds and streams a w x N panel of B for a handful of useful products (measured on
// R-MAT 2^20, fill 2 %: 4 TFLOP/s "executed", 0.09 TFLOP/s useful).  Such block-rows are taken out of the MFMA plans at
// create time and kept as rows of (column, value) pairs -- the nonzeros of their blocks in the reference's order (blocks
// ascending, k ascending) -- and multiplied the way the bytes want it: one wave per row, lanes across the columns of C, per
// nonzero ONE contiguous N-float row of B (row-major; a column-major or gathered B is transposed once per call: 2 x |B|
// bytes) and one FMA per element.  HBM/L2-bound: N * 4 bytes per nonzero.  This is the "wavefront-level partial sums for
// thin / ragged blocks" leg of the path; exact zeros of A are skipped (0 * inf of the reference's dense loop is not
// reproduced: finite B is the contract, as for its padded columns).
#ifndef SP_BATCH
#define SP_BATCH 16     /* rows of B a wave keeps in flight (measured 8 -> 16: +1..10 %) */
#endif
template <int VEC> struct SpVec;
template <> struct SpVec<1> { typedef float T; };
template <> struct SpVec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct SpVec<4> { typedef float T __attribute__((ext_vector_type(4))); };

template <int VEC> struct SpRaw16;        // VEC 16-bit values as loaded
template <> struct SpRaw16<1> { typedef unsigned short T; };
template <> struct SpRaw16<2> { typedef unsigned short T __attribute__((ext_vector_type(2))); };
template <> struct SpRaw16<4> { typedef unsigned short T __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ float sp_widen(unsigned short u, bool bf16) {
    return bf16 ? __builtin_bit_cast(float, (uint32_t)u << 16) : (float)__builtin_bit_cast(_Float16, u);
}

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
    float* out;                // row-major out: C itself (ld = ldc, row = crow) or the scratch (ld = N, row = ordinal)
    int64_t ldo;
    int32_t out_is_c, accumulate, N;
};

template <int VEC, int BK>
__device__ __forceinline__ typename SpVec<VEC>::T sparse_row_partial(const SparseParams& p, int64_t p0, int64_t p1, int n0, int lane) {
    typedef typename SpVec<VEC>::T V;
    typedef typename std::conditional<BK == 0, float, unsigned short>::type E;
    typedef typename std::conditional<BK == 0, V, typename SpRaw16<VEC>::T>::type L;
    V acc = (V)(0.0f);
    const bool in = VEC > 1 || n0 < p.N;
    const E* Bl = (const E*)p.B + (in ? n0 : 0);
    for (int64_t q = p0; q < p1; q += SP_BATCH) {
        const int n = (int)(p1 - q < SP_BATCH ? p1 - q : SP_BATCH);            // wave-uniform
        int cl = 0;
        float vl = 0.0f;
        if (lane < n) { cl = p.col[q + lane]; vl = p.val[q + lane]; }
        L b[SP_BATCH];
        if (VEC == 1 && p.b_col_stride > 0) {                    // column-major B in place (few sparse rows: cheaper than transposing all of B)
            const E* Bc = (const E*)p.B + (in ? (int64_t)n0 * p.b_col_stride : 0);
#pragma unroll
            for (int t = 0; t < SP_BATCH; t++) {
                const int64_t c = __builtin_amdgcn_readlane(cl, t);
                const int64_t off = p.shard_rows > 0 ? (c / p.shard_rows) * p.shard_stride + c % p.shard_rows : c;
                b[t] = *reinterpret_cast<const L*>(Bc + off);
            }
        } else {
#pragma unroll
            for (int t = 0; t < SP_BATCH; t++) {
                const int c = __builtin_amdgcn_readlane(cl, t);  // lanes >= n hold column 0: a valid row, never used
                b[t] = *reinterpret_cast<const L*>(Bl + (int64_t)c * p.ldb);
            }
        }
#pragma unroll
        for (int t = 0; t < SP_BATCH; t++) {
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vl), t));
            if (t < n) {
                if constexpr (BK == 0) acc += v * b[t];
                else if constexpr (VEC == 1) acc += v * sp_widen(b[t], BK == 2);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; e++) acc[e] += v * sp_widen(b[t][e], BK == 2);
                }
            }
        }
    }
    return acc;
}

template <int VEC>
__device__ __forceinline__ void sparse_row_store(const SparseParams& p, int ord, typename SpVec<VEC>::T acc, int n0) {
    typedef typename SpVec<VEC>::T V;
    if (VEC == 1 && n0 >= p.N) return;
    float* o = p.out + (int64_t)(p.out_is_c ? p.crow[ord] : ord) * p.ldo + n0;
    if (p.out_is_c && p.accumulate) acc += *reinterpret_cast<const V*>(o);
    *reinterpret_cast<V*>(o) = acc;
}

// rows of ordinary length: one wave per row, 4 rows per workgroup; blockIdx.y walks N in chunks of 64 * VEC columns
template <int VEC, int BK>
__global__ __launch_bounds__(kThreads) void sparse_rows_kernel(SparseParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= p.n_list) return;
    const int ord = p.list[slot];
    const int n0 = (blockIdx.y * 64 + lane) * VEC;
    const int64_t p0 = p.rowptr[ord], p1 = p.rowptr[ord + 1];
    sparse_row_store<VEC>(p, ord, sparse_row_partial<VEC, BK>(p, p0, p1, n0, lane), n0);
}

// long rows (hubs): cut into segments of <= kSpSeg nonzeros, one wave per segment writes a partial row; a second launch adds
// the partial rows of every long row in segment order (deterministic) and stores the row
struct SpSegRec { int64_t p0; int32_t cnt, pad; };
struct SpLongRec { int32_t ord, seg_begin, n_seg, pad; };

template <int VEC, int BK>
__global__ __launch_bounds__(kThreads) void sparse_segments_kernel(SparseParams p, const SpSegRec* segs, int32_t n_segs, float* part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= n_segs) return;
    const SpSegRec sg = segs[slot];
    const int n0 = (blockIdx.y * 64 + lane) * VEC;
    typename SpVec<VEC>::T acc = sparse_row_partial<VEC, BK>(p, sg.p0, sg.p0 + sg.cnt, n0, lane);
    if (VEC == 1 && n0 >= p.N) return;
    *reinterpret_cast<typename SpVec<VEC>::T*>(part + (int64_t)slot * p.N + n0) = acc;
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_reduce_kernel(SparseParams p, const SpLongRec* rows, int32_t n_rows, const float* part) {
    typedef typename SpVec<VEC>::T V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= n_rows) return;
    const SpLongRec r = rows[slot];
    const int n0 = (blockIdx.y * 64 + lane) * VEC;
    if (VEC == 1 && n0 >= p.N) return;
    V acc = (V)(0.0f);
    for (int sgi = 0; sgi < r.n_seg; sgi++) acc += *reinterpret_cast<const V*>(part + (int64_t)(r.seg_begin + sgi) * p.N + n0);
    sparse_row_store<VEC>(p, r.ord, acc, n0);
}

// B (column-major, ld = ldb, or the gathered slabs) -> row-major rows x N (ld = N); 64 x 64 tiles through LDS: a wave reads 64
// consecutive rows of one column (256 contiguous bytes) and writes 64 consecutive columns of one row (256 contiguous bytes)
template <class E>
__global__ __launch_bounds__(kThreads) void b_to_row_major_kernel(const E* __restrict__ B, int64_t ldb, int64_t shard_rows, int64_t shard_stride,
                                                                  int64_t rows, int N, E* __restrict__ out) {
    __shared__ E tile[64][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 1-D grid, column tile fastest: the workgroups that run together complete whole rows of the row-major copy (and read the same
    // 64 rows of every column tile) instead of each touching one 128 / 256-byte piece of 64 rows that the next piece follows much later
    const int n_ct = (N + 63) / 64;
    const int64_t r0 = (int64_t)(blockIdx.x / n_ct) * 64;
    const int n0 = (int)(blockIdx.x % n_ct) * 64;
    const int64_t r = r0 + lane;
    // All 16 loads of a lane are issued before the first one is used: addresses are clamped into the matrix instead of guarded (a
    // guarded load in a rolled loop was one load in flight per wave - 2.5 TB/s whatever the access pattern), the guard is applied
    // to the value.
    const int64_t rc = r < rows ? r : rows - 1;
    const int64_t roff = shard_rows > 0 ? (rc / shard_rows) * shard_stride + (rc % shard_rows) : rc;   // shard_rows % 64 need not hold: per lane
    E v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {                               // read: lanes along the rows (contiguous in a column)
        const int n = n0 + wave + 4 * q;
        v[q] = B[roff + (int64_t)(n < N ? n : N - 1) * ldb];
    }
#pragma unroll
    for (int q = 0; q < 16; q++) tile[wave + 4 * q][lane] = (r < rows && n0 + wave + 4 * q < N) ? v[q] : (E)0;
    __syncthreads();
    const int n = n0 + lane;
#pragma unroll
    for (int q = 0; q < 16; q++) {                               // write: lanes along the columns (contiguous in a row)
        const int j = wave + 4 * q;
        const int64_t rr = r0 + j;
        if (rr < rows && n < N) out[rr * N + n] = tile[lane][j];
    }
}

// scratch (row-major, one row per sparse row) -> the column-major C rows they belong to
__global__ __launch_bounds__(kThreads) void sparse_c_scatter_kernel(const float* __restrict__ src, const int32_t* __restrict__ crow, int64_t n_rows, int N,
                                                                    float* __restrict__ C, int64_t ldc, int accumulate) {
    __shared__ float tile[64][65];                               // 64 sparse rows x 64 columns
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_ct = (N + 63) / 64;                              // 1-D grid, column tile fastest (see b_to_row_major_kernel)
    const int64_t t0 = (int64_t)(blockIdx.x / n_ct) * 64;
    const int n0 = (int)(blockIdx.x % n_ct) * 64;
    const int64_t t = t0 + lane;
    const int32_t r = crow[t < n_rows ? t : n_rows - 1];         // consecutive sparse rows are mostly consecutive rows of C (requested with the tile's loads)
    {
        const int n = n0 + lane;
        const int nc = n < N ? n : N - 1;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {                           // read: lanes along the columns of one scratch row; 16 loads in flight (clamped, not guarded)
            const int64_t t = t0 + wave + 4 * q;
            v[q] = src[(t < n_rows ? t : n_rows - 1) * N + nc];
        }
#pragma unroll
        for (int q = 0; q < 16; q++) tile[wave + 4 * q][lane] = (t0 + wave + 4 * q < n_rows && n < N) ? v[q] : 0.0f;
    }
    __syncthreads();
    if (t >= n_rows) return;
    if (accumulate) {
        float old[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int n = n0 + wave + 4 * q;
            old[q] = C[r + (int64_t)(n < N ? n : N - 1) * ldc];
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int n = n0 + wave + 4 * q;
            if (n < N) C[r + (int64_t)n * ldc] = old[q] + tile[lane][wave + 4 * q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) {                           // write: lanes along the rows of one column of C
            const int n = n0 + wave + 4 * q;
            if (n < N) C[r + (int64_t)n * ldc] = tile[lane][wave + 4 * q];
        }
    }
}

// ---- row-block pack (multi-GPU exchange of only the needed rows of B) ---------------------------------
// dst chunk i <- src chunk ids[i]; a chunk is one w x N tile of B (block_bytes, a multiple of 16).  One workgroup per
// chunk and grid.y slice; 16-byte loads / stores, fully coalesced: HBM-bound, bytes = 2 x n_blocks x block_bytes.
__global__ __launch_bounds__(kThreads) void pack_blocks_kernel(const u32x4* __restrict__ src, const int32_t* __restrict__ ids,
                                                               u32x4* __restrict__ dst, int64_t block_vec) {
    const int64_t from = (int64_t)ids[blockIdx.x] * block_vec, to = (int64_t)blockIdx.x * block_vec;
    const int64_t stride = (int64_t)gridDim.y * kThreads;
    // four loads of a thread in flight (clamped addresses, guarded stores): the pack is the head of the exchange's critical path and a
    // 16 KB tile is four rounds of the workgroup - one round at a time is four memory latencies in a row
    for (int64_t i = (int64_t)blockIdx.y * kThreads + threadIdx.x; i < block_vec; i += 4 * stride) {
        u32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = __builtin_nontemporal_load(src + from + std::min<int64_t>(i + q * stride, block_vec - 1));
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (i + q * stride < block_vec) dst[to + i + q * stride] = v[q];
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

// Zero fill of a long run of rows of C (block-rows without blocks under accumulate = 0).  An R-MAT matrix has hundreds of thousands
// of empty rows; clustering puts them into ONE block-row, i.e. one contiguous row range of the permuted C.  As 64-row fix-up tiles
// that range is written in 128 / 256-byte pieces (2.4 TB/s measured); here every line (a column of a column-major C: `nrows`
// contiguous floats; a row of a row-major C) is streamed with 16-byte stores.
__global__ __launch_bounds__(kThreads) void vbs_zero_rows_kernel(float* C, int64_t ldc, int c_row_major, int64_t row0, int64_t nrows, int N) {
    const int64_t n_lines = c_row_major ? nrows : (int64_t)N;
    const int64_t line_len = c_row_major ? (int64_t)N : nrows;
    float* base = c_row_major ? C + row0 * ldc : C + row0;
    for (int64_t line = blockIdx.y; line < n_lines; line += gridDim.y) {
        float* p = base + line * ldc;
        const int64_t head = std::min<int64_t>(line_len, (int64_t)(((16u - (uint32_t)((uintptr_t)p & 15u)) & 15u) >> 2));
        const int64_t body4 = (line_len - head) >> 2, tail = (line_len - head) & 3;
        f32x4* q = reinterpret_cast<f32x4*>(p + head);
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < body4; e += (int64_t)gridDim.x * kThreads) q[e] = z;
        if (blockIdx.x == 0) {
            if ((int64_t)threadIdx.x < head) p[threadIdx.x] = 0.0f;
            if ((int64_t)threadIdx.x < tail) p[head + 4 * body4 + threadIdx.x] = 0.0f;
        }
    }
}

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
    float* d_A = nullptr;                    // fp32: the reference's mab; 16-bit handles: packed slices (see sparta_vbs_create)
    int kp16 = 0;                            // 16-bit handles: k depth of a step (32 or 64)
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
    // stream plan (w % 32 == 0): see vbs_spmm_f32_stream_kernel
    StepRec* d_steps[2] = {nullptr, nullptr};      // per tile type: [0] <= 32 rows, [1] 33..64 rows
    std::vector<StepRec> h_steps[2];               // host copies (padded), source of the gathered-B variants
    StepRec* d_steps_g[2] = {nullptr, nullptr};    // step lists for sparta_vbs_spmm_gathered with shard_rows == g_shard_rows
    int64_t g_shard_rows = 0;
    int32_t* d_wrange[2] = {nullptr, nullptr};
    FixRec* d_fix = nullptr;
    std::vector<std::pair<int64_t, int64_t>> zero_ranges;   // long runs of rows without blocks (local C rows), see vbs_zero_rows_kernel
    int32_t* d_fix_slots = nullptr;
    int64_t n_steps[2] = {0, 0};
    int32_t n_workers = 0, n_fix = 0, n_split = 0, n_slots = 0;
    int32_t max_tile_slots = 0;            // most partial images of one split tile
    int32_t* d_big_fix = nullptr;          // fix records with more than 2 * kFixGroup images (group stage)
    int32_t n_big_fix = 0;
    void* d_ws = nullptr;
    size_t d_ws_bytes = 0;
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
    void* d_sp_part = nullptr;             // partial rows of the segments
    size_t d_sp_part_bytes = 0;
    void* d_Brm = nullptr;                 // row-major copy of a column-major / gathered B
    size_t d_Brm_bytes = 0;
    void* d_spC = nullptr;                 // row-major results awaiting the scatter into a column-major C
    size_t d_spC_bytes = 0;
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

// fp32 -> fp16 / bf16 bits, round to nearest even (what the device conversion kernel does too)
inline uint16_t to_h16(float v, bool bf16) {
    if (bf16) {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    const _Float16 h = (_Float16)v;
    uint16_t o;
    std::memcpy(&o, &h, 2);
    return o;
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
    if (v->d_fix) (void)hipFree(v->d_fix);
    if (v->d_fix_slots) (void)hipFree(v->d_fix_slots);
    if (v->d_big_fix) (void)hipFree(v->d_big_fix);
    if (v->d_ws) (void)hipFree(v->d_ws);
    if (v->d_btail) (void)hipFree(v->d_btail);
    if (v->d_tune) (void)hipFree(v->d_tune);
    if (v->d_B16) (void)hipFree(v->d_B16);
    if (v->d_clk) (void)hipFree(v->d_clk);
    if (v->tev0) (void)hipEventDestroy(v->tev0);
    if (v->tev1) (void)hipEventDestroy(v->tev1);
    if (v->d_sp_rowptr) (void)hipFree(v->d_sp_rowptr);
    if (v->d_sp_col) (void)hipFree(v->d_sp_col);
    if (v->d_sp_val) (void)hipFree(v->d_sp_val);
    if (v->d_sp_crow) (void)hipFree(v->d_sp_crow);
    if (v->d_sp_list) (void)hipFree(v->d_sp_list);
    if (v->d_sp_segs) (void)hipFree(v->d_sp_segs);
    if (v->d_sp_long) (void)hipFree(v->d_sp_long);
    if (v->d_sp_part) (void)hipFree(v->d_sp_part);
    if (v->d_Brm) (void)hipFree(v->d_Brm);
    if (v->d_spC) (void)hipFree(v->d_spC);
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

namespace {

// ---- host side of the stream kernels: the plan ---------------------------------------------------------------------------
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
    std::vector<uint16_t> a16;                // 16-bit handles: A re-laid-out as dense row-major TM x kp slices, one per step
    int n_workers = 0, n_split = 0;
    int plan_aligned[2] = {0, 0};
    int64_t kp = SK_KP;
};

constexpr int64_t kZeroRangeRows = 2048;    // block-rows without blocks at least this tall are zero-filled by vbs_zero_rows_kernel

int build_stream_plans(const StreamPlanIn& in, StreamPlanHost& P) {
    using sparta::fail;
    const int64_t cols = in.cols, w = in.w, br0 = in.br0, br1 = in.br1, jab_lo = in.jab_lo, mab_lo = in.mab_lo;
    const int64_t* row_part = in.row_part; const int64_t* nzcount = in.nzcount; const int64_t* jab = in.jab; const float* mab = in.mab;
    const int32_t dtype = in.dtype, device = in.device;
    const bool h16 = dtype != SPARTA_F32;
    std::vector<StepRec>(&steps)[2] = P.steps;
    std::vector<int32_t>(&wrange)[2] = P.wrange;
    std::vector<FixRec>& fix = P.fix;
    std::vector<int32_t>& fix_slots = P.fix_slots;
    std::vector<uint16_t>& a16 = P.a16;
    int& n_workers = P.n_workers; int& n_split = P.n_split;
    int(&plan_aligned)[2] = P.plan_aligned;
    // ---- stream plans (persistent kernels): flatten tiles into 32-deep steps, cut into equal-cost worker ranges ----
    // One plan per tile TYPE: ty = 1 tiles of 33..64 rows (two 32-row MFMA tiles per wave and step), ty = 0 tiles of
    // <= 32 rows (one).  Each type runs in its own launch of a kernel instantiated for that type only.  A single kernel
    // that picks the variant per step looks equivalent but compiles badly: at every join of the two variants the register
    // allocator reconciles the in-flight A/B registers and the accumulators with v_mov behind s_waitcnt vmcnt(0) / the
    // MFMA drain, which collapses the 3-step prefetch (measured: 72 non-MFMA VALU per step, 69 % of the matrix peak).
    // k depth of a step: 32 for fp32; the 16-bit kernels take 64 when the block width allows (their steps are short: fewer, fatter)
    const int64_t kp = !h16 ? SK_KP : (w % 64 == 0 ? 64 : 32);
    P.kp = kp;
    if (w % SK_KP == 0) {
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        int per_cu = 2;
        if (const char* e = std::getenv("SPARTA_WORKERS_PER_CU")) per_cu = std::max(1, std::min(3, atoi(e)));
        n_workers = ((cus * per_cu + 7) / 8) * 8;
        // modelled cost of a step and of a tile's epilogue, per type
        int c2 = 20, c1 = 13, ct = 6;
        if (h16) { c2 = 12; c1 = 10; }                   // load-bound steps: cost ~ bytes moved, (64 + 128) vs (32 + 128) rows and columns
        if (const char* e = std::getenv("SPARTA_COST_MODEL")) sscanf(e, "%d,%d,%d", &c2, &c1, &ct);
        int64_t split_penalty = 120;                     // cost units (~0.11 us each) the fix-up launch adds to a split plan
        if (const char* e = std::getenv("SPARTA_SPLIT_PENALTY")) split_penalty = atoll(e);
        bool interleave = false;                         // SPARTA_STREAM_INTERLEAVE=1: deal whole tiles round-robin inside an XCD (measured: +-2 %, L2 locality is not the limit)
        if (const char* e = std::getenv("SPARTA_STREAM_INTERLEAVE")) interleave = atoi(e) != 0;
        double slot_bias = 0.0;                          // SPARTA_SLOT_BIAS: extra share of the workgroup dispatched first onto a CU
        if (const char* e = std::getenv("SPARTA_SLOT_BIAS")) slot_bias = std::max(-0.9, std::min(0.9, atof(e)));
        int align_mode = -1;                             // SPARTA_STREAM_ALIGN=0 always split, 1 never split, unset: cheaper one
        if (const char* e = std::getenv("SPARTA_STREAM_ALIGN")) align_mode = atoi(e) ? 1 : 0;
        if ((int64_t)cols > INT32_MAX)
            return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: matrix too large for 32-bit step indexing");
        for (int ty = 0; ty < 2; ty++) {
            std::vector<StepRec>& st = steps[ty];
            struct TileSpan { int64_t first, last; int32_t c_row, mt; };     // step range of a tile
            std::vector<TileSpan> spans;
            std::vector<int64_t> cum;                                        // cumulative cost BEFORE step s
            int64_t total_cost = 0;
            {
                int64_t jo2 = 0, mo2 = 0;
                const int64_t row0 = row_part[br0];
                for (int64_t ib = br0; ib < br1; ib++) {
                    const int64_t h = row_part[ib + 1] - row_part[ib];
                    const int64_t nb = nzcount[ib];
                    const bool skipped = in.skip && in.skip[ib - br0];
#ifdef SPARTA_TIMELINE                              // developer build only (make timeline): a probe makes the products WRONG on purpose
                    const int dbg_probe = [] { const char* e = std::getenv("SPARTA_DBG_PROBE"); return e ? atoi(e) : 0; }();
#else
                    constexpr int dbg_probe = 0;
#endif
                    const bool zero_range = nb == 0 && !skipped && h >= kZeroRangeRows;     // one streamed fill instead of h / 64 fix-up tiles
                    if (zero_range && ty == 0) P.zero_ranges.emplace_back(row_part[ib] - row0, h);
                    for (int64_t r0 = 0; r0 < h && !skipped && !zero_range; r0 += SK_TM) {
                        const int32_t mt = (int32_t)std::min<int64_t>(SK_TM, h - r0);
                        if ((mt > 32 ? 1 : 0) != ty) continue;
                        const int32_t c_row = (int32_t)(row_part[ib] - row0 + r0);
                        if (nb == 0) {                                      // nothing to multiply: the fix-up kernel writes the zeros
                            fix.push_back(FixRec{c_row, mt, 0, 0});
                            continue;
                        }
                        TileSpan sp{(int64_t)st.size(), 0, c_row, mt};
                        for (int64_t b = 0; b < nb; b++) {
                            const int64_t jb = jab[jab_lo + jo2 + b];
                            for (int64_t ks = 0; ks < w; ks += kp) {
                                StepRec r;
                                r.a_off = mo2 + r0 + (b * w + ks) * h;
                                if (h16) {                                  // pack this step's slice: [row][k], rows past the tile zero
                                    const int64_t tms = ty ? 64 : 32;
                                    r.a_off = (int64_t)a16.size();
                                    a16.resize(a16.size() + (size_t)(tms * kp), 0);
                                    uint16_t* dst = a16.data() + r.a_off;
                                    const float* blk = mab + mab_lo + mo2 + b * h * w;       // column-major h x w block
                                    for (int64_t rr = 0; rr < mt; rr++)
                                        for (int64_t kk = 0; kk < kp; kk++)
                                            dst[rr * kp + kk] = to_h16(blk[(ks + kk) * h + r0 + rr], dtype == SPARTA_BF16);
                                }
                                r.b_row = (int32_t)(jb * w + ks);
                                r.h = (int32_t)h;
                                r.c_row = c_row;
                                r.mt_flags = mt;
                                if ((jb + 1) * w > cols) { r.mt_flags |= STEP_TAIL; r.b_row = (int32_t)ks; }   // read from the zero-padded B_tail
                                r.slot = -1;
                                r.pad = 0;
                                if (dbg_probe) {                            // developer probe (timing only, results are wrong): which stream bounds a step
                                    if (dbg_probe & 1) r.b_row = (int32_t)ks;                                   // every panel of B is the same (cache-hot) one
                                    if ((dbg_probe & 2) && h16) r.a_off = (b * w + ks) / kp * (ty ? 64 : 32) * kp;   // every tile reads the first slices of A
                                    if ((dbg_probe & 2) && !h16) r.a_off = r0 + (b * w + ks) * h;
                                    if (dbg_probe & 4) r.c_row = 0;                                             // every tile writes the first rows of C
                                }
                                cum.push_back(total_cost);
                                total_cost += ty ? c2 : c1;
                                st.push_back(r);
                            }
                        }
                        total_cost += ct;
                        sp.last = (int64_t)st.size() - 1;
                        st[(size_t)sp.first].mt_flags |= STEP_FIRST;
                        st[(size_t)sp.last].mt_flags |= STEP_LAST;
                        spans.push_back(sp);
                    }
                    jo2 += nb;
                    mo2 += nb * h * w;
                }
            }
            if ((int64_t)st.size() > INT32_MAX - 64)
                return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: matrix too large for 32-bit step indexing");
            const int64_t S = (int64_t)st.size();
            if (S == 0) continue;
            cum.push_back(total_cost);
            // boundaries: worker (x, j) = the j-th of the P/8 sub-ranges of XCD x's eighth; workgroup id = x + 8 j
            const int per_x = n_workers / 8;
            std::vector<int64_t> bnd((size_t)n_workers + 1, S);
            bnd[0] = 0;
            // Share of a worker.  The two (three) workgroups that share a CU do not progress at the same rate: the SIMD arbiter
            // serves the OLDER wave first, so the workgroup dispatched first (j < #CU per XCD) runs ahead -- measured on equal
            // ranges: the older one finished 512 steps in ~780 us, the younger one in 1030 us, the last 250 us alone on the CU at
            // single-occupancy speed.  Give the older one `slot_bias` more work and the younger one as much less.
            std::vector<double> wpos((size_t)n_workers, 1.0), wcum((size_t)n_workers + 1, 0.0);
            {
                const int cus_x = std::max(1, per_x / per_cu);
                for (int pos = 0; pos < n_workers; pos++) {
                    const int slot = std::min(per_cu - 1, (pos % per_x) / cus_x);
                    wpos[(size_t)pos] = per_cu == 1 ? 1.0 : 1.0 + slot_bias * (1.0 - 2.0 * slot / (double)(per_cu - 1));
                }
                for (int pos = 0; pos < n_workers; pos++) wcum[(size_t)pos + 1] = wcum[(size_t)pos] + wpos[(size_t)pos];
            }
            for (int k = 1; k < n_workers; k++) {
                const int64_t target = (int64_t)((double)total_cost * wcum[(size_t)k] / wcum[(size_t)n_workers]);
                int64_t pos = std::lower_bound(cum.begin(), cum.end(), target) - cum.begin();
                bnd[(size_t)k] = std::min<int64_t>(std::max(pos, bnd[(size_t)k - 1]), S);
            }
            // Alternative: ranges that end on tile boundaries (no split tile, no fix-up launch).  Splitting balances to one
            // step but pays the fix-up (a second launch that re-reads the partial images: ~10 us measured, `split_penalty`
            // cost units); whole tiles cost at most one tile of imbalance.  Many short tiles -> aligned; few long -> split.
            {
                const int64_t step_cost = ty ? c2 : c1;
                auto tile_cost = [&](size_t t) { return (spans[t].last - spans[t].first + 1) * step_cost + ct; };
                int64_t lo = 0, hi = total_cost * 2;
                for (size_t t = 0; t < spans.size(); t++) lo = std::max(lo, tile_cost(t));
                auto cap = [&](int64_t L, int64_t bin) { return bin < n_workers ? (int64_t)((double)L * wpos[(size_t)bin]) : L; };
                auto bins_needed = [&](int64_t L) {                 // bin b holds at most L x (its worker's share)
                    int64_t bins = 1, cur = 0;
                    for (size_t t = 0; t < spans.size(); t++) {
                        const int64_t c = tile_cost(t);
                        if (cur > 0 && cur + c > cap(L, bins - 1)) { bins++; cur = 0; }
                        cur += c;
                    }
                    return bins;
                };
                while (lo < hi) {                                   // smallest makespan L that fits n_workers contiguous bins
                    const int64_t mid = lo + (hi - lo) / 2;
                    if (bins_needed(mid) <= n_workers) hi = mid; else lo = mid + 1;
                }
                const int64_t split_makespan = (total_cost + n_workers - 1) / n_workers + split_penalty;
                const bool aligned = align_mode == 1 || (align_mode < 0 && lo <= split_makespan);
                if (aligned) {
                    std::fill(bnd.begin(), bnd.end(), S);
                    bnd[0] = 0;
                    int64_t bin = 0, cur = 0;
                    for (size_t t = 0; t < spans.size(); t++) {
                        const int64_t c = tile_cost(t);
                        if (cur > 0 && cur + c > cap(lo, bin) && bin + 1 < n_workers) { bin++; bnd[(size_t)bin] = spans[t].first; cur = 0; }
                        cur += c;
                    }
                    if (interleave) {
                        // Whole tiles can go to any worker.  Keep the 64 workers of an XCD close together in the matrix at every
                        // moment: the XCD takes a contiguous eighth of the tiles (by cost) and deals them, in matrix order, to
                        // its least-loaded worker (uniform tiles: worker j gets tiles j, j+64, ...).  The B rows the XCD
                        // touches at one time are then a narrow moving window that stays in its 4 MB L2, instead of 64
                        // windows spread over the whole eighth.  The step list is rebuilt in worker order.
                        std::vector<std::vector<size_t>> mine((size_t)n_workers);
                        size_t t = 0;
                        int64_t seen = 0;
                        for (int x = 0; x < 8; x++) {
                            const int64_t upto = total_cost * (x + 1) / 8;
                            std::vector<int64_t> load((size_t)per_x, 0);
                            while (t < spans.size() && (x == 7 || seen + tile_cost(t) / 2 <= upto)) {
                                size_t best = 0;
                                for (size_t j = 1; j < load.size(); j++) if (load[j] < load[best]) best = j;
                                mine[(size_t)x * per_x + best].push_back(t);
                                load[best] += tile_cost(t);
                                seen += tile_cost(t);
                                t++;
                            }
                        }
                        std::vector<StepRec> ns;
                        std::vector<TileSpan> nspans;
                        ns.reserve(st.size());
                        nspans.reserve(spans.size());
                        for (int pos = 0; pos < n_workers; pos++) {
                            bnd[(size_t)pos] = (int64_t)ns.size();
                            for (size_t tt : mine[(size_t)pos]) {
                                TileSpan sp = spans[tt];
                                const int64_t len = sp.last - sp.first + 1;
                                ns.insert(ns.end(), st.begin() + sp.first, st.begin() + sp.last + 1);
                                sp.first = (int64_t)ns.size() - len;
                                sp.last = (int64_t)ns.size() - 1;
                                nspans.push_back(sp);
                            }
                        }
                        bnd[(size_t)n_workers] = S;
                        st.swap(ns);
                        spans.swap(nspans);
                    }
                }
                plan_aligned[ty] = aligned ? 1 : 0;
            }
            wrange[ty].assign((size_t)n_workers * 2, 0);
            std::vector<int32_t> wid_of_pos((size_t)n_workers);
            for (int pos = 0; pos < n_workers; pos++) {
                const int x = pos / per_x, j = pos % per_x;
                const int wid = x + 8 * j;
                wid_of_pos[(size_t)pos] = wid;
                wrange[ty][(size_t)wid * 2] = (int32_t)bnd[(size_t)pos];
                wrange[ty][(size_t)wid * 2 + 1] = (int32_t)bnd[(size_t)pos + 1];
            }
            // segments: a tile cut by a boundary is split; every segment writes one workspace image
            // (slots are numbered densely over both types: both launches finish before the fix-up kernel reads them)
            size_t ti = 0;
            for (int pos = 0; pos < n_workers; pos++) {
                const int64_t s0 = bnd[(size_t)pos], s1 = bnd[(size_t)pos + 1];
                if (s0 >= s1) continue;
                while (ti < spans.size() && spans[ti].last < s0) ti++;
                for (size_t t = ti; t < spans.size() && spans[t].first < s1; t++) {
                    const int64_t a = std::max(spans[t].first, s0), b = std::min(spans[t].last, s1 - 1);
                    const bool whole = a == spans[t].first && b == spans[t].last;
                    st[(size_t)a].mt_flags |= STEP_FIRST;
                    st[(size_t)b].mt_flags |= STEP_LAST;
                    if (!whole) {
                        const int32_t slot = (int32_t)fix_slots.size();          // dense: one image per segment, both types
                        st[(size_t)b].mt_flags |= STEP_SPLIT;
                        st[(size_t)b].slot = slot;
                        if (a == spans[t].first) {                          // first segment of the tile opens its fix-up record
                            fix.push_back(FixRec{spans[t].c_row, spans[t].mt, (int32_t)fix_slots.size(), 0});
                            n_split++;
                        }
                        fix_slots.push_back(slot);
                        fix.back().n_slots++;
                    }
                }
            }
        }
    }

    return SPARTA_OK;
}

}  // namespace

extern "C" {

int sparta_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// `ext` (sparta_vbs_create_from_csr): the sparse-row part of the matrix decided and collected by the hybrid host builder --
// those block-rows have nzcount = 0 in the arrays given here and must get neither tiles nor zero-fill records.
static int create_core(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t w, const int64_t* row_part,
                       const int64_t* nzcount, const int64_t* jab, const float* mab, int64_t br0, int64_t br1, int32_t dtype,
                       int32_t device, const sparta::HybridSparse* ext) {
    using sparta::fail;
    if (!out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: out is NULL");
    *out = nullptr;
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
    if (ndev <= 0) return fail(SPARTA_ERR_NO_DEVICE, "sparta_vbs_create: no HIP device visible (this path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SPARTA_ERR_INVALID, "sparta_vbs_create: device index out of range");

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
        sparse_flag = ext->flag;
        bool any = false;
        for (uint8_t f : sparse_flag) any = any || f;
        if (!any) sparse_flag.clear();
        sp_rowptr.assign(1, 0);
        for (size_t t = 0; t < ext->crow.size(); t++) {
            for (int64_t k = ext->rowptr[t]; k < ext->rowptr[t + 1]; k++) {
                const float a = stored(ext->val[(size_t)k]);
                if (a != 0.0f && ext->col[(size_t)k] < cols) { sp_col.push_back(ext->col[(size_t)k]); sp_val.push_back(a); }
            }
            sp_crow.push_back(ext->crow[t]);
            sp_rowptr.push_back((int64_t)sp_col.size());
        }
    } else {
        double K = 24.0;
        if (const char* e = std::getenv("SPARTA_SPARSE_K")) K = atof(e);
        if (K > 0.0) {
            sparse_flag.assign((size_t)(br1 - br0), 0);
            int64_t jo2 = 0, mo2 = 0, n_flagged = 0;
            const int64_t row0 = row_part[br0];
            std::vector<int64_t> cnt;
            sp_rowptr.push_back(0);
            for (int64_t ib = br0; ib < br1; ib++) {
                const int64_t h = row_part[ib + 1] - row_part[ib], nb = nzcount[ib];
                const float* blk = mab + mab_lo + mo2;
                const int64_t n_el = nb * h * w;
                int64_t nnz = 0;
                for (int64_t q = 0; q < n_el; q++) nnz += stored(blk[q]) != 0.0f;
                const int64_t kdep = h16 && w % 64 == 0 ? 64 : 32;                     // k depth of a step of the kernels this handle would use
                const double steps_br = (double)nb * (double)((w + kdep - 1) / kdep) * (double)((h + 31) / 32);
                if (h > 0 && nb > 0 && (double)nnz < K * steps_br && (int64_t)sp_col.size() + nnz < ((int64_t)1 << 40)) {
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
        kSpSeg = total < ((int64_t)4 << 20) ? 128 : (total < ((int64_t)16 << 20) ? 256 : 512);
    }
    const int64_t kSpLong = 2 * kSpSeg;
    for (size_t t = 0; t < sp_crow.size(); t++) {
        const int64_t p0 = sp_rowptr[t], n = sp_rowptr[t + 1] - p0;
        if (n <= kSpLong) { sp_list.push_back((int32_t)t); continue; }
        SpLongRec lr{(int32_t)t, (int32_t)sp_segs.size(), 0, 0};
        for (int64_t o = 0; o < n; o += kSpSeg) { sp_segs.push_back(SpSegRec{p0 + o, (int32_t)std::min<int64_t>(kSpSeg, n - o), 0}); lr.n_seg++; }
        sp_long.push_back(lr);
    }
    n_sp_short = (int64_t)sp_list.size(); n_sp_long = (int64_t)sp_long.size();
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
        if (int rc = build_stream_plans(pin, plan)) return rc;
    }
    std::vector<StepRec>(&steps)[2] = plan.steps;
    std::vector<int32_t>(&wrange)[2] = plan.wrange;
    std::vector<FixRec>& fix = plan.fix;
    std::vector<int32_t>& fix_slots = plan.fix_slots;
    std::vector<uint16_t>& a16 = plan.a16;
    const int n_workers = plan.n_workers, n_split = plan.n_split;
    const int64_t kp = plan.kp;

    sparta_vbs* v = new (std::nothrow) sparta_vbs;
    if (!v) return fail(SPARTA_ERR_ALLOC, "sparta_vbs_create: out of host memory");
    v->device = device; v->dtype = dtype;
    v->zero_ranges = plan.zero_ranges;
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
    if (!h16) {
        v->a_bytes = (nztot + 128) * (int64_t)sizeof(float);
        CREATE_TRY(hipMalloc((void**)&v->d_A, (size_t)v->a_bytes));
        CREATE_TRY(hipMemset(v->d_A, 0, (size_t)v->a_bytes));
        if (nztot > 0) CREATE_TRY(hipMemcpy(v->d_A, mab + mab_lo, (size_t)nztot * sizeof(float), hipMemcpyHostToDevice));
    } else {
        // the look-ahead of the pipeline reads up to 5 slices past the last one (never multiplied): pad
        v->a_bytes = ((int64_t)a16.size() + 8 * 64 * 64) * (int64_t)sizeof(uint16_t);
        CREATE_TRY(hipMalloc((void**)&v->d_A, (size_t)v->a_bytes));
        CREATE_TRY(hipMemset(v->d_A, 0, (size_t)v->a_bytes));
        if (!a16.empty()) CREATE_TRY(hipMemcpy(v->d_A, a16.data(), a16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
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
    if (!steps[0].empty() || !steps[1].empty() || !fix.empty() || !plan.zero_ranges.empty()) {
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
            CREATE_TRY(hipMalloc((void**)&v->d_wrange[ty], wrange[ty].size() * sizeof(int32_t)));
            CREATE_TRY(hipMemcpy(v->d_wrange[ty], wrange[ty].data(), wrange[ty].size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        if (!fix.empty()) {
            CREATE_TRY(hipMalloc((void**)&v->d_fix, fix.size() * sizeof(FixRec)));
            CREATE_TRY(hipMemcpy(v->d_fix, fix.data(), fix.size() * sizeof(FixRec), hipMemcpyHostToDevice));
        }
        CREATE_TRY(hipMalloc((void**)&v->d_fix_slots, std::max<size_t>(fix_slots.size(), 1) * sizeof(int32_t)));
        if (!fix_slots.empty()) CREATE_TRY(hipMemcpy(v->d_fix_slots, fix_slots.data(), fix_slots.size() * sizeof(int32_t), hipMemcpyHostToDevice));
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
            CREATE_TRY(hipMalloc(&v->d_sp_long, sp_long.size() * sizeof(SpLongRec)));
            CREATE_TRY(hipMemcpy(v->d_sp_long, sp_long.data(), sp_long.size() * sizeof(SpLongRec), hipMemcpyHostToDevice));
        }
    }
    CREATE_TRY(hipEventCreate(&v->ev0));
    CREATE_TRY(hipEventCreate(&v->ev1));
    CREATE_TRY(hipEventCreate(&v->tev0));
    CREATE_TRY(hipEventCreate(&v->tev1));
#undef CREATE_TRY
    *out = v;
    return SPARTA_OK;
}

int sparta_vbs_create_range(sparta_vbs_t** out, int64_t rows, int64_t cols, int64_t block_rows, int64_t w, const int64_t* row_part,
                            const int64_t* nzcount, const int64_t* jab, const float* mab, int64_t br0, int64_t br1, int32_t dtype,
                            int32_t device) {
    return create_core(out, rows, cols, block_rows, w, row_part, nzcount, jab, mab, br0, br1, dtype, device, nullptr);
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

int sparta_vbs_sparse_info(const sparta_vbs_t* A, int64_t* info) {
    if (!A || !info) return sparta::fail(SPARTA_ERR_INVALID, "sparta_vbs_sparse_info: NULL argument");
    info[0] = A->n_sp_rows; info[1] = A->sp_nnz; info[2] = A->n_sp_short; info[3] = A->n_sp_long;
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

// shared implementation of sparta_vbs_spmm / sparta_vbs_spmm_gathered
template <int KP, bool MI2, bool DEEP>
void launch_h16_d(bool bf16, bool gathered, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (gathered) {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_stream_kernel<KP, MI2, true, true, DEEP>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_stream_kernel<KP, MI2, false, true, DEEP>), grid, dim3(kThreads), 0, st, sp);
    } else {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_stream_kernel<KP, MI2, true, false, DEEP>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_stream_kernel<KP, MI2, false, false, DEEP>), grid, dim3(kThreads), 0, st, sp);
    }
}
// SPARTA_H16_DEPTH=4 selects the four-register-set pipeline (loads 5 steps ahead); default: two sets, 3 steps ahead -- measured equal
// within 1 % on every shape (scripts/h16_depth_ab.py): the 16-bit steps are not bound by the bytes in flight
bool h16_deep() { const char* e = std::getenv("SPARTA_H16_DEPTH"); return e && atoi(e) == 4; }   // read per launch: scripts flip it between timings
// Which 16-bit kernel: SPARTA_H16_PATH=lds | direct forces one; default (auto): operands straight into registers
// (vbs_spmm_h16_direct_kernel) for 32-deep steps of one 32-row MFMA tile -- 36.4 vs 38.0 us on the flagship, bit-identical --
// and the LDS-staged kernel for everything else (64-row tiles / 64-deep steps: the four waves would each fetch the whole A
// slice and a wave-load would touch 64 cache lines: 57 vs 29 us).  Read per launch: scripts flip it between timings.
bool h16_direct(int kp, bool mi2) {
    const char* e = std::getenv("SPARTA_H16_PATH");
    if (e && e[0] == 'l') return false;
    if (e && e[0] == 'd') return true;
    return kp == 32 && !mi2;
}
template <int KP, bool MI2>
void launch_h16_direct(bool bf16, bool gathered, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (gathered) {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, MI2, true, true>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, MI2, false, true>), grid, dim3(kThreads), 0, st, sp);
    } else {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, MI2, true, false>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, MI2, false, false>), grid, dim3(kThreads), 0, st, sp);
    }
}
template <int KP, bool MI2>
void launch_h16(bool bf16, bool gathered, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (h16_direct(KP, MI2)) { launch_h16_direct<KP, MI2>(bf16, gathered, grid, st, sp); return; }
    if (h16_deep()) launch_h16_d<KP, MI2, true>(bf16, gathered, grid, st, sp);
    else launch_h16_d<KP, MI2, false>(bf16, gathered, grid, st, sp);
}

// long runs of rows without blocks, accumulate = 0: streamed zero fill (vbs_zero_rows_kernel), one launch per run
void launch_zero_ranges(sparta_vbs_t* A, float* C, int64_t ldc, bool c_row_major, int n_cols, hipStream_t st) {
    for (const auto& zr : A->zero_ranges) {
        const int64_t n_lines = c_row_major ? zr.second : (int64_t)n_cols, line_len = c_row_major ? (int64_t)n_cols : zr.second;
        const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (line_len / 4 + kThreads - 1) / kThreads));
        const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_lines, 16384));
        hipLaunchKernelGGL(vbs_zero_rows_kernel, dim3(gx, gy), dim3(kThreads), 0, st, C, ldc, (int)c_row_major, zr.first, zr.second, n_cols);
    }
}

// step lists for a gathered B (slab index + row inside the slab), rebuilt when the slab height changes
int ensure_gathered_steps(sparta_vbs_t* A, int64_t shard_rows, hipStream_t st) {
    if (A->g_shard_rows == shard_rows) return SPARTA_OK;
    for (int ty = 0; ty < 2; ty++) {
        if (A->h_steps[ty].empty()) continue;
        std::vector<StepRec> g = A->h_steps[ty];
        for (StepRec& r : g) { r.pad = (int32_t)(r.b_row / shard_rows); r.b_row = (int32_t)(r.b_row % shard_rows); }
        if (!A->d_steps_g[ty]) HIP_TRY(hipMalloc((void**)&A->d_steps_g[ty], g.size() * sizeof(StepRec)));
        HIP_TRY(hipMemcpyAsync(A->d_steps_g[ty], g.data(), g.size() * sizeof(StepRec), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));                       // g goes out of scope
    }
    A->g_shard_rows = shard_rows;
    return SPARTA_OK;
}

// sparse-row path, shared by the fp32 and the 16-bit handles.  bk: element type of B (0 fp32, 1 fp16, 2 bf16); C is fp32.
template <int VEC, int BK>
void launch_sparse_kernels(sparta_vbs_t* A, SparseParams q, unsigned gy, hipStream_t st) {
    if (A->n_sp_short > 0) {
        q.list = A->d_sp_list; q.n_list = (int32_t)A->n_sp_short;
        hipLaunchKernelGGL((sparse_rows_kernel<VEC, BK>), dim3((unsigned)((A->n_sp_short + 3) / 4), gy), dim3(kThreads), 0, st, q);
    }
    if (A->n_sp_long > 0) {
        float* part = (float*)A->d_sp_part;
        hipLaunchKernelGGL((sparse_segments_kernel<VEC, BK>), dim3((unsigned)((A->n_sp_segs + 3) / 4), gy), dim3(kThreads), 0, st, q,
                           (const SpSegRec*)A->d_sp_segs, (int32_t)A->n_sp_segs, part);
        hipLaunchKernelGGL(sparse_reduce_kernel<VEC>, dim3((unsigned)((A->n_sp_long + 3) / 4), gy), dim3(kThreads), 0, st, q,
                           (const SpLongRec*)A->d_sp_long, (int32_t)A->n_sp_long, (const float*)part);
    }
}

int launch_sparse_rows(sparta_vbs_t* A, const void* dB, int64_t ldb, bool b_row_major, int64_t shard_rows, int64_t shard_stride, int bk,
                       int32_t n_cols, float* dC, int64_t ldc, bool c_row_major, bool accumulate, hipStream_t st) {
    const size_t esz = bk == 0 ? 4 : 2;
    SparseParams q;
    q.rowptr = A->d_sp_rowptr; q.col = A->d_sp_col; q.val = A->d_sp_val; q.crow = A->d_sp_crow;
    q.list = nullptr; q.n_list = 0;
    q.N = n_cols; q.accumulate = accumulate;
    q.b_col_stride = 0; q.shard_rows = 0; q.shard_stride = 0;
    // A column-major B read in place costs one 64-byte line per ELEMENT (16 x the bytes of a row-major row); transposing costs
    // 2 x |B| once.  In place wins while  nnz * 16 < 2 * cols.
    const bool in_place = !(b_row_major && shard_rows == 0) && A->sp_nnz * 8 < A->cols;
    if (b_row_major && shard_rows == 0) { q.B = dB; q.ldb = ldb; }
    else if (in_place) { q.B = dB; q.ldb = 0; q.b_col_stride = ldb; q.shard_rows = shard_rows; q.shard_stride = shard_stride; }
    else {
        if (int rc = ensure_scratch(&A->d_Brm, &A->d_Brm_bytes, (size_t)A->cols * (size_t)n_cols * esz)) return rc;
        const int64_t n_wg = ((A->cols + 63) / 64) * (int64_t)((n_cols + 63) / 64);
        if (n_wg > INT32_MAX) return sparta::fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: B too large for the transpose grid");
        const dim3 grid((unsigned)n_wg);
        if (bk == 0) hipLaunchKernelGGL(b_to_row_major_kernel<float>, grid, dim3(kThreads), 0, st, (const float*)dB, ldb, shard_rows, shard_stride, A->cols, (int)n_cols, (float*)A->d_Brm);
        else hipLaunchKernelGGL(b_to_row_major_kernel<unsigned short>, grid, dim3(kThreads), 0, st, (const unsigned short*)dB, ldb, shard_rows, shard_stride, A->cols, (int)n_cols, (unsigned short*)A->d_Brm);
        q.B = A->d_Brm; q.ldb = n_cols;
    }
    if (c_row_major) { q.out = dC; q.ldo = ldc; q.out_is_c = 1; }
    else {
        if (int rc = ensure_scratch(&A->d_spC, &A->d_spC_bytes, (size_t)A->n_sp_rows * (size_t)n_cols * sizeof(float))) return rc;
        q.out = (float*)A->d_spC; q.ldo = n_cols; q.out_is_c = 0;
    }
    if (A->n_sp_long > 0)
        if (int rc = ensure_scratch(&A->d_sp_part, &A->d_sp_part_bytes, (size_t)A->n_sp_segs * (size_t)n_cols * sizeof(float))) return rc;
    // widest vector the shapes allow: every row start VEC-element aligned, N a multiple of 64 * VEC (no ragged chunk)
    auto aligned = [&](int v) {
        return n_cols % (64 * v) == 0 && q.ldb % v == 0 && q.ldo % v == 0 && ((uintptr_t)q.B % (esz * v)) == 0 && ((uintptr_t)q.out % (4 * v)) == 0;
    };
    const int vec = in_place ? 1 : (aligned(4) ? 4 : (aligned(2) ? 2 : 1));
    const unsigned gy = (unsigned)((n_cols + 64 * vec - 1) / (64 * vec));
#define SPARTA_SP_DISPATCH(V_)                                                             \
    do {                                                                                   \
        if (bk == 0) launch_sparse_kernels<V_, 0>(A, q, gy, st);                   \
        else if (bk == 1) launch_sparse_kernels<V_, 1>(A, q, gy, st);              \
        else launch_sparse_kernels<V_, 2>(A, q, gy, st);                           \
    } while (0)
    if (vec == 4) SPARTA_SP_DISPATCH(4); else if (vec == 2) SPARTA_SP_DISPATCH(2); else SPARTA_SP_DISPATCH(1);
#undef SPARTA_SP_DISPATCH
    if (!q.out_is_c)
        hipLaunchKernelGGL(sparse_c_scatter_kernel, dim3((unsigned)(((A->n_sp_rows + 63) / 64) * (int64_t)((n_cols + 63) / 64))), dim3(kThreads), 0, st,
                           (const float*)A->d_spC, A->d_sp_crow, A->n_sp_rows, (int)n_cols, dC, ldc, (int)accumulate);
    HIP_TRY(hipGetLastError());
    return SPARTA_OK;
}

// 16-bit handles (SPARTA_F16 / SPARTA_BF16): A and B in the 16-bit type, fp32 accumulation, fp32 C.  Device pointers: B is a
// 16-bit column-major matrix (ldb in elements, even).  Host pointers keep the reference's contract (fp32 B in, fp32 C out):
// B is converted on the device (round to nearest even).
int spmm16_impl(sparta_vbs_t* A, const void* B, int64_t ldb, int32_t b_layout, int64_t shard_rows, int64_t shard_stride, int32_t n_cols, void* C,
                int64_t ldc, int32_t c_layout, int32_t accumulate, int32_t ptr_space, hipStream_t st, int32_t algo, float* dt_ms) {
    using sparta::fail;
    if (algo != SPARTA_SPMM_MFMA) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: SPARTA_SPMM_EXACT needs an fp32 handle");
    if (shard_rows != 0 && ptr_space != SPARTA_PTR_DEVICE) return fail(SPARTA_ERR_INVALID, "sparta_vbs_spmm_gathered: device pointers only");
    if (b_layout != SPARTA_COL_MAJOR) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: 16-bit handles need a column-major B (k contiguous)");
    if (n_cols % kTN != 0) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: 16-bit handles need n_cols % 128 == 0");
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
        if (bf16) hipLaunchKernelGGL((vbs_convert_h16_kernel<true>), dim3(1024), dim3(kThreads), 0, st, (const float*)A->d_B, ldb, A->cols, (int64_t)n_cols, (uint16_t*)A->d_B16, ldb16);
        else hipLaunchKernelGGL((vbs_convert_h16_kernel<false>), dim3(1024), dim3(kThreads), 0, st, (const float*)A->d_B, ldb, A->cols, (int64_t)n_cols, (uint16_t*)A->d_B16, ldb16);
        dB = (const uint16_t*)A->d_B16;
        dC = (float*)A->d_C;
    } else if (ldb % 2 != 0) {
        return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: 16-bit B needs an even leading dimension (16-byte loads start on 4-byte boundaries)");
    }
    // 32-bit byte offsets inside a 128-column slab (see the fp32 path): 127 x ld x element size < 2^31
    if (ldb16 * 128 * 2 >= ((int64_t)1 << 31) - 65536 || (c_layout == SPARTA_ROW_MAJOR ? ldc * 64 : ldc * 128) * 4 >= ((int64_t)1 << 31) - 65536)
        return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: leading dimension too large for the 16-bit stream kernels (ldb < 8.3 M, ldc < 4.1 M elements)");
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
    sp.N = n_cols; sp.w = (int32_t)A->w; sp.B_tail = nullptr; sp.clk = nullptr;
    if (A->n_steps[0] + A->n_steps[1] > 0) {
        if (prof) HIP_TRY(hipEventRecord(A->cev[0][0], st));
        if (A->has_tail && shard_rows == 0) {
            if (int rc = ensure_scratch(&A->d_btail, &A->d_btail_bytes, (size_t)A->w * n_cols * sizeof(uint16_t))) return rc;
            const int64_t row0 = ((A->cols - 1) / A->w) * A->w;
            hipLaunchKernelGGL(vbs_tail_copy_h16_kernel, dim3(32), dim3(kThreads), 0, st, dB, ldb16, row0, A->cols, (int)A->w, (int)n_cols, (uint16_t*)A->d_btail);
            sp.B_tail = (const float*)A->d_btail;
        }
        const dim3 grid((unsigned)A->n_workers, (unsigned)n_nt);
        const int probe_ty = A->n_steps[1] >= A->n_steps[0] ? 1 : 0;
        for (int ty = 1; ty >= 0; ty--) {
            if (A->n_steps[ty] == 0) continue;
            sp.steps = shard_rows > 0 ? A->d_steps_g[ty] : A->d_steps[ty]; sp.worker_range = A->d_wrange[ty];
            sp.clk = (prof && ty == probe_ty) ? A->d_clk : nullptr;
            const bool gth = shard_rows > 0;
            if (A->kp16 == 64) { if (ty) launch_h16<64, true>(bf16, gth, grid, st, sp); else launch_h16<64, false>(bf16, gth, grid, st, sp); }
            else { if (ty) launch_h16<32, true>(bf16, gth, grid, st, sp); else launch_h16<32, false>(bf16, gth, grid, st, sp); }
        }
        if (prof) { HIP_TRY(hipEventRecord(A->cev[0][1], st)); A->class_ran[0] = true; }
    }
    const bool fix_launch = A->n_fix > 0 && !(accumulate && A->n_split == 0);   // C += 0 for the block-rows without blocks: nothing to launch
    const bool zero_launch = !A->zero_ranges.empty() && !accumulate;
    if (fix_launch || zero_launch) {
        if (prof) HIP_TRY(hipEventRecord(A->cev[1][0], st));
        if (fix_launch && A->n_big_fix > 0)
            hipLaunchKernelGGL(vbs_spmm_f32_fixup_group_kernel, dim3((unsigned)A->n_big_fix, (unsigned)n_nt, (unsigned)((A->max_tile_slots + kFixGroup - 1) / kFixGroup)),
                               dim3(kThreads), 0, st, A->d_fix, A->d_big_fix, A->d_fix_slots, (float*)A->d_ws, (int64_t)slab);
        if (fix_launch)
            hipLaunchKernelGGL(vbs_spmm_f32_fixup_kernel, dim3((unsigned)A->n_fix, (unsigned)n_nt), dim3(kThreads), 0, st, A->d_fix, A->d_fix_slots,
                               (const float*)A->d_ws, (int64_t)slab, dC, ldc, (int)(c_layout == SPARTA_ROW_MAJOR), (int)(accumulate != 0));
        if (zero_launch) launch_zero_ranges(A, dC, ldc, c_layout == SPARTA_ROW_MAJOR, n_cols, st);
        if (prof) { HIP_TRY(hipEventRecord(A->cev[1][1], st)); A->class_ran[1] = true; }
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

    if (algo == SPARTA_SPMM_EXACT && A->ext_sparse && A->n_sp_rows > 0)
        return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_spmm: SPARTA_SPMM_EXACT needs the dense image of every block-row (handle made by sparta_vbs_create_from_csr)");
    if (dt_ms) HIP_TRY(hipEventRecord(A->ev0, st));
    if (algo == SPARTA_SPMM_EXACT) {
        if (A->n_brows > 0) {
            hipLaunchKernelGGL(vbs_spmm_f32_exact_kernel, dim3((unsigned)A->n_brows), dim3(kThreads), 0, st, A->d_brows, A->d_jab,
                               A->d_A, dB, dC, ldb, ldc, A->cols, (int)n_cols, (int)A->w, (int)(b_layout == SPARTA_ROW_MAJOR),
                               (int)(c_layout == SPARTA_ROW_MAJOR), (int)accumulate, shard_rows, shard_stride);
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
            StreamParams sp;
            sp.A = A->d_A; sp.B = dB; sp.C = Cout; sp.ws = (float*)A->d_ws;
            sp.ldb = ldb; sp.ldc = ldc; sp.cols = A->cols; sp.shard_rows = shard_rows; sp.shard_stride = shard_stride;
            sp.ws_slab_stride = (int64_t)slab; sp.accumulate = accumulate != 0; sp.c_row_major = c_layout == SPARTA_ROW_MAJOR;
            sp.N = n_cols; sp.w = (int32_t)A->w; sp.B_tail = nullptr;
            sp.clk = nullptr;
            if (A->n_steps[0] + A->n_steps[1] > 0) {
                if (prof) HIP_TRY(hipEventRecord(A->cev[0][0], st));
                if (A->has_tail && shard_rows == 0) {
                    if (int rc = ensure_scratch(&A->d_btail, &A->d_btail_bytes, (size_t)A->w * n_cols * sizeof(float))) return rc;
                    const int64_t row0 = ((A->cols - 1) / A->w) * A->w;
                    hipLaunchKernelGGL(vbs_tail_copy_kernel, dim3(32), dim3(kThreads), 0, st, dB, ldb, (int)(b_layout == SPARTA_ROW_MAJOR),
                                       row0, A->cols, (int)A->w, (int)n_cols, (float*)A->d_btail);
                    sp.B_tail = (const float*)A->d_btail;
                }
                const dim3 grid((unsigned)A->n_workers, (unsigned)n_nt);
                // heavier type first; the clock probe rides on the launch with more steps
                const int probe_ty = A->n_steps[1] >= A->n_steps[0] ? 1 : 0;
                for (int ty = 1; ty >= 0; ty--) {
                    if (A->n_steps[ty] == 0) continue;
                    sp.steps = shard_rows > 0 ? A->d_steps_g[ty] : A->d_steps[ty]; sp.worker_range = A->d_wrange[ty];
                    sp.clk = (prof && ty == probe_ty) ? A->d_clk : nullptr;
                    if (ty) {
                        if (shard_rows > 0) hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<false, true, true>), grid, dim3(kThreads), 0, st, sp);
                        else if (b_layout == SPARTA_ROW_MAJOR) hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<true, false, true>), grid, dim3(kThreads), 0, st, sp);
                        else hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<false, false, true>), grid, dim3(kThreads), 0, st, sp);
                    } else {
                        if (shard_rows > 0) hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<false, true, false>), grid, dim3(kThreads), 0, st, sp);
                        else if (b_layout == SPARTA_ROW_MAJOR) hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<true, false, false>), grid, dim3(kThreads), 0, st, sp);
                        else hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<false, false, false>), grid, dim3(kThreads), 0, st, sp);
                    }
                }
                if (prof) { HIP_TRY(hipEventRecord(A->cev[0][1], st)); A->class_ran[0] = true; }
            }
            const bool fix_launch = A->n_fix > 0 && !(accumulate && A->n_split == 0);   // C += 0 for the block-rows without blocks: nothing to launch
            const bool zero_launch = !A->zero_ranges.empty() && !accumulate;
            if (fix_launch || zero_launch) {
                if (prof) HIP_TRY(hipEventRecord(A->cev[1][0], st));
                if (fix_launch && A->n_big_fix > 0)
                    hipLaunchKernelGGL(vbs_spmm_f32_fixup_group_kernel,
                                       dim3((unsigned)A->n_big_fix, (unsigned)n_nt, (unsigned)((A->max_tile_slots + kFixGroup - 1) / kFixGroup)), dim3(kThreads), 0, st,
                                       A->d_fix, A->d_big_fix, A->d_fix_slots, (float*)A->d_ws, (int64_t)slab);
                if (fix_launch)
                    hipLaunchKernelGGL(vbs_spmm_f32_fixup_kernel, dim3((unsigned)A->n_fix, (unsigned)n_nt), dim3(kThreads), 0, st, A->d_fix,
                                       A->d_fix_slots, (const float*)A->d_ws, (int64_t)slab, Cout, ldc, (int)(c_layout == SPARTA_ROW_MAJOR),
                                       (int)(accumulate != 0));
                if (zero_launch) launch_zero_ranges(A, Cout, ldc, c_layout == SPARTA_ROW_MAJOR, n_cols, st);
                if (prof) { HIP_TRY(hipEventRecord(A->cev[1][1], st)); A->class_ran[1] = true; }
            }
            return SPARTA_OK;
        };
        // one launch per tile class (<=16 / <=32 / <=64 rows), one workgroup per tile
        auto run_class = [&](float* Cout, bool generic, bool prof) -> int {
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
                if (generic) { if (p.b_row_major) launch_tile_class<true, true>(c, p, st); else launch_tile_class<false, true>(c, p, st); }
                else { if (p.b_row_major) launch_tile_class<true, false>(c, p, st); else launch_tile_class<false, false>(c, p, st); }
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

        // ---- the block-rows kept as sparse rows (disjoint rows of C: order against the MFMA launches does not matter) ----
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
    hipLaunchKernelGGL(pack_blocks_kernel, dim3((unsigned)n_blocks, gy), dim3(kThreads), 0, (hipStream_t)stream, (const u32x4*)src, ids_dev,
                       (u32x4*)dst, vec);
    HIP_TRY(hipGetLastError());
    return SPARTA_OK;
}

}  // extern "C"
