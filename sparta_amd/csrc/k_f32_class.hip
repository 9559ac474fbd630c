// k_f32_class.hip -- per-class fp32 MFMA kernels (one workgroup per row tile) and the exact-order parity kernel.
// Part of the device side of libsparta_amd.so; see vbs_device.hpp for the translation-unit map and DESIGN.md section 3.
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
#include "vbs_kernel_common.hpp"

using namespace sparta_dev;

namespace {

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

// ---- exact-order kernel (parity aid) -----------------------------------------------------------------
// One thread per element of C; the sum runs block by block, k ascending, with an UNFUSED multiply and
// add -- the operation order and rounding of the reference's loop nest (src/general/vbr.cpp:358-363,
// compiled for baseline x86-64: no FMA).  Bit-identical to VBR::multiply for finite inputs.
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

template <int MF, int WM, int WN, int MI, int NI, int KP, bool BRM, bool GENERIC>
void launch_class(const SpmmParams& p, hipStream_t st) {
    if (p.n_tiles == 0) return;
    const int64_t grid = (int64_t)p.n_tiles * p.n_ntiles;
    hipLaunchKernelGGL((vbs_spmm_f32_kernel<MF, WM, WN, MI, NI, KP, BRM, GENERIC>), dim3((unsigned)grid), dim3(kThreads), 0, st, p);
}

template <bool BRM, bool GENERIC>
void launch_tile_class(int c, const SpmmParams& p, hipStream_t st) {
    switch (c) {
        case 0: launch_class<16, 1, 4, 1, 2, kKP, BRM, GENERIC>(p, st); break;   // <=16 x 128, 16x16x4 MFMA
        case 1: launch_class<32, 1, 4, 1, 1, kKP, BRM, GENERIC>(p, st); break;   // <=32 x 128
        default: launch_class<32, 2, 2, 1, 2, kKP, BRM, GENERIC>(p, st); break;  // <=64 x 128
    }
}

}  // namespace

namespace sparta_dev {

void launch_f32_class(int cls, bool b_row_major, bool generic, const SpmmParams& p, hipStream_t st) {
    if (generic) { if (b_row_major) launch_tile_class<true, true>(cls, p, st); else launch_tile_class<false, true>(cls, p, st); }
    else { if (b_row_major) launch_tile_class<true, false>(cls, p, st); else launch_tile_class<false, false>(cls, p, st); }
}

void launch_f32_exact(unsigned n_brows, hipStream_t st, const BlockRowDesc* rows, const int32_t* jab, const float* A, const float* B, float* C,
                      int64_t ldb, int64_t ldc, int64_t cols, int N, int w, int b_row_major, int c_row_major, int accumulate, int64_t shard_rows,
                      int64_t shard_stride) {
    hipLaunchKernelGGL(vbs_spmm_f32_exact_kernel, dim3(n_brows), dim3(kThreads), 0, st, rows, jab, A, B, C, ldb, ldc, cols, N, w, b_row_major,
                       c_row_major, accumulate, shard_rows, shard_stride);
}

}  // namespace sparta_dev
