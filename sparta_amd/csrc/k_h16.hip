// k_h16.hip -- fp16 / bf16 storage with fp32 accumulation: the LDS-staged and the direct (operands straight into the MFMA
// registers) persistent stream kernels, the 16-bit B-tail copy and the fp32 -> 16-bit conversion of the host-pointer path.
// Part of the device side of libsparta_amd.so; see vbs_device.hpp for the translation-unit map and DESIGN.md section 8.
#include "vbs_kernel_common.hpp"

using namespace sparta_dev;

namespace {

// =====================================================================================================
// 16-bit (fp16 / bf16) storage, fp32 accumulation: vbs_spmm_h16_stream_kernel<KP, MI2, BF16>
// Same persistent design, step records, plans and epilogue as the fp32 stream kernel; what changes is the operand path.
// * A is re-laid-out ONCE at sparta_vbs_create: every step's slice is a dense TM x KP chunk stored [k chunk of 8][row][8] (rows
//   past the tile zero) -- the order in which the direct kernel's lanes want it: lane (row, g) takes k chunk 2 q + g, so a wave
//   load is contiguous -- the chunks of a tile back to back.  The 16-bit MFMA wants 8 consecutive k of one row per lane; the
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
    const int ac = tid % (TM * CPC);                     // A: 16-byte chunk ac (+ 256 q) of the contiguous slice (clamped: duplicates are harmless)
    const int64_t ld_t = (int64_t)p.w;                   // leading dimension of B_tail
    const uint32_t voffB = (uint32_t)((bc * 8 + bj0 * p.ldb) * 2), voffBt = (uint32_t)((bc * 8 + (n0 + bj0) * ld_t) * 2);
    const uint32_t qstepB = (uint32_t)(CPP * p.ldb * 2), qstepBt = (uint32_t)(CPP * ld_t * 2);
    const int64_t n0off = (int64_t)n0 * p.ldb;
    const uint32_t lwB = (uint32_t)((bj0 * LDK + bc * 8) * 2);
    const uint32_t lwA = (uint32_t)((BSZ + (ac % TM) * LDK + (ac / TM) * 8) * 2);   // slice in memory: [k chunk][row][8] (chunk ac = row ac % TM, k chunk ac / TM)
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
        for (int q = 0; q < NAL; q++) *reinterpret_cast<u32x4*>(ldsb + lwA + (ST * STAGE + q * (kThreads / TM) * 8) * 2) = ra[q];
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
                        if (p.c_nt) {
#pragma unroll
                            for (int q = 0; q < 16; q++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 2);
                        } else {
#pragma unroll
                            for (int q = 0; q < 16; q++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 0);
                        }
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
// 16-bit storage without a workgroup-wide stage: vbs_spmm_h16_direct_kernel<KP, MI2, BF16, GATHERED>
// A 16-bit step has 64-256 cycles of MFMA per wave; in the LDS-staged kernel above what a step costs is its bookkeeping: the
// register -> LDS -> register round trip of BOTH operands, one workgroup barrier and the four waves waiting for each other
// (measured with the in-kernel timeline: ~1300 cycles per step with every byte cache-hot).  Here the four waves never meet
// (same scheme as vbs_spmm_f32_direct_kernel, k_f32_direct.hip):
//   * the packed A slices (row-major, k contiguous) already ARE the per-lane operand layout of v_mfma_f32_32x32x16: lane (lm, g)
//     needs 8 consecutive k of row lm, one 16-byte load; each wave loads the slice itself (the four waves share it through L1);
//   * B (column-major, k contiguous) is the same layout too, but loading it in MFMA shape -- lane (lm, g) its own column --
//     touches 32 different cache lines per wave instruction, and the texture addresser then sets the pace (round 1's form of
//     this kernel: 36.0 us on the flagship).  So a wave loads ITS 32 columns 4 (KP = 32) or 8 (KP = 64) lanes per column,
//     writes them to a wave-PRIVATE LDS image Bs[column][k] (+8 padding) and reads its fragments back with ds_read_b128:
//     program order and lgkmcnt are all the synchronisation there is.
// Loads run three steps ahead (A: four register sets, B: two staging sets + two LDS stages); step records: one window VGPR,
// constant-lane v_readlane.  Same plans, accumulator layout, epilogue and fix-up as the other stream kernels.
// =====================================================================================================
#ifndef SPARTA_H16_ACCA
#define SPARTA_H16_ACCA 1     // four-accumulator kernels: accumulators pinned to AGPRs (inline-assembly MFMA)
#endif
#ifndef SPARTA_H16_ILV
#define SPARTA_H16_ILV 1      /* two-tile (pair / hub) instantiations: the step's LDS writes and loads between its MFMAs, held there by scheduling barriers */
#endif
#ifndef SPARTA_H16_PROBE
#define SPARTA_H16_PROBE 0        /* developer probes, TIMING ONLY (results wrong): 1 no B loads, 2 no A loads, 4 no epilogue, 8 no LDS round trip, 64 every tile stores to the first rows of C, 128 every step reads the first slice of A, 256 every step reads the first rows of B, 512 one store per tile instead of 16 (non-temporal path) */
#endif

// DEEP: loads 7 steps ahead of their MFMAs instead of 3 (eight A sets, six live B staging sets, two record windows, rounds of eight steps).  A 16-bit
// step of a 32-wide block is two MFMAs (64 cycles) and ~300 cycles of a wave's time: three steps ahead are ~0.4 us, less than a loaded HBM access.
// TAIL = false: no block column hangs over the last row of B (cols % w == 0), the steps never read B_tail -- with the slices of A in step order
// (vbs_plan.cpp) a step then costs 30 scalar / vector instructions around its loads and MFMAs instead of 50, and ONE wave per SIMD issues one
// instruction per ~5 cycles: with neither A nor B loaded the flagship's kernel still took 11.9 of its 22.2 us (SPARTA_H16_PROBE = 3).
// WC = 64 (one-tile plans of 32-wide blocks, vbs_plan.cpp `wide16`): a wave owns 64 columns of the slab, the workgroup's wave pairs (0, 1) and (2, 3) are two
// SUB-WORKERS with their own step ranges -- two tiles per CU at a time, the slice of A through L1 twice instead of four times, and the ~20 instructions a
// step spends on records, addresses and waits pay for four MFMAs instead of two.  Nothing else changes: there is no barrier for the pairs to meet at.
// SLAB (with WC = 64): the four waves work on ONE tile again, 64 columns each -- a 256-column slab per workgroup (grid.y = N / 256), so that A is read once per 256
// columns of B instead of once per 128.  Runs on either plan (the two sub-worker ranges of a workgroup are adjacent); plans without split tiles only.
// SLAB with MI2 (round 3; 64-row tiles, the dense hub of a power-law matrix): FOUR accumulators per wave -- rows 0..31 / 32..63 x the wave's two groups of 32
// columns -- so that a step's slice of A (64 x KP) and its panel of B (KP x 64 per wave) feed 4 NK MFMAs instead of 2 NK: A is read once per 256 columns of B,
// B fragments are used twice.  One workgroup per CU (the 16-bit plans' own choice): the accumulators live in the upper half of the 512-register file.
// Split tiles are allowed: a wave writes its 64 x 64 piece into the slot images of the two 128-column slabs its workgroup covers, in the layout the
// fix-up kernel reads (four waves x 32 columns).
template <int KP, bool MI2, bool BF16, bool GATHERED, bool CSTAGE = false, bool DEEP = false, bool TAIL = true, int WC = 32, bool SLAB = false>
__global__ __launch_bounds__(kThreads, (MI2 && WC == 64) ? 1 : 2) void vbs_spmm_h16_direct_kernel(const StreamParams p) {
    static_assert(!SLAB || WC == 64, "256-column slabs are four 64-column waves");
    static_assert(!(CSTAGE && MI2), "the C ring holds 64 rows: tiles of <= 32 rows only");
    static_assert(WC == 32 || (WC == 64 && !CSTAGE && !DEEP && (!MI2 || SLAB)), "64-column waves: without ring, three steps ahead; two-tile form as 256-column slabs only");
    constexpr bool QUAD = MI2 && WC == 64;               // four accumulators per wave
    constexpr int NG = WC / 32;                          // column groups of 32 per wave
    constexpr int D = DEEP ? 7 : 3;                      // steps between a step's loads and its MFMAs
    constexpr int TN = kTN, TM = MI2 ? 64 : 32;
    constexpr int NK = KP / 16;                          // MFMAs (k groups of 16) per step and 32-row tile = 16-byte loads per lane and operand
    constexpr int NA = MI2 ? 2 : 1;
    constexpr int LPC = KP / 8;                          // lanes per column of a coalesced B load (16 bytes = 8 k each): 4 or 8
    constexpr int CPI = 64 / LPC;                        // columns per wave instruction: 16 or 8 (NK instructions cover the wave's 32)
    constexpr int RB = (KP + 8) * 2;                     // bytes per column of the LDS image (8 elements of padding: conflict-free b128)
    constexpr int WSTAGE = 32 * RB;                      // bytes per wave and stage
    __shared__ __attribute__((aligned(16))) char lds[4 * 2 * NG * WSTAGE + (CSTAGE ? 4 * kCRingFloats * 4 : 0)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 31, g = lane >> 5;
    const int n0 = blockIdx.y * (SLAB ? 2 * TN : TN);
    // WC = 64: sub-worker (wave >> 1) of this workgroup, wave (wave & 1) of its two; SLAB: the whole workgroup walks both sub-ranges (adjacent) or its one range
    const int worker = (WC == 64 && !SLAB) ? 2 * (int)blockIdx.x + __builtin_amdgcn_readfirstlane(wave >> 1) : (int)blockIdx.x;
    const int wv = (WC == 64 && !SLAB) ? (wave & 1) : wave;
    const int s_begin = (SLAB && p.sub_ranges) ? p.worker_range[4 * worker] : p.worker_range[2 * worker];
    const int n = ((SLAB && p.sub_ranges) ? p.worker_range[4 * worker + 3] : p.worker_range[2 * worker + 1]) - s_begin;
    if (n <= 0) return;
    clock_probe(p.clk, 0);
    // split-tile workspace: one image per 128-column slab; a 256-column workgroup of the four-accumulator form covers slabs 2 y and 2 y + 1 (waves 0, 1 / 2, 3)
    float* ws = p.ws + (int64_t)(QUAD ? 2 * (int)blockIdx.y + __builtin_amdgcn_readfirstlane(wave >> 1) : (int)blockIdx.y) * p.ws_slab_stride;
    const uint16_t* A16 = reinterpret_cast<const uint16_t*>(p.A);
    const uint16_t* B16 = reinterpret_cast<const uint16_t*>(p.B);
    const uint16_t* Bt16 = reinterpret_cast<const uint16_t*>(p.B_tail);

    // step records: one window of 8 (lane = 8 * record + field) per round of four steps, see vbs_spmm_f32_direct_kernel
    const int32_t* srec = reinterpret_cast<const int32_t*>(p.steps + s_begin);
    int vwin = srec[lane];
    int vwin1 = DEEP ? srec[64 + lane] : 0;              // DEEP: the following eight records (positions 8 .. 15)
    int vnext = 0;
#define field(pos, f) ((pos) < 8 ? __builtin_amdgcn_readlane(vwin, 8 * ((pos) & 7) + (f)) : __builtin_amdgcn_readlane(vwin1, 8 * ((pos) & 7) + (f)))
    enum { F_AOFF_LO = 0, F_AOFF_HI = 1, F_BROW = 2, F_H = 3, F_CROW = 4, F_FLAGS = 5, F_SLOT = 6, F_SHARD = 7 };

    const int bc = lane / LPC, bk = (lane % LPC) * 8;    // B load q: column CPI q + bc of the wave's 32, k = bk .. bk + 7
    const int64_t ld_t = (int64_t)p.w;                   // leading dimension of B_tail
    // this wave's 32 columns start at column nw of B / C: folded into the SCALAR bases, so that the 32-bit per-lane offsets only span 32 columns
    // (a column-major B or C with a leading dimension of 2^23 -- configs[4] on one GPU -- is 2 GB per 128 columns, beyond a 32-bit offset)
    const int nw = n0 + WC * __builtin_amdgcn_readfirstlane(wv);
    const uint32_t voffB = (uint32_t)((bc * p.ldb + bk) * 2);
    const uint32_t voffBt = (uint32_t)(((n0 + WC * wv + bc) * ld_t + bk) * 2);
    const uint32_t qstepB = (uint32_t)(CPI * p.ldb * 2), qstepBt = (uint32_t)(CPI * ld_t * 2);      // bytes from one column group to the next
    const uint32_t gstepB = (uint32_t)(32 * p.ldb * 2), gstepBt = (uint32_t)(32 * ld_t * 2);        // ... from one group of 32 columns to the next (WC = 64)
    const uint32_t voffA = (uint32_t)((g * TM + lm) * 16);     // slice in memory: [k chunk = 2 q + g][row][8]: a wave load is one (TM = 32) or two contiguous pieces
    const int64_t n0off = (int64_t)nw * p.ldb;
    const uint32_t voffC = p.c_row_major ? (uint32_t)((lm * p.ldc + 4 * g) * 4) : (uint32_t)((lm + (4 * g) * p.ldc) * 4);
    char* const ldsw = lds + wave * (2 * NG * WSTAGE);   // this wave's two stages (NG images of 32 columns each)
    const uint32_t lwB = (uint32_t)(bc * RB + bk * 2);   // write: column bc + CPI q
    const uint32_t lrB = (uint32_t)(lm * RB + 16 * g);   // read: column lm, k = 16 q + 8 g .. + 7

    CRing cr;                                            // CSTAGE: finished tiles wait here for whole aligned blocks of 32 rows (vbs_kernel_common.hpp)
    cr.ring = reinterpret_cast<float*>(lds + 4 * 2 * NG * WSTAGE) + wave * kCRingFloats;

    struct ASet { u32x4 a[NA][NK]; };
    struct BSet { u32x4 b[NG * NK]; };
    ASet as0, as1, as2, as3;                             // A fragments of the steps i mod 4
    BSet bs0, bs1;                                       // B staging (steps of even / odd index)

    // the slices of A are laid out in step order: one running offset, taken from the first record of the range
    int64_t g_aoff = ((int64_t)(uint32_t)field(0, F_AOFF_LO) | ((int64_t)field(0, F_AOFF_HI) << 32)) - (int64_t)TM * KP;
    uint32_t vo_cur = voffB;
    int32_t tail_prev = 0;
    auto issue_loads = [&](auto pos_tag, BSet& rb, ASet& ra) __attribute__((always_inline)) -> int32_t {
        constexpr int s = decltype(pos_tag)::value;      // position of the step's record in the window
        const int32_t flags = field(s, F_FLAGS);
        g_aoff += (int64_t)TM * KP;
        const int32_t tail = TAIL && (flags & STEP_TAIL) != 0;
        if (tail != tail_prev) {
            vo_cur = tail ? voffBt : voffB;
            asm volatile("" : "+v"(vo_cur));
            tail_prev = tail;
        }
        const int64_t gk0 = (SPARTA_H16_PROBE & 256) ? 0 : field(s, F_BROW);
        const uint16_t* bptr = tail ? Bt16 + gk0 : B16 + (GATHERED ? (int64_t)field(s, F_SHARD) * p.shard_stride : (int64_t)0) + gk0 + n0off;
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(bptr), 0, 0x7ffffff0, 0x00020000);
        const uint32_t qs = tail ? qstepBt : qstepB;
        const uint32_t gs = tail ? gstepBt : gstepB;
        if (SPARTA_H16_PROBE & 128) g_aoff = 0;
        if (!(SPARTA_H16_PROBE & 1)) {
#pragma unroll
            for (int c = 0; c < NG; c++)
#pragma unroll
                for (int q = 0; q < NK; q++) rb.b[c * NK + q] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo_cur, gs * c + qs * q, 0);
        }
        // (pair tiles, vbs_plan.cpp: a half of the slice whose block-row has no block in this column is zeros -- a descriptor of zero records returns them without a fetch)
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A16 + g_aoff), 0, (MI2 && (flags & STEP_LO_ABSENT)) ? 0 : 0x7ffffff0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A16 + g_aoff), 0, (MI2 && (flags & STEP_HI_ABSENT)) ? 0 : 0x7ffffff0, 0x00020000);
        if (!(SPARTA_H16_PROBE & 2) && !((SPARTA_H16_PROBE & 16) && wave != 0) && !((SPARTA_H16_PROBE & 32) && (wave & 1))) {   // 16: only wave 0 loads A; 32: waves 0 and 2
#pragma unroll
            for (int mi = 0; mi < NA; mi++)
#pragma unroll
                for (int q = 0; q < NK; q++) ra.a[mi][q] = __builtin_amdgcn_raw_buffer_load_b128(mi == 0 ? rA : rA1, voffA, (uint32_t)((2 * q * TM + 32 * mi) * 16), 0);
        }
        return flags;
    };
    auto write_b = [&](auto stage_tag, const BSet& rb) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_tag)::value;
#pragma unroll
        for (int c = 0; c < NG; c++)
#pragma unroll
            for (int q = 0; q < NK; q++) *reinterpret_cast<u32x4*>(ldsw + lwB + (ST * NG + c) * WSTAGE + q * CPI * RB) = rb.b[c * NK + q];
    };

    f32x16 acc0, acc1, acc2, acc3;                        // QUAD: acc0 / acc1 = rows 0..31 / 32..63 of the first 32 columns, acc2 / acc3 of the second
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; acc2[r] = 0.0f; acc3[r] = 0.0f; }
    if constexpr (SPARTA_H16_ACCA && QUAD) asm volatile("" : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3));
    // The four-accumulator instantiations run one work-group per CU (up to 512 registers a lane: 256 VGPRs + 256 AGPRs).  Left to itself the register allocator keeps the
    // accumulators in VGPRs between steps and copies all of them into AGPRs and back around every step's MFMAs (16 x NA x NG v_accvgpr_write + as many reads per
    // step).  ACC_A pins them: the MFMA is issued as inline assembly whose accumulator operand is constrained to the AGPRs ("+a"), so the accumulators never leave
    // them until the tile's epilogue.  The hazard recogniser does not look inside inline assembly: the wait states an MFMA result needs before a VALU read
    // (v_accvgpr_read in the epilogue) and a VALU write needs before an MFMA read (the zeroing) are written out as s_nop below.
    constexpr bool ACC_A = SPARTA_H16_ACCA && QUAD;
    auto mfma = [&](const u32x4& bf, const u32x4& af, f32x16& acc) __attribute__((always_inline)) {
        if constexpr (ACC_A) {
            // (s_nop 1: a fragment the allocator parked in AGPRs comes back through v_accvgpr_read right in front of the MFMA: a VALU write needs two wait states
            // before an MFMA reads the register)
            if constexpr (BF16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(bf), "v"(af));
            else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(bf), "v"(af));
        } else {
            if constexpr (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf), __builtin_bit_cast(bf16x8, af), acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf), __builtin_bit_cast(f16x8, af), acc, 0, 0, 0);
        }
    };

    int32_t fq0 = 0, fq1 = 0, fq2 = 0, fq_new = 0;
    // one step (index i mod 4 = u): fragments of B from LDS stage u & 1, the next step's panel into the other stage, the MFMAs, the
    // tile epilogue if it ends here, then staging set nb and A set na are refilled with step i + D (wb = the set that holds step i + 1: the same as nb for D = 3)
    auto step = [&](auto u_tag, int32_t flags, ASet& wa, BSet& wb, BSet& nb, ASet& na) __attribute__((always_inline)) {
        constexpr int i = decltype(u_tag)::value;
        constexpr int PAR = i & 1;
        u32x4 fb[NG * NK];
        if (SPARTA_H16_PROBE & 8) {
#pragma unroll
            for (int q = 0; q < NG * NK; q++) fb[q] = wb.b[q];
        } else {
#pragma unroll
            for (int c = 0; c < NG; c++)
#pragma unroll
                for (int q = 0; q < NK; q++) fb[c * NK + q] = *reinterpret_cast<const u32x4*>(ldsw + lrB + (PAR * NG + c) * WSTAGE + q * 32);
            if constexpr (!(SPARTA_H16_ILV && MI2 && !DEEP && !TAIL)) write_b(std::integral_constant<int, 1 - PAR>{}, wb);         // W(i + 1)  (two-tile instantiations: between the MFMAs, below)
        }
        constexpr bool ILV = SPARTA_H16_ILV && MI2 && !DEEP && !TAIL && !(SPARTA_H16_PROBE & 8);
        if constexpr (!ILV) {
#pragma unroll
            for (int q = 0; q < NK; q++) {
                mfma(fb[q], wa.a[0][q], acc0);
                if constexpr (MI2) mfma(fb[q], wa.a[1][q], acc1);
                if constexpr (WC == 64 && !MI2) mfma(fb[NK + q], wa.a[0][q], acc1);   // the wave's second 32 columns (acc1: free in the one-tile kernel)
                if constexpr (QUAD) { mfma(fb[NK + q], wa.a[0][q], acc2); mfma(fb[NK + q], wa.a[1][q], acc3); }
            }
            fq_new = issue_loads(std::integral_constant<int, i + D>{}, nb, na);   // G(i + D)
        } else {
            // ONE wave per SIMD here: nothing but this wave's own instruction order overlaps its MFMAs with the rest of its step.  An MFMA occupies the pipe for 32
            // cycles and takes four to issue: the instruction behind it in program order issues into the other 28.  So: MFMA, one LDS write of the next step's panel
            // (W(i + 1)) or one or two loads of step i + D, MFMA, ... -- kept in that order by scheduling barriers (it is one basic block: the scheduler would
            // cluster the MFMAs again).  TAIL = false instantiations only (no B_tail switch: no branch inside the step).
            constexpr int NM = NK * NA * NG;                 // MFMAs of the step: 4 (pair tiles of 32-wide blocks), 8, 16 (hub tiles)
            constexpr int NW = NG * NK, NLB = NG * NK, NLA = NA * NK;
            const int32_t flags_n = field(i + D, F_FLAGS);
            g_aoff += (int64_t)TM * KP;
            if (SPARTA_H16_PROBE & 128) g_aoff = 0;
            const int64_t gk0 = (SPARTA_H16_PROBE & 256) ? 0 : field(i + D, F_BROW);
            const uint16_t* bptr = B16 + (GATHERED ? (int64_t)field(i + D, F_SHARD) * p.shard_stride : (int64_t)0) + gk0 + n0off;
            const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(bptr), 0, 0x7ffffff0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A16 + g_aoff), 0, (flags_n & STEP_LO_ABSENT) ? 0 : 0x7ffffff0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A16 + g_aoff), 0, (flags_n & STEP_HI_ABSENT) ? 0 : 0x7ffffff0, 0x00020000);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NM>([&](auto t_tag) __attribute__((always_inline)) {
                constexpr int t = decltype(t_tag)::value;
                constexpr int q = t / (NA * NG), c = (t / NA) % NG, mi = t % NA;       // k group, column group, row half
                f32x16& acc = (c == 0) ? (mi == 0 ? acc0 : acc1) : (mi == 0 ? acc2 : acc3);
                mfma(fb[c * NK + q], wa.a[mi][q], acc);
                __builtin_amdgcn_sched_barrier(0);
                // what goes behind MFMA t: first the NW writes of W(i + 1), then the loads of step i + D spread over the remaining MFMAs
                constexpr int slots = NM;
                constexpr int per_w = (NW + slots / 2 - 1) / (slots / 2);             // writes per slot in the first half
                if constexpr (t < slots / 2) {
#pragma unroll
                    for (int x = t * per_w; x < (t + 1) * per_w && x < NW; x++)
                        *reinterpret_cast<u32x4*>(ldsw + lwB + ((1 - PAR) * NG + x / NK) * WSTAGE + (x % NK) * CPI * RB) = wb.b[x];
                } else {
                    constexpr int h = t - slots / 2, nh = slots - slots / 2;
                    constexpr int lb0 = h * NLB / nh, lb1 = (h + 1) * NLB / nh, la0 = h * NLA / nh, la1 = (h + 1) * NLA / nh;
#pragma unroll
                    for (int x = lb0; x < lb1; x++) nb.b[x] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo_cur, gstepB * (x / NK) + qstepB * (x % NK), 0);
#pragma unroll
                    for (int x = la0; x < la1; x++) na.a[x / NK][x % NK] = __builtin_amdgcn_raw_buffer_load_b128((x / NK) == 0 ? rA : rA1, voffA, (uint32_t)((2 * (x % NK) * TM + 32 * (x / NK)) * 16), 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            fq_new = flags_n;
        }
        if ((flags & STEP_LAST) && !(SPARTA_H16_PROBE & 4)) {
            // the last MFMA's 8 passes + 3 before its accumulator may be read.  The accumulators are operands of the statement: their reads below depend on it
            // (without that the compiler hoists the v_accvgpr_reads common to the branches of the epilogue above it, right behind the last MFMA)
            if constexpr (ACC_A) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3));
            if (flags & STEP_SPLIT) {
                const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(ws + (int64_t)field(i, F_SLOT) * SK_SLOT_FLOATS, 0, SK_SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    if constexpr (QUAD) {               // this wave's 64 columns are the places of waves 2 (wv & 1) and 2 (wv & 1) + 1 in its slab's image (four waves x 32 columns)
                        const uint32_t vt = (uint32_t)((2 * (wv & 1)) * 64 + lane) * 4u;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc0[q]), rW, vt, (uint32_t)(q * kThreads * 4), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc2[q]), rW, vt, (uint32_t)(q * kThreads * 4 + 256), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc1[q]), rW, vt, (uint32_t)((16 + q) * kThreads * 4), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc3[q]), rW, vt, (uint32_t)((16 + q) * kThreads * 4 + 256), 0);
                    } else if constexpr (WC == 64) {    // the image is laid out for four waves x 32 columns: this wave fills the places of waves 2 wv and 2 wv + 1; rows 32..63: none
                        const uint32_t vt = (uint32_t)((2 * wv) * 64 + lane) * 4u;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc0[q]), rW, vt, (uint32_t)(q * kThreads * 4), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc1[q]), rW, vt, (uint32_t)(q * kThreads * 4 + 256), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(0u, rW, vt, (uint32_t)((16 + q) * kThreads * 4), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(0u, rW, vt, (uint32_t)((16 + q) * kThreads * 4 + 256), 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc0[q]), rW, (uint32_t)tid * 4u, (uint32_t)(q * kThreads * 4), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc1[q]), rW, (uint32_t)tid * 4u, (uint32_t)((16 + q) * kThreads * 4), 0);
                    }
                }
            } else if (CSTAGE) {
                if constexpr (CSTAGE) cr.park(p, nw, lm, g, voffC, acc0, field(i, F_CROW), flags & 0xffff);
            } else {
                const int mt = flags & 0xffff;
                const int64_t c_row = (SPARTA_H16_PROBE & 64) ? 0 : field(i, F_CROW);
                float* cbase = p.c_row_major ? p.C + c_row * p.ldc + nw : p.C + c_row + (int64_t)nw * p.ldc;
                const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, 0x7ffffff0, 0x00020000);
                const uint32_t jstep = p.c_row_major ? 4u : (uint32_t)p.ldc * 4u;
                const uint32_t mistep = p.c_row_major ? (uint32_t)p.ldc * 128u : 128u;
#pragma unroll
                for (int cg = 0; cg < NG; cg++) {                              // the wave's groups of 32 columns (WC = 64: two), each with its own scalar base, so that
                    // the per-lane offsets keep spanning 32 columns of C
                    const __amdgpu_buffer_rsrc_t rCm = cg == 1
                        ? __builtin_amdgcn_make_buffer_rsrc(p.c_row_major ? cbase + 32 : cbase + 32 * p.ldc, 0, 0x7ffffff0, 0x00020000) : rC;
#pragma unroll
                    for (int mi = 0; mi < NA; mi++) {                          // rows 0..31 / 32..63 of the tile (MI2)
                        if (mi * 32 + lm < mt) {
                            float v[16];
#pragma unroll
                            for (int q = 0; q < 16; q++) v[q] = QUAD ? (cg == 0 ? (mi == 0 ? acc0[q] : acc1[q]) : (mi == 0 ? acc2[q] : acc3[q]))
                                                                     : ((cg + mi) == 0 ? acc0[q] : acc1[q]);
                            if (p.accumulate) {
                                uint32_t old[16];
#pragma unroll
                                for (int q = 0; q < 16; q++) old[q] = __builtin_amdgcn_raw_buffer_load_b32(rCm, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 0);
#pragma unroll
                                for (int q = 0; q < 16; q++) v[q] += __uint_as_float(old[q]);
                            }
                            if (p.c_nt) {
#pragma unroll
                                for (int q = 0; q < ((SPARTA_H16_PROBE & 512) ? 1 : 16); q++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rCm, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 2);
                            } else {
#pragma unroll
                                for (int q = 0; q < 16; q++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rCm, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)mi * mistep, 0);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; q++) { acc0[q] = 0.0f; acc1[q] = 0.0f; if constexpr (QUAD) { acc2[q] = 0.0f; acc3[q] = 0.0f; } }
            if constexpr (ACC_A) asm volatile("s_nop 4" : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3));
        }
    };

    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using c2 = std::integral_constant<int, 2>;
    using c3 = std::integral_constant<int, 3>;
    // the window of steps [i, i + 4): requested one round earlier (4 x NK (NG + NA) loads are issued in between and memory returns in order)
    constexpr int kSwapCnt = 2 * NK * (NG + NA) < 63 ? 2 * NK * (NG + NA) : 63;      // the loads of two steps may stay in flight across the swap
    auto window_swap = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(vnext) : "n"(kSwapCnt) : "memory");
        vwin = vnext;
    };
    auto window_request = [&](int i) __attribute__((always_inline)) {
        const int32_t* nb = srec + (int64_t)(i + 4) * 8 + lane;
        asm volatile("global_load_dword %0, %1, off" : "=&v"(vnext) : "v"(nb) : "memory");
    };
    if constexpr (DEEP) {
        // step i: LDS stage i & 1, A set i & 7; writes out staging set (i + 1) & 7 and fills staging set and A set (i + 7) & 7 with step i + 7.  Records:
        // window 0 = [8 k, 8 k + 8), window 1 = the next eight (step i + 7 is position 7 of window 0 or 0 .. 6 of window 1); the window after that
        // is requested at the head of a round and swapped in at the head of the next (8 steps' loads are issued in between, memory returns in order:
        // the wait below leaves the loads of seven steps in flight)
        constexpr int kLoadsAhead = D * NK * (1 + NA);
        static_assert(kLoadsAhead <= 63, "vmcnt holds 6 bits");
        ASet asd[8];
        BSet bsd[8];
        int32_t fqd[8];
        auto swap8 = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(vnext) : "n"(kLoadsAhead) : "memory");
            vwin = vwin1;
            vwin1 = vnext;
        };
        auto request8 = [&](int i) __attribute__((always_inline)) {
            const int32_t* nb = srec + (int64_t)(i + 16) * 8 + lane;
            asm volatile("global_load_dword %0, %1, off" : "=&v"(vnext) : "v"(nb) : "memory");
        };
        static_for<0, 7>([&](auto j) __attribute__((always_inline)) { fqd[j] = issue_loads(j, bsd[j], asd[j]); });
        write_b(c0{}, bsd[0]);                           // W(0)
        const int n8 = n & ~7;
        for (int i = 0; i < n8; i += 8) {
            if (i > 0) swap8();
            request8(i);
            static_for<0, 8>([&](auto u) __attribute__((always_inline)) {
                constexpr int U = decltype(u)::value;
                step(u, fqd[U], asd[U], bsd[(U + 1) & 7], bsd[(U + 7) & 7], asd[(U + 7) & 7]);
                fqd[(U + 7) & 7] = fq_new;
            });
        }
        if (n > n8) {
            if (n8 > 0) swap8();
            const int r = n - n8;
            static_for<0, 7>([&](auto u) __attribute__((always_inline)) {
                constexpr int U = decltype(u)::value;
                if (U < r) {
                    step(u, fqd[U], asd[U], bsd[(U + 1) & 7], bsd[(U + 7) & 7], asd[(U + 7) & 7]);
                    fqd[(U + 7) & 7] = fq_new;
                }
            });
        }
    } else {
        fq0 = issue_loads(c0{}, bs0, as0);
        fq1 = issue_loads(c1{}, bs1, as1);
        write_b(c0{}, bs0);                                  // W(0)
        fq2 = issue_loads(c2{}, bs0, as2);
        const int n4 = n & ~3;
        for (int i = 0; i < n4; i += 4) {
            if (i > 0) window_swap();
            window_request(i);
            step(c0{}, fq0, as0, bs1, bs1, as3);
            fq0 = fq1; fq1 = fq2; fq2 = fq_new;
            step(c1{}, fq0, as1, bs0, bs0, as0);
            fq0 = fq1; fq1 = fq2; fq2 = fq_new;
            step(c2{}, fq0, as2, bs1, bs1, as1);
            fq0 = fq1; fq1 = fq2; fq2 = fq_new;
            step(c3{}, fq0, as3, bs0, bs0, as2);
            fq0 = fq1; fq1 = fq2; fq2 = fq_new;
        }
        if (n > n4) {
            if (n4 > 0) window_swap();
            step(c0{}, fq0, as0, bs1, bs1, as3);
            fq0 = fq1; fq1 = fq2; fq2 = fq_new;
            if (n - n4 >= 2) {
                step(c1{}, fq0, as1, bs0, bs0, as0);
                fq0 = fq1; fq1 = fq2; fq2 = fq_new;
            }
            if (n - n4 == 3) step(c2{}, fq0, as2, bs1, bs1, as1);
        }
    }
    if constexpr (CSTAGE) cr.flush(p, nw, lm, g, voffC, true);
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

// the column tail of a 16-bit product whose n_cols is not a multiple of 128: C[:, col0 + j] (+)= Ct[:, j], j < n_t (Ct column-major, ld = rows)
__global__ __launch_bounds__(kThreads) void vbs_col_tail_merge_kernel(const float* Ct, int64_t rows, float* C, int64_t ldc, int c_row_major, int col0,
                                                                      int n_t, int accumulate) {
    const int64_t total = rows * (int64_t)n_t;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        // column-major C: consecutive threads walk a column (coalesced on both sides); row-major C: consecutive threads walk a row of C
        const int64_t r = c_row_major ? e / n_t : e % rows;
        const int j = (int)(c_row_major ? e % n_t : e / rows);
        float* dst = c_row_major ? C + r * ldc + col0 + j : C + r + (int64_t)(col0 + j) * ldc;
        const float v = Ct[r + (int64_t)j * rows];
        *dst = accumulate ? *dst + v : v;
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
// Which 16-bit kernel: the no-barrier kernel (vbs_spmm_h16_direct_kernel) for every shape -- flagship 23.0 us against 36.7 for the LDS-staged
// kernel, 64 x 64 blocks 23.7 against 27.8, (w 64, 32 rows) 22.9 against 30.9, N = 256 46 against 53-71; SPARTA_H16_PATH=lds selects the
// LDS-staged kernel (A/B runs, tests).  Read per launch: scripts flip it between timings.
bool h16_direct(int kp, bool mi2) {
    const char* e = std::getenv("SPARTA_H16_PATH");
    if (e && e[0] == 'l') return false;
    return true;
}
// SPARTA_H16_AHEAD=7: the <= 32-row tiles of 32-wide blocks load seven steps ahead instead of three (DEEP).  Measured on the flagship, interleaved
// runs: 22.9 -> 22.1 us while a step still cost 50 instructions, 21.3 -> 21.8 us since it costs 26 (TAIL = false, slices in step order); banded 200k
// through the C ring 30.7 -> 31.4 us.  Off by default; read per launch (scripts flip it between timings).
bool h16_ahead7() { const char* e = std::getenv("SPARTA_H16_AHEAD"); return e && atoi(e) == 7; }
template <int KP, bool MI2, bool GATHERED, bool CSTAGE, bool DEEP, int WC = 32>
void launch_h16_direct_v(bool bf16, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (sp.B_tail != nullptr) {                          // some step reads the zero-padded copy of the last rows of B
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, MI2, true, GATHERED, CSTAGE, DEEP, true, WC>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, MI2, false, GATHERED, CSTAGE, DEEP, true, WC>), grid, dim3(kThreads), 0, st, sp);
    } else {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, MI2, true, GATHERED, CSTAGE, DEEP, false, WC>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, MI2, false, GATHERED, CSTAGE, DEEP, false, WC>), grid, dim3(kThreads), 0, st, sp);
    }
}
template <int KP, bool MI2>
void launch_h16_direct(bool bf16, bool gathered, bool c_stage, bool wide, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if constexpr (!MI2 && KP == 32) {
        if (wide) {                                      // the plan holds two sub-worker ranges per workgroup (vbs_plan.cpp, wide16): this kernel and no other
            if (gathered) launch_h16_direct_v<KP, false, true, false, false, 64>(bf16, grid, st, sp);
            else launch_h16_direct_v<KP, false, false, false, false, 64>(bf16, grid, st, sp);
            return;
        }
    }
    if constexpr (!MI2) {
        if (c_stage && !gathered) {                      // tiles of arbitrary height, column-major C: through the C ring (short tiles: seven steps ahead
            launch_h16_direct_v<KP, false, false, true, false>(bf16, grid, st, sp);     // measured slower there, banded 200k 31.4 against 30.7 us)
        } else if (gathered) {
            launch_h16_direct_v<KP, false, true, false, false>(bf16, grid, st, sp);
        } else {
            constexpr bool kCanDeep = KP == 32;          // (64-wide blocks: eight A sets + eight staging sets are 256 registers)
            if (kCanDeep && h16_ahead7()) launch_h16_direct_v<KP, false, false, false, kCanDeep>(bf16, grid, st, sp);
            else launch_h16_direct_v<KP, false, false, false, false>(bf16, grid, st, sp);
        }
    } else {
        if (gathered) launch_h16_direct_v<KP, true, true, false, false>(bf16, grid, st, sp);
        else launch_h16_direct_v<KP, true, false, false, false>(bf16, grid, st, sp);
    }
}
template <int KP, bool MI2>
void launch_h16(bool bf16, bool gathered, bool c_stage, bool wide, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (h16_direct(KP, MI2) || wide) { launch_h16_direct<KP, MI2>(bf16, gathered, c_stage, wide, grid, st, sp); return; }
    if (h16_deep()) launch_h16_d<KP, MI2, true>(bf16, gathered, grid, st, sp);
    else launch_h16_d<KP, MI2, false>(bf16, gathered, grid, st, sp);
}

}  // namespace

namespace sparta_dev {

bool h16_uses_direct_kernel(int kp, bool mi2) { return h16_direct(kp, mi2); }

void launch_h16_stream(int kp, bool mi2, bool bf16, bool gathered, bool c_stage, bool wide, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (kp == 64) { if (mi2) launch_h16<64, true>(bf16, gathered, false, false, grid, st, sp); else launch_h16<64, false>(bf16, gathered, c_stage, false, grid, st, sp); }
    else { if (mi2) launch_h16<32, true>(bf16, gathered, false, false, grid, st, sp); else launch_h16<32, false>(bf16, gathered, c_stage && !wide, wide, grid, st, sp); }
}

void launch_h16_slab256(bool bf16, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (sp.B_tail != nullptr) {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<32, false, true, false, false, false, true, 64, true>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<32, false, false, false, false, false, true, 64, true>), grid, dim3(kThreads), 0, st, sp);
    } else {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<32, false, true, false, false, false, false, 64, true>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<32, false, false, false, false, false, false, 64, true>), grid, dim3(kThreads), 0, st, sp);
    }
}

// 64-row tiles over 256-column slabs, four accumulators per wave (QUAD): grid = (workers, n_cols / 256).  KP = 64: 64-wide blocks (the hub of a power-law matrix);
// KP = 32: the pair tiles of 32-wide blocks (vbs_plan.cpp)
template <int KP>
static void launch_h16_quad_t(bool bf16, bool gathered, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (gathered) {                                      // (a gathered B has no partial last block column: cols = n_shards * shard_rows, shard_rows % w == 0)
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, true, true, true, false, false, false, 64, true>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, true, false, true, false, false, false, 64, true>), grid, dim3(kThreads), 0, st, sp);
    } else if (sp.B_tail != nullptr) {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, true, true, false, false, false, true, 64, true>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, true, false, false, false, false, true, 64, true>), grid, dim3(kThreads), 0, st, sp);
    } else {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, true, true, false, false, false, false, 64, true>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_h16_direct_kernel<KP, true, false, false, false, false, false, 64, true>), grid, dim3(kThreads), 0, st, sp);
    }
}
void launch_h16_quad(int kp, bool bf16, bool gathered, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (kp == 64) launch_h16_quad_t<64>(bf16, gathered, grid, st, sp);
    else launch_h16_quad_t<32>(bf16, gathered, grid, st, sp);
}

void launch_tail_copy_h16(hipStream_t st, const uint16_t* B, int64_t ldb, int64_t row0, int64_t cols, int w, int N, uint16_t* B_tail) {
    hipLaunchKernelGGL(vbs_tail_copy_h16_kernel, dim3(32), dim3(kThreads), 0, st, B, ldb, row0, cols, w, N, B_tail);
}

void launch_col_tail_merge(hipStream_t st, const float* Ct, int64_t rows, float* C, int64_t ldc, int c_row_major, int col0, int n_t, int accumulate) {
    const int64_t total = rows * (int64_t)n_t;
    const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (total + kThreads - 1) / kThreads));
    hipLaunchKernelGGL(vbs_col_tail_merge_kernel, dim3(g), dim3(kThreads), 0, st, Ct, rows, C, ldc, c_row_major, col0, n_t, accumulate);
}

void launch_convert_h16(bool bf16, hipStream_t st, const float* src, int64_t ld_in, int64_t rows, int64_t n_cols, uint16_t* dst, int64_t ld_out) {
    if (bf16) hipLaunchKernelGGL((vbs_convert_h16_kernel<true>), dim3(1024), dim3(kThreads), 0, st, src, ld_in, rows, n_cols, dst, ld_out);
    else hipLaunchKernelGGL((vbs_convert_h16_kernel<false>), dim3(1024), dim3(kThreads), 0, st, src, ld_in, rows, n_cols, dst, ld_out);
}

}  // namespace sparta_dev
