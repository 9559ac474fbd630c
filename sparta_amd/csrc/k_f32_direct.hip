// k_f32_direct.hip -- fp32 stream kernel without a workgroup-wide stage: A arrives in MFMA fragment order straight from global memory,
// every wave converts its own 32 columns of the B panel in a wave-private LDS image; no barrier.  Part of the device side of
// libsparta_amd.so; see vbs_device.hpp for the translation-unit map and DESIGN.md section 3.2.
#include "vbs_kernel_common.hpp"

using namespace sparta_dev;

namespace {

// =====================================================================================================
// vbs_spmm_f32_direct_kernel: the one-tile fp32 stream kernel with NO workgroup-wide stage and NO barrier.
//
// A step multiplies a <= 32-row x 32 slice of A with a 32 x 128 panel of B; wave v owns columns [32v, 32v+32).  No element of
// the panel is used by more than ONE wave: the only thing the four waves share is the 4 KB slice of A.  So
//   * A arrives in MFMA fragment order straight from global memory into registers: the host lays every step's slice out as
//     a_frag [j = 0..3][g][row][4] = A[row][k = 16 g + 4 j + e] (k may be summed in any order as long as both operands agree), each
//     wave loads it itself with four contiguous 1 KB loads (L1 / L2 hits for three of the four waves);
//   * B is column-major (k contiguous): the MFMA wants lane (i = lane & 31, g = lane >> 5) to hold 16 k of column i, but a load
//     in that shape touches 32 different 128-byte lines per instruction and the texture addresser, not the MFMA pipe, then sets
//     the pace (measured on the flagship: 60.4 us; the same bytes loaded 8 lanes per line: 50.6 us).  So every wave loads ITS 32
//     columns 8 lanes per line, writes them to a wave-PRIVATE LDS image Bs[column][k] (+4 padding) and reads its fragments back
//     with ds_read_b128: a layout conversion inside one wave -- program order and lgkmcnt are all the synchronisation there is.
// The waves of a workgroup never meet: each walks the worker's step list on its own (records through v_readlane as in
// vbs_spmm_f32_stream_kernel), loads three steps ahead (A: four register sets, B: two staging sets + two LDS stages), same
// epilogue and same workspace images for split tiles, so plans, fix-up kernel and step records are shared with the LDS kernel.
// Column-major B only, no gathered B; tiles of <= 32 rows (the one-tile plan).
// =====================================================================================================
// CSTAGE (column-major C, tiles of arbitrary height): finished tiles are parked in the C ring (vbs_kernel_common.hpp, CRing) and stored as aligned blocks.
// TAIL = false: no block column hangs over the last row of B (cols % w == 0, B_tail is never read): the choice between B and B_tail -- a dozen scalar
// instructions per step -- folds away (see vbs_spmm_h16_direct_kernel).
#ifndef SPARTA_DIRECT_WAVES
#define SPARTA_DIRECT_WAVES 2      /* workgroups per CU the register budget allows (developer builds: 3 with SPARTA_WORKERS_PER_CU=3) */
#endif
template <bool CSTAGE, bool TAIL = true>
__global__ __launch_bounds__(kThreads, SPARTA_DIRECT_WAVES) void vbs_spmm_f32_direct_kernel(const StreamParams p) {
    constexpr int TN = kTN, LDBW = 36;                  // Bs[column][k], 32 k + 4 padding: conflict-free ds_read_b128 / ds_write_b128
    constexpr int WSTAGE = 32 * LDBW;                   // floats per wave and stage
    __shared__ __attribute__((aligned(16))) float lds[4 * 2 * WSTAGE + (CSTAGE ? 4 * kCRingFloats : 0)];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 31, g = lane >> 5;
    const int n0 = blockIdx.y * TN;
    const int s_begin = p.worker_range[2 * blockIdx.x];
    const int n = p.worker_range[2 * blockIdx.x + 1] - s_begin;
    if (n <= 0) return;
    clock_probe(p.clk, 0);
    float* ws = p.ws + (int64_t)blockIdx.y * p.ws_slab_stride;

    const int32_t* srec = reinterpret_cast<const int32_t*>(p.steps + s_begin);
    // Step records: ONE VGPR holds the 8 records [4k, 4k + 8) (lane = 8 * record + field) while steps 4k .. 4k + 3 run -- they need the
    // records of steps i (epilogue) and i + 3 (loads), positions 0 .. 6 of the window.  The loop is unrolled by four, so every field is
    // ONE v_readlane with a constant lane (the two-batch scheme of the LDS kernels costs two readlanes and a select per field: VALU work
    // that takes the pipe away from the MFMAs).  The next window is requested at step 4k and swapped in at step 4k + 4.
    int vwin = srec[lane];
    int vnext = 0;
#define field(pos, f) __builtin_amdgcn_readlane(vwin, 8 * (pos) + (f))
    enum { F_AOFF_LO = 0, F_AOFF_HI = 1, F_BROW = 2, F_H = 3, F_CROW = 4, F_FLAGS = 5, F_SLOT = 6, F_SHARD = 7 };

    // per-lane constants.  B load q (0..3): column 8 q + (lane >> 3) of the wave's 32, k = 4 (lane & 7) .. + 3
    const int bc = lane >> 3, bk = (lane & 7) * 4;
    const int64_t ld_t = (int64_t)p.w;                                            // leading dimension of B_tail (column-major, w rows)
    const uint32_t voffB = (uint32_t)(((32 * wave + bc) * p.ldb + bk) * 4);
    const uint32_t voffBt = (uint32_t)(((n0 + 32 * wave + bc) * ld_t + bk) * 4);
    const uint32_t qstepB = (uint32_t)(8 * p.ldb * 4), qstepBt = (uint32_t)(8 * ld_t * 4);
    const uint32_t voffA = (uint32_t)((g * 32 + lm) * 16);                        // a_frag: [j][g][row][4]; j advances by 1 KB (soffset)
    const uint32_t voffC = p.c_row_major ? (uint32_t)((lm * p.ldc + 32 * wave + 4 * g) * 4) : (uint32_t)((lm + (32 * wave + 4 * g) * p.ldc) * 4);
    const int64_t n0off = (int64_t)n0 * p.ldb;
    char* const ldsw = reinterpret_cast<char*>(lds) + wave * (2 * WSTAGE * 4);   // this wave's two stages
    const uint32_t lwC = (uint32_t)(bc * LDBW * 4);                               // write: column bc + 8 q -> + 8 q LDBW floats, + 4 x the position of the k
    const uint32_t lrB = (uint32_t)((lm * LDBW + 16 * g) * 4);                    // read: column lm, k = 16 g + 4 j .. + 3

    u32x4 bs0[4], bs1[4];                                // B staging sets (steps of even / odd index)
    u32x4 as0[4], as1[4], as2[4], as3[4];                // A fragment sets (step index mod 4)
    uint32_t kt0 = 0, kt1 = 0, kt2 = 0, kt3 = 0;         // ... and the LDS positions of this lane's four k of that step (k-compaction, vbs_plan.cpp): one byte each

    const float* a_cur = p.A + (int64_t)s_begin * kAFragSlice;  // slice of the next step to be requested (steps are requested in order): [32-byte table, padding][fragments]
    uint32_t vo_cur = voffB;
    int32_t tail_prev = 0;
    // G(step) in two halves (B panel; A fragments + position table) so that a step can place them between its MFMA pairs
    auto issue_loads_b = [&](auto pos_tag, u32x4 (&rb)[4]) __attribute__((always_inline)) -> int32_t {
        constexpr int s = decltype(pos_tag)::value;      // position of the step's record in the window
        const int32_t flags = field(s, F_FLAGS);
        const int32_t tail = TAIL && (flags & STEP_TAIL) != 0;
        if (tail != tail_prev) {
            vo_cur = tail ? voffBt : voffB;
            asm volatile("" : "+v"(vo_cur));
            tail_prev = tail;
        }
        const int64_t gk0 = field(s, F_BROW);
        const float* bptr = tail ? p.B_tail + gk0 : p.B + gk0 + n0off;
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bptr), 0, 0x7ffffff0, 0x00020000);
        const uint32_t qs = tail ? qstepBt : qstepB;
#ifndef SPARTA_DIRECT_PROBE
#define SPARTA_DIRECT_PROBE 0     /* developer probes, TIMING ONLY (results wrong): 1 no B loads, 2 no A loads, 4 no epilogue, 8 no LDS round trip, 64 every tile stores to the first rows of C */
#endif
#ifndef SPARTA_DIRECT_BLOAD_AUX
#define SPARTA_DIRECT_BLOAD_AUX 0
#endif
        if (!(SPARTA_DIRECT_PROBE & 1)) {
#pragma unroll
            for (int q = 0; q < 4; q++) rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rB, vo_cur, qs * q, SPARTA_DIRECT_BLOAD_AUX);
        }
        return flags;
    };
    auto issue_loads_a = [&](int32_t flags, u32x4 (&ra)[4], uint32_t& kt) __attribute__((always_inline)) {
        // the fragments this step's MFMAs will read: 1 KB per group of four MFMAs; the groups behind (columns of zeros, compacted away) are
        // not fetched -- loads past the end of the descriptor return zeros without touching memory
        const int32_t n_quads = (((flags >> STEP_KPAIRS_SHIFT) & 7) >> 1) + 1;
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_cur + 16), 0, n_quads * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_cur), 0, 32, 0x00020000);
        kt = __builtin_amdgcn_raw_buffer_load_b32(rT, (uint32_t)((lane & 7) * 4), 0, 0);
        if (!(SPARTA_DIRECT_PROBE & 2)) {
#pragma unroll
#ifndef SPARTA_DIRECT_ALOAD_AUX
#define SPARTA_DIRECT_ALOAD_AUX 0
#endif
            for (int j = 0; j < 4; j++) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rA, voffA, 1024 * j, SPARTA_DIRECT_ALOAD_AUX);
        }
        a_cur += kAFragSlice;
    };
    auto issue_loads = [&](auto pos_tag, u32x4 (&rb)[4], u32x4 (&ra)[4], uint32_t& kt) __attribute__((always_inline)) -> int32_t {
        const int32_t flags = issue_loads_b(pos_tag, rb);
        issue_loads_a(flags, ra, kt);
        return flags;
    };
    int32_t fq0 = 0, fq1 = 0, fq2 = 0, fq_new = 0;

    // the lane's four k of a column (k = bk + e) go to the step's compacted positions: Bs[column][pos[k]]
    auto write_b = [&](auto stage_tag, const u32x4 (&rb)[4], uint32_t kt) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_tag)::value;
        char* wp[4];
#pragma unroll
        for (int e = 0; e < 4; e++) wp[e] = ldsw + lwC + ((kt >> (8 * e)) & 0xffu) * 4u;
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) *reinterpret_cast<uint32_t*>(wp[e] + (ST * WSTAGE + 8 * q * LDBW) * 4) = rb[q][e];
    };
    // the same, one group of 8 columns at a time (wp[e] = the four write addresses of the step, computed once)
    auto write_b_q = [&](auto stage_tag, auto q_tag, const u32x4 (&rb)[4], char* const (&wp)[4]) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_tag)::value, q = decltype(q_tag)::value;
#pragma unroll
        for (int e = 0; e < 4; e++) *reinterpret_cast<uint32_t*>(wp[e] + (ST * WSTAGE + 8 * q * LDBW) * 4) = rb[q][e];
    };

#ifndef SPARTA_DIRECT_TWOACC
#define SPARTA_DIRECT_TWOACC 0     /* 1: the MFMA pairs alternate between two accumulators (two chains in flight per wave), added in the epilogue */
#endif
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[r] = 0.0f; acc2[r] = 0.0f; }

    CRing cr;                                            // CSTAGE: finished tiles wait here for whole aligned blocks of 32 rows (vbs_kernel_common.hpp)
    cr.ring = lds + 4 * 2 * WSTAGE + wave * kCRingFloats;

    // one step: fragments of B from LDS stage PAR, the next step's panel into the other stage, 16 MFMAs, then the staging set that was
    // just written out and the A set of step i - 1 are refilled with step i + 3.  (Measured and dropped: reading the fragments of step
    // i + 1 during step i -- 16 more registers, 56.0 us against 54.1.)
#ifndef SPARTA_DIRECT_FBPRE
#define SPARTA_DIRECT_FBPRE 0      /* 1: the B fragments of step i + 1 are read from LDS behind pair 6 of step i (two fragment sets) instead of at the head of step i + 1 */
#endif
    f32x4 fbE[4], fbO[4];                                // SPARTA_DIRECT_FBPRE: fragments of the even / odd steps
    auto step = [&](auto u_tag, int32_t flags, u32x4 (&wa)[4], u32x4 (&nb)[4], u32x4 (&na)[4], uint32_t ktw, uint32_t& ktn) __attribute__((always_inline)) {
        constexpr int i = decltype(u_tag)::value;        // step index mod 4 = position of its record in the window
        constexpr int PAR = i & 1;
        f32x4 fbl[4];
        f32x4 (&fb)[4] = SPARTA_DIRECT_FBPRE ? (PAR ? fbO : fbE) : fbl;
        f32x4 (&fbn)[4] = PAR ? fbE : fbO;
        if (SPARTA_DIRECT_PROBE & 8) {
#pragma unroll
            for (int j = 0; j < 4; j++) fb[j] = __builtin_bit_cast(f32x4, nb[j]);
        } else if (!SPARTA_DIRECT_FBPRE) {
#pragma unroll
            for (int j = 0; j < 4; j++) fb[j] = *reinterpret_cast<const f32x4*>(ldsw + lrB + (PAR * WSTAGE + 4 * j) * 4);
        }
        // The step's other work sits BETWEEN its MFMA pairs: a wave issues in order, and an MFMA of this chain waits 64 cycles for the one before it -- whatever
        // independent instruction comes next in program order issues inside that wait.  Pairs 0..3 are followed by the LDS writes of one group of eight columns of
        // the NEXT step's panel (W(i + 1); the wait for that panel's loads now hides behind the first pair), pair 4 by the B loads of step i + 3, pair 5 by its A
        // loads.  The pairs a step does not need are skipped (scalar branches); the work between them always runs.
        char* wp[4];
#pragma unroll
        for (int e = 0; e < 4; e++) wp[e] = ldsw + lwC + ((ktw >> (8 * e)) & 0xffu) * 4u;
        const int32_t n_pairs = ((flags >> STEP_KPAIRS_SHIFT) & 7) + 1;
        int32_t flags_new = 0;
        static_for<0, 8>([&](auto t2_tag) __attribute__((always_inline)) {
            constexpr int t2 = decltype(t2_tag)::value;
            if (t2 < n_pairs) {
                if constexpr (SPARTA_DIRECT_TWOACC) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[t2 >> 1][2 * (t2 & 1)], __uint_as_float(wa[t2 >> 1][2 * (t2 & 1)]), acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[t2 >> 1][2 * (t2 & 1) + 1], __uint_as_float(wa[t2 >> 1][2 * (t2 & 1) + 1]), acc2, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[t2 >> 1][2 * (t2 & 1)], __uint_as_float(wa[t2 >> 1][2 * (t2 & 1)]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[t2 >> 1][2 * (t2 & 1) + 1], __uint_as_float(wa[t2 >> 1][2 * (t2 & 1) + 1]), acc, 0, 0, 0);
                }
            }
#ifndef SPARTA_DIRECT_ORDER
#define SPARTA_DIRECT_ORDER 0
#endif
            if constexpr (SPARTA_DIRECT_ORDER == 0) {
                if constexpr (t2 < 4) {
                    if (!(SPARTA_DIRECT_PROBE & 8)) write_b_q(std::integral_constant<int, 1 - PAR>{}, t2_tag, nb, wp);
                } else if constexpr (t2 == 4) {
                    flags_new = issue_loads_b(std::integral_constant<int, i + 3>{}, nb);          // G(i + 3), B half: refills the staging set just written out
                } else if constexpr (t2 == 5) {
                    issue_loads_a(flags_new, na, ktn);                                            // G(i + 3), A half
                }
                if constexpr (SPARTA_DIRECT_FBPRE && t2 == 6) {                                   // W(i + 1) is complete (program order, one wave): its fragments for the next step
#pragma unroll
                    for (int j = 0; j < 4; j++) fbn[j] = *reinterpret_cast<const f32x4*>(ldsw + lrB + ((1 - PAR) * WSTAGE + 4 * j) * 4);
                }
            } else {                                     // A loads first (they depend on nothing), then the writes, then the B loads
                if constexpr (t2 == 0) {
                    flags_new = field(i + 3, F_FLAGS);
                    issue_loads_a(flags_new, na, ktn);
                } else if constexpr (t2 < 5) {
                    if (!(SPARTA_DIRECT_PROBE & 8)) write_b_q(std::integral_constant<int, 1 - PAR>{}, std::integral_constant<int, t2 - 1>{}, nb, wp);
                } else if constexpr (t2 == 5) {
                    flags_new = issue_loads_b(std::integral_constant<int, i + 3>{}, nb);
                }
            }
        });
        fq_new = flags_new;
        if ((flags & STEP_LAST) && !(SPARTA_DIRECT_PROBE & 4)) {
            // epilogue (as in vbs_spmm_f32_stream_kernel): stored from copies, accumulators cleared here
            if constexpr (SPARTA_DIRECT_TWOACC) {
#pragma unroll
                for (int q = 0; q < 16; q++) { acc[q] += acc2[q]; acc2[q] = 0.0f; }
            }
            if (flags & STEP_SPLIT) {
                const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(ws + (int64_t)field(i, F_SLOT) * SK_SLOT_FLOATS, 0, SK_SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[q]), rW, (uint32_t)tid * 4u, (uint32_t)(q * kThreads * 4), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(0u, rW, (uint32_t)tid * 4u, (uint32_t)((16 + q) * kThreads * 4), 0);   // rows 32..63 of the image: none
                }
            } else if (CSTAGE) {
                cr.park(p, n0, lm, g, voffC, acc, field(i, F_CROW), flags & 0xffff);
            } else {
                const int mt = flags & 0xffff;
                const int64_t c_row = (SPARTA_DIRECT_PROBE & 64) ? 0 : field(i, F_CROW);      // probe 64: every tile stores to the first rows of C
                float* cbase = p.c_row_major ? p.C + c_row * p.ldc + n0 : p.C + c_row + (int64_t)n0 * p.ldc;
                const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, 0x7ffffff0, 0x00020000);
                const uint32_t jstep = p.c_row_major ? 4u : (uint32_t)p.ldc * 4u;          // bytes per output column
                if (lm < mt) {
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) v[q] = acc[q];
                    if (p.accumulate) {
                        uint32_t old[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) old[q] = __builtin_amdgcn_raw_buffer_load_b32(rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep, 0);
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] += __uint_as_float(old[q]);
                    }
                    if (p.c_nt) {
#pragma unroll
                        for (int q = 0; q < 16; q++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep, 2);
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; q++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.0f;
        }
    };

    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using c2 = std::integral_constant<int, 2>;
    using c3 = std::integral_constant<int, 3>;
    // the window of steps [i, i + 4): requested one round earlier (32 loads are issued in between: vmcnt(16) is a free wait).  Both sides
    // are inline asm on purpose, see vbs_spmm_f32_stream_kernel.
    auto window_swap = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(16)" : "+v"(vnext) : : "memory");
        vwin = vnext;
    };
    auto window_request = [&](int i) __attribute__((always_inline)) {
        const int32_t* nb = srec + (int64_t)(i + 4) * 8 + lane;
        asm volatile("global_load_dword %0, %1, off" : "=&v"(vnext) : "v"(nb) : "memory");
    };
    fq0 = issue_loads(c0{}, bs0, as0, kt0);
    fq1 = issue_loads(c1{}, bs1, as1, kt1);
    write_b(c0{}, bs0, kt0);                             // W(0)
    if (SPARTA_DIRECT_FBPRE) {
#pragma unroll
        for (int j = 0; j < 4; j++) fbE[j] = *reinterpret_cast<const f32x4*>(ldsw + lrB + (4 * j) * 4);
    }
    fq2 = issue_loads(c2{}, bs0, as2, kt2);
    // step i: LDS stage i & 1, A set i & 3; writes out staging set (i + 1) & 1 and refills it, and A set (i + 3) & 3, with step i + 3
    const int n4 = n & ~3;
    for (int i = 0; i < n4; i += 4) {
        if (i > 0) window_swap();
        window_request(i);
        step(c0{}, fq0, as0, bs1, as3, kt1, kt3);
        fq0 = fq1; fq1 = fq2; fq2 = fq_new;
        step(c1{}, fq0, as1, bs0, as0, kt2, kt0);
        fq0 = fq1; fq1 = fq2; fq2 = fq_new;
        step(c2{}, fq0, as2, bs1, as1, kt3, kt1);
        fq0 = fq1; fq1 = fq2; fq2 = fq_new;
        step(c3{}, fq0, as3, bs0, as2, kt0, kt2);
        fq0 = fq1; fq1 = fq2; fq2 = fq_new;
    }
    if (n > n4) {
        if (n4 > 0) window_swap();
        step(c0{}, fq0, as0, bs1, as3, kt1, kt3);
        fq0 = fq1; fq1 = fq2; fq2 = fq_new;
        if (n - n4 >= 2) {
            step(c1{}, fq0, as1, bs0, as0, kt2, kt0);
            fq0 = fq1; fq1 = fq2; fq2 = fq_new;
        }
        if (n - n4 == 3) step(c2{}, fq0, as2, bs1, as1, kt3, kt1);
    }
    if (CSTAGE) cr.flush(p, n0, lm, g, voffC, true);
    clock_probe(p.clk, 2);
#undef field
}

}  // namespace

namespace {

// The reference-layout image of A (column-major h x w blocks) rebuilt from the fragment image of the one-tile plan: slice q of a_frag = step q, element (row m, column k
// of the 32-deep slice) at fragment position pos[k] (k-compaction: vbs_plan.cpp), i.e. frag[((j * 2 + g) * 32 + m) * 4 + e] with pos[k] = 16 g + 4 j + e; all-zero
// columns hold zeros there too.  Used when a handle that dropped its reference-layout image (vbs_capi.cpp: one image of A, not two, once the no-barrier kernel has won the
// plan-time autotune) is asked for something that reads it after all: SPARTA_SPMM_EXACT, a row-major or gathered B, the per-class kernels.  The copy is exact.
__global__ __launch_bounds__(kThreads) void vbs_f32_legacy_from_frag_kernel(const StepRec* steps, int64_t n_steps, const float* a_frag, float* A) {
    for (int64_t q = blockIdx.x; q < n_steps; q += gridDim.x) {
        const int64_t a_off = steps[q].a_off, h = steps[q].h;
        const int mt = steps[q].mt_flags & 0xffff;
        const float* sl = a_frag + q * kAFragSlice;
        const unsigned char* pos = reinterpret_cast<const unsigned char*>(sl);
        const float* frag = sl + 16;
        for (int x = threadIdx.x; x < 32 * 32; x += kThreads) {
            const int k = x >> 5, m = x & 31;
            if (m >= mt) continue;
            const int kp = pos[k], g = kp >> 4, j = (kp & 15) >> 2, e = kp & 3;
            A[a_off + (int64_t)k * h + m] = frag[((j * 2 + g) * 32 + m) * 4 + e];
        }
    }
}

}  // namespace

namespace sparta_dev {

void launch_f32_legacy_from_frag(hipStream_t st, const StepRec* steps, int64_t n_steps, const float* a_frag, float* A) {
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_steps, 65536));
    hipLaunchKernelGGL(vbs_f32_legacy_from_frag_kernel, dim3(grid), dim3(kThreads), 0, st, steps, n_steps, a_frag, A);
}

void launch_f32_direct(bool c_stage, dim3 grid, hipStream_t st, const StreamParams& sp) {
    static const bool lean_off = [] { const char* e = std::getenv("SPARTA_F32_TAILFREE"); return e && atoi(e) == 0; }();     // (A/B runs)
    if (sp.B_tail != nullptr || lean_off) {
        if (c_stage) hipLaunchKernelGGL((vbs_spmm_f32_direct_kernel<true, true>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_f32_direct_kernel<false, true>), grid, dim3(kThreads), 0, st, sp);
    } else {
        if (c_stage) hipLaunchKernelGGL((vbs_spmm_f32_direct_kernel<true, false>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_f32_direct_kernel<false, false>), grid, dim3(kThreads), 0, st, sp);
    }
}

}  // namespace sparta_dev
