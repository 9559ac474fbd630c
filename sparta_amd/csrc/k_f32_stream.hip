// k_f32_stream.hip -- persistent fp32 "stream" kernels (the product path for w % 32 == 0, N % 128 == 0), the fix-up of split
// tiles, the zero-padded B-tail copy and the streamed zero fill.  Part of the device side of libsparta_amd.so; see
// vbs_device.hpp for the translation-unit map and DESIGN.md section 3.2.
#include "vbs_kernel_common.hpp"

using namespace sparta_dev;

namespace {

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
    if (p.stagger > 0 && blockIdx.x >= (gridDim.x >> 1))          // developer knob: de-phase the two workgroups that share a CU
        for (int k = 0; k < p.stagger; k++) __builtin_amdgcn_s_sleep(1);
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
#ifdef SPARTA_EXTRA_VALU
        {   // developer probe: what one more VALU instruction per step costs (the loop's VALU work competes with the co-resident wave's MFMAs)
            int dummy = 0;
#pragma unroll
            for (int e = 0; e < SPARTA_EXTRA_VALU; e++) asm volatile("v_mov_b32 %0, %1" : "+v"(dummy) : "v"(vo_cur));
        }
#endif
#ifdef SPARTA_EXTRA_SALU
        {
            int dummy = 0;
#pragma unroll
            for (int e = 0; e < SPARTA_EXTRA_SALU; e++) asm volatile("s_add_i32 %0, %0, 1" : "+s"(dummy));
        }
#endif
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
#undef field

// A tile that dominates the plan (a hub block-row) is split over hundreds of workers; adding its partial images one after the
// other in ONE workgroup is a latency-bound chain (measured: 512 images, 350 us).  First stage for such plans: blockIdx.z = group
// of kFixGroup consecutive images, summed in order into the group's first image; the fix-up kernel then adds the group leaders
// (stride kFixGroup).  Fixed grouping -> the result is reproducible run to run.
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


}  // namespace

namespace sparta_dev {

void launch_f32_stream(bool mi2, bool b_row_major, bool gathered, dim3 grid, hipStream_t st, const StreamParams& sp) {
    if (mi2) {
        if (gathered) hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<false, true, true>), grid, dim3(kThreads), 0, st, sp);
        else if (b_row_major) hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<true, false, true>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<false, false, true>), grid, dim3(kThreads), 0, st, sp);
    } else {
        if (gathered) hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<false, true, false>), grid, dim3(kThreads), 0, st, sp);
        else if (b_row_major) hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<true, false, false>), grid, dim3(kThreads), 0, st, sp);
        else hipLaunchKernelGGL((vbs_spmm_f32_stream_kernel<false, false, false>), grid, dim3(kThreads), 0, st, sp);
    }
}

void launch_fixup_group(dim3 grid, hipStream_t st, const FixRec* fix, const int32_t* big, const int32_t* fix_slots, float* ws_all, int64_t ws_slab_stride) {
    hipLaunchKernelGGL(vbs_spmm_f32_fixup_group_kernel, grid, dim3(kThreads), 0, st, fix, big, fix_slots, ws_all, ws_slab_stride);
}

void launch_fixup(dim3 grid, hipStream_t st, const FixRec* fix, const int32_t* fix_slots, const float* ws_all, int64_t ws_slab_stride, float* C, int64_t ldc,
                  int c_row_major, int accumulate) {
    hipLaunchKernelGGL(vbs_spmm_f32_fixup_kernel, grid, dim3(kThreads), 0, st, fix, fix_slots, ws_all, ws_slab_stride, C, ldc, c_row_major, accumulate);
}

void launch_tail_copy(hipStream_t st, const float* B, int64_t ldb, int b_row_major, int64_t row0, int64_t cols, int w, int N, float* B_tail) {
    hipLaunchKernelGGL(vbs_tail_copy_kernel, dim3(32), dim3(kThreads), 0, st, B, ldb, b_row_major, row0, cols, w, N, B_tail);
}

void launch_zero_rows(dim3 grid, hipStream_t st, float* C, int64_t ldc, int c_row_major, int64_t row0, int64_t nrows, int N) {
    hipLaunchKernelGGL(vbs_zero_rows_kernel, grid, dim3(kThreads), 0, st, C, ldc, c_row_major, row0, nrows, N);
}

}  // namespace sparta_dev
