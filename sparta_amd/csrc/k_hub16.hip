// k_hub16.hip -- the GEMM-shaped kernel for the dense part of a 16-bit handle (the hub of a power-law matrix under the fixed 64 x 64 grid):
// vbs_spmm_h16_hub_kernel.  Part of the device side of libsparta_amd.so; see vbs_device.hpp (HubStep / HubTile / HubParams) and DESIGN.md section 12.
//
// Why another kernel.  The no-barrier kernels of k_h16.hip give every wave its own panel of B and its own copy of the slice of A: 64 KB through the
// texture path per 64 MFMAs.  On long dense tiles -- 10^3..10^4 steps per tile, the 64-row block-rows of a hub -- that load path, not the matrix pipe,
// sets the pace: 0.13 of the bf16 MFMA peak (round 3).  Here a workgroup owns a GROUP TILE -- kHubG = 2 block-rows of 64 rows walked over the union of
// their block columns -- x one 256-column slab of C, and a step's 128 x KP slice of A and KP x 256 panel of B are loaded ONCE per workgroup, straight
// into LDS (buffer_load ... lds: no staging registers), for all four waves: 48 KB per 128 MFMAs.
//   * wave (wr, wc) owns sub-tile wr (64 rows) x columns [128 wc, 128 wc + 128): 2 x 4 accumulators of 32 x 32, D = Bpanel^T . Atile^T as in the
//     other kernels (an accumulator register holds 32 consecutive rows of one column of C: whole 128-byte runs of the reference's column-major C);
//   * LDS holds NS stages of [A: 2 sub-tiles x 64 rows x KP][B: 256 columns x KP], rows of KP 16-bit elements (64 or 128 bytes), 16-byte chunk c of
//     row r stored at position c ^ swz(r), swz(r) = (r / rows per 256 bytes) mod chunks per row: a ds_read_b128 of 16 lanes with 16 different rows
//     then covers all 64 banks.  The LDS-direct load writes lane l's 16 bytes at base + 16 l, so the swizzle sits on the SOURCE side: the slices of A
//     are stored pre-swizzled (vbs_plan.cpp), a lane of a B load fetches the chunk that belongs at its position;
//   * one barrier per step: wait (counted vmcnt) for this wave's loads of step i, barrier, issue the loads of step i + NS - 1 into the stage that
//     step i - 1 has just left, multiply step i.  The loads of NS - 2 steps stay in flight across every barrier;
//   * a sub-tile that has no block in a step's block column (flags bits 0..1) is not fetched: a descriptor of zero records, the LDS-direct load then writes
//     zeros without touching memory (checked on the hardware: scripts/ubench/glds_probe.hip) and its two waves multiply zeros;
//   * tile ends: the 2 x 4 accumulators go to C (same lane -> element map and store forms as k_h16.hip) or, for a tile shared with other workers, into
//     the workspace images of its two sub-tiles and two 128-column slabs, in the layout vbs_spmm_f32_fixup_kernel reads.
// Step records come through SCALAR loads (one s_load_dwordx8 per step, requested one step before the step's loads are issued): a step here is 32 MFMAs per wave
// behind a barrier, its fragment reads are waited for with lgkmcnt(0) anyway.  (First version: 8 records per VGPR fetched by an inline-assembly vector load,
// as in k_h16.hip -- the register allocator put a copy of the destination right behind the load, i.e. in front of the data: memory faults.  Do not hide a load
// whose destination is loop-carried from the compiler.)
#include "vbs_kernel_common.hpp"

using namespace sparta_dev;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef SPARTA_HUB_A_AUX
#define SPARTA_HUB_A_AUX 0          /* cache policy of the loads of A (developer A/B): 0 default, 2 non-temporal */
#endif
#ifndef SPARTA_HUB_PROBE
#define SPARTA_HUB_PROBE 0        /* developer probes, TIMING ONLY (results wrong): 1 no B loads, 2 no A loads, 4 no MFMAs, 8 no fragment reads, 16 no epilogue */
#endif

// LW (loader waves): the workgroup has EIGHT waves -- waves 4..7 issue the step's LDS-direct loads and nothing else, waves 0..3 multiply and never touch vector memory
// outside a tile's epilogue.  An LDS-direct load holds its wave's instruction stream for 60-100 cycles: in a wave that also owns the SIMD's MFMAs that is matrix-pipe
// time (twelve loads between 32 MFMAs: 1.04 PFLOP/s with every operand cache-hot, 1.8 without loads); a loader wave beside it on the same SIMD costs an issue slot.
template <int KP, int NS, int G, bool BF16, bool GATHERED, int WPC, bool LW>
__global__ __launch_bounds__(64 * 2 * G * (LW ? 2 : 1), (G / 2) * (LW ? 2 : 1) * WPC) void vbs_spmm_h16_hub_kernel(const HubParams p) {
    // G = 2: four waves, 128 rows x 256 columns per workgroup; G = 4: EIGHT waves (two per SIMD), 256 rows x 256 columns: the panel of B feeds four sub-tiles --
    // 64 KB through the load path per 256 MFMAs instead of 48 KB per 128.  The load path (LDS-direct or through registers alike: ~50-70 GB/s per CU from the L2s,
    // less from beyond) is what bounds this kernel, so bytes per flop is the lever.
    static_assert(G == 2 || G == 4, "group tiles of two or four 64-row sub-tiles");
    static_assert(!(LW && G == 4), "loader waves: with the four-wave form only");
    constexpr int NWV = 2 * G;                           // compute waves: wave (wr, wc) = (sub-tile, half slab)
    constexpr int RB = KP * 2;                           // bytes per LDS row (one row of a slice / one column of a panel)
    constexpr int CPR = RB / 16;                         // 16-byte chunks per row: 8 / 4
    constexpr int R256 = 256 / RB;                       // rows per 256 bytes of LDS: 2 / 4
    constexpr int SLICE = 64 * RB;                       // bytes of one sub-tile's slice
    constexpr int A_BYTES = G * SLICE, B_BYTES = 256 * RB, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = SLICE / 2048;                     // LDS-direct loads per wave and step: its half of its own sub-tile's slice (4 / 2 pieces of 1 KB)
    constexpr int NB = B_BYTES / 1024 / NWV;             // ... and its share of the panel of B (G = 2: 8 / 4, G = 4: 4 / 2)
    constexpr int CPI = 1024 / RB;                       // columns of B per 1 KB piece: 8 / 16
    constexpr int WCOLS = NB * CPI;                      // columns of the panel a wave loads
    constexpr int NKG = KP / 16;                         // k groups (MFMAs per accumulator) per step
    constexpr int AHEAD = NS - 1;                        // steps between a step's loads and its MFMAs
    constexpr int LPS = NA + NB;
    static_assert((AHEAD - 1) * LPS <= 63, "vmcnt holds 6 bits");
    static_assert((WCOLS / R256) % CPR == 0, "the swizzle of a column of B must not depend on the wave");
    static_assert(NS * STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char lds[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id % NWV;                      // place among the compute (or the loader) waves
    const int wr = wave >> 1, wc = wave & 1;
    const int lm = lane & 31, g = lane >> 5;
    // workgroup id -> (worker, slab).  Workgroups are dealt round-robin over the 8 XCDs (id & 7) and start in id order, one per CU.  (1) the slabs of a worker are
    // neighbours on ONE XCD (ids 8 apart): they start together and the second one finds the slices of A in that XCD's L2; (2) the workers that run at the SAME TIME are
    // a contiguous block of the plan's worker order -- the plan puts the workers that walk the same rows of B next to each other (vbs_plan.cpp, chunk-major units) --
    // and inside the block an XCD holds consecutive ones: round r of the dispatch = workers [r * 8 rq, (r + 1) * 8 rq), XCD x the rq = (CUs per XCD) / slabs from x * rq on
    const int xcd = (int)blockIdx.x & 7, jj = (int)blockIdx.x >> 3;
    const int slab = jj % p.n_slabs, qx = jj / p.n_slabs;                    // qx: this worker's place among its XCD's workers, in dispatch order
    const int per_x = p.n_workers >> 3;
    const int rq = per_x / p.n_slabs > 0 ? per_x / p.n_slabs : 1;
    const int round = qx / rq;
    const int rq_here = per_x - round * rq < rq ? per_x - round * rq : rq;   // (the last round may be short)
    const int worker = round * 8 * rq + xcd * rq_here + (qx - round * rq);
    const int s_begin = p.worker_range[2 * worker];
    const int n = p.worker_range[2 * worker + 1] - s_begin;
    if (n <= 0) return;
    const int n0 = slab * 256;
    // n_cols is a multiple of 128: in the last slab of a call with n_cols % 256 == 128 the upper 128 columns do not exist -- the waves that would load them get
    // descriptors of zero records (zeros in LDS, no access), the waves that would own them (wc = 1) store nothing
    const bool bcols_ok = n0 + WCOLS * wave < p.n_cols, ccols_ok = n0 + 128 * wc < p.n_cols;

    // step records, read through the constant address space (scalar loads): record j as eight SGPRs
    typedef const __attribute__((address_space(4))) int32_t* crec_t;
    const crec_t srec = (crec_t)(reinterpret_cast<const int32_t*>(p.steps + s_begin));
    struct Rec { int32_t f[8]; };
    enum { F_A_LO = 0, F_A_HI = 1, F_BROW = 2, F_SHARD = 3, F_FLAGS = 4, F_SLOT = 5, F_TILE = 6 };
    auto load_rec = [&](int j) __attribute__((always_inline)) -> Rec {
        Rec r;
#pragma unroll
        for (int f = 0; f < 8; f++) r.f[f] = srec[(int64_t)j * 8 + f];
        return r;
    };

    // ---- per-lane constants ----
    const int colp = lane / CPR, pos = lane % CPR;       // B loads: column colp of the piece's CPI, LDS position pos of its row
    const int swz0 = (colp / R256) & (CPR - 1);          // swizzle of that column in pieces q = 0, 2, ..; odd pieces (KP = 64 only): + 4
    const int swz1 = ((CPI + colp) / R256) & (CPR - 1);
    const uint32_t voffB0 = (uint32_t)(colp * (int64_t)p.ldb * 2) + (uint32_t)((pos ^ swz0) * 16);
    const uint32_t voffB1 = (uint32_t)(colp * (int64_t)p.ldb * 2) + (uint32_t)((pos ^ swz1) * 16);
    const uint32_t voffT0 = (uint32_t)(colp * p.w * 2) + (uint32_t)((pos ^ swz0) * 16);
    const uint32_t voffT1 = (uint32_t)(colp * p.w * 2) + (uint32_t)((pos ^ swz1) * 16);
    const uint32_t qstepB = (uint32_t)(CPI * p.ldb * 2), qstepT = (uint32_t)(CPI * p.w * 2);
    const int64_t wcolB = (int64_t)(n0 + WCOLS * wave) * p.ldb, wcolT = (int64_t)(n0 + WCOLS * wave) * p.w;      // the columns of the panel this wave loads
    const uint32_t voffA = (uint32_t)lane * 16u;
    // fragment reads: row / column lm of a 32-row group, chunk 2 kg + g at position (2 kg + g) ^ swz(lm) = (g ^ swz(lm)) ^ 2 kg
    const uint32_t frag0 = (uint32_t)(lm * RB + ((g ^ ((lm / R256) & (CPR - 1))) * 16));
    const uint32_t voffC = p.c_row_major ? (uint32_t)((lm * p.ldc + 4 * g) * 4) : (uint32_t)((lm + (4 * g) * p.ldc) * 4);
    char* const lds0 = lds;

    // the loads of one step: descriptors and offsets first (scalar work), then LPS LDS-direct loads issued ONE AT A TIME -- the step body places them between
    // its MFMAs: an LDS-direct load costs the wave 60-100 cycles of issue, and with one wave per SIMD nothing else feeds the matrix pipe meanwhile (all twelve in
    // front of the MFMAs: 0.78 PFLOP/s on a dense hub; no loads at all: 1.52)
    // the context of the loads being issued: plain variables of the kernel body (as members of a struct handed to the lambdas, the HOST pass of hipcc 7.2 dropped
    // the kernel's stubs without a diagnostic -- "undefined symbol __device_stub__..." at link time -- whenever the LDS-direct builtin took a descriptor from it)
    __amdgpu_buffer_rsrc_t c_rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.B), 0, 0, 0x00020000), c_rA = c_rB;
    uint32_t c_vb0 = 0, c_vb1 = 0, c_qs = 0;
    char* c_st = lds0;
    auto prepare = [&](const Rec& rec, int stage) __attribute__((always_inline)) {
        const int32_t flags = rec.f[F_FLAGS];
        const int64_t a_off = (int64_t)(uint32_t)rec.f[F_A_LO] | ((int64_t)rec.f[F_A_HI] << 32);
        const uint16_t* ap = p.A + a_off;
        const int32_t tail = flags & STEP_TAIL;
        const int64_t brow = rec.f[F_BROW];
        const uint16_t* bptr = tail ? p.B_tail + brow + wcolT : p.B + (GATHERED ? (int64_t)rec.f[F_SHARD] * p.shard_stride : (int64_t)0) + brow + wcolB;
        c_rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(bptr), 0, bcols_ok ? 0x7ffffff0 : 0, 0x00020000);
        // this wave loads half of the slice of its OWN sub-tile (wr).  The slices of the sub-tiles that have a block in this column lie back to back; a sub-tile
        // without one gets a descriptor of zero records: the load writes zeros to LDS without touching memory
        const int32_t before = __builtin_popcount(flags & ((1 << wr) - 1));
        c_rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ap + before * (SLICE / 2)), 0, ((flags >> wr) & 1) ? SLICE : 0, 0x00020000);
        c_vb0 = tail ? voffT0 : voffB0; c_vb1 = tail ? voffT1 : voffB1; c_qs = tail ? qstepT : qstepB;
        c_st = lds0 + stage * STAGE;
    };
    // (t is a compile-time constant at every call site: a plain int parameter, folded after inlining)
    auto issue_one = [&](int t) __attribute__((always_inline)) {
        // t = 0 .. NB - 1: pieces of the panel of B; NB .. LPS - 1: pieces of the slice of A
        if (t < NB) {
            if (!(SPARTA_HUB_PROBE & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rB, (lds_ptr_t)(c_st + A_BYTES + (wave * NB + t) * 1024), 16, (t & 1) ? c_vb1 : c_vb0, c_qs * (uint32_t)t, 0, 0);
        } else {
            const int q = wc * NA + (t - NB);            // piece of the slice
            if (!(SPARTA_HUB_PROBE & 2))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rA, (lds_ptr_t)(c_st + wr * SLICE + q * 1024), 16, voffA, (uint32_t)(q * 1024), 0, SPARTA_HUB_A_AUX);
        }
    };
    auto issue = [&](const Rec& rec, int stage) __attribute__((always_inline)) -> int32_t {       // all of a step's loads back to back (prologue)
        prepare(rec, stage);
#pragma unroll
        for (int t = 0; t < LPS; t++) issue_one(t);
        return rec.f[F_FLAGS];
    };

    if constexpr (LW) {
        if (wave_id >= NWV) {                            // ---- a loader wave: wait for its loads of step i, barrier, issue step i + AHEAD ----
#pragma unroll
            for (int k = 0; k < AHEAD; k++) issue(load_rec(k), k);
            Rec lnx = load_rec(AHEAD);
            int lstage = 0;
            for (int i = 0; i < n; i++) {
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"((AHEAD - 1) * LPS) : "memory");
                __builtin_amdgcn_s_barrier();
                const Rec rec = lnx;
                lnx = load_rec(i + AHEAD + 1);
                int jstage = lstage + AHEAD; if (jstage >= NS) jstage -= NS;
                issue(rec, jstage);
                lstage = lstage + 1 == NS ? 0 : lstage + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[rt][ct][r] = 0.0f;
    // The accumulators are pinned to the AGPRs (inline-assembly MFMA, "+a"): left to itself the register allocator keeps them in VGPRs between steps and copies
    // each one into AGPRs in front of every MFMA (16 v_accvgpr_write per MFMA -- seen in the ISA of this kernel as soon as the step's order was pinned with
    // scheduling barriers; the same trap as the four-accumulator kernel of k_h16.hip, round 3).  The hazard recogniser does not look inside inline assembly, so the
    // wait states are written out: two in front of each MFMA (a fragment that was parked in an AGPR comes back through v_accvgpr_read right in front of it), 24 between
    // the last MFMA and the epilogue's reads, four behind the zeroing.
#define HUB_ACC_ALL "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3])
    asm volatile("s_nop 4" : HUB_ACC_ALL);
    auto mfma = [&](const u32x4& bf, const u32x4& af, auto& a) __attribute__((always_inline)) {        // (generic: the host pass never instantiates the body -- it would reject the AMDGPU constraints silently and drop the kernel stubs)
        if constexpr (BF16) { asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a) : "v"(bf), "v"(af)); }
        else { asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(a) : "v"(bf), "v"(af)); }
    };

    // flags / slot / tile of the steps in flight (the record window may have moved on by the time a step is multiplied)
    int32_t fq[NS], sq[NS], tq[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) { fq[k] = 0; sq[k] = -1; tq[k] = 0; }
#pragma unroll
    for (int k = 0; k < AHEAD; k++) {                    // prologue: steps 0 .. AHEAD - 1
        const Rec r = load_rec(k);
        sq[k] = r.f[F_SLOT]; tq[k] = r.f[F_TILE]; fq[k] = r.f[F_FLAGS];
        if constexpr (!LW) issue(r, k);
    }
    Rec nxt = load_rec(AHEAD);                           // the record of the step whose loads the next iteration issues

    int stage = 0;                                       // stage of step i
    for (int i = 0; i < n; i++) {
        // (1) this wave's loads of step i have landed: all but the loads of the AHEAD - 1 younger steps
        if constexpr (!LW) asm volatile("s_waitcnt vmcnt(%0)" : : "n"((AHEAD - 1) * LPS) : "memory");
        // (2) everybody's have, and everybody has finished reading the stage of step i - 1
        __builtin_amdgcn_s_barrier();
        // (3) the loads of step i + AHEAD go into the stage that step i - 1 has left, BETWEEN the MFMAs of (4) step i; the record of step i + AHEAD + 1 is
        // requested for the next iteration
        const Rec rec = nxt;
        nxt = load_rec(i + AHEAD + 1);
        int jstage = stage + AHEAD; if (jstage >= NS) jstage -= NS;
        sq[AHEAD] = rec.f[F_SLOT]; tq[AHEAD] = rec.f[F_TILE]; fq[AHEAD] = rec.f[F_FLAGS];
        if constexpr (!LW) prepare(rec, jstage);
        const int32_t flags = fq[0];
        // (a sub-tile without a block in this column multiplies the zeros its loads wrote: no branch around the MFMAs -- a join there makes the register allocator
        // keep a second home for the accumulators and copy all 128 registers into and out of it every step)
        if (!(SPARTA_HUB_PROBE & 4)) {
            const uint32_t sa = (uint32_t)(stage * STAGE + wr * SLICE), sb = (uint32_t)(stage * STAGE + A_BYTES + 128 * wc * RB);
            u32x4 af[NKG][2], bf[NKG][4];
            auto read_frags = [&](auto kg_tag) __attribute__((always_inline)) {
                constexpr int kg = decltype(kg_tag)::value;
                const uint32_t fr = frag0 ^ (uint32_t)(32 * kg);
                if (SPARTA_HUB_PROBE & 8) {
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) af[kg][rt] = u32x4{fr, fr + 1u, fr + 2u, (uint32_t)rt};
#pragma unroll
                    for (int ct = 0; ct < 4; ct++) bf[kg][ct] = u32x4{fr, fr + 3u, fr + 5u, (uint32_t)ct};
                } else {
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) af[kg][rt] = *reinterpret_cast<const u32x4*>(lds0 + (sa + fr) + rt * 32 * RB);
#pragma unroll
                    for (int ct = 0; ct < 4; ct++) bf[kg][ct] = *reinterpret_cast<const u32x4*>(lds0 + (sb + fr) + ct * 32 * RB);
                }
            };
            read_frags(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NM = 8 * NKG;                                    // MFMAs of the step
            constexpr int SPREAD = NM - NM / 8;                            // the loads are all out before the last eighth of the MFMAs
            static_for<0, NM>([&](auto m_tag) __attribute__((always_inline)) {
                constexpr int m = decltype(m_tag)::value;
                constexpr int kg = m / 8, rt = (m % 8) / 4, ct = m % 4;
                mfma(bf[kg][ct], af[kg][rt], acc[rt][ct]);
                __builtin_amdgcn_sched_barrier(0);
                // behind MFMA m: the fragments of the next k group (behind the group's first MFMA), the loads t with t * SPREAD / LPS == m
                if constexpr (m % 8 == 0 && kg + 1 < NKG) read_frags(std::integral_constant<int, kg + 1>{});
                constexpr int t0 = (m * LPS + SPREAD - 1) / SPREAD, t1 = ((m + 1) * LPS + SPREAD - 1) / SPREAD;
                if constexpr (!LW) {
#pragma unroll
                    for (int t = (t0 < LPS ? t0 : LPS); t < (t1 < LPS ? t1 : LPS); t++) issue_one(t);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        } else if constexpr (!LW) {
#pragma unroll
            for (int t = 0; t < LPS; t++) issue_one(t);
        }
        if ((flags & STEP_LAST) && ccols_ok && !(SPARTA_HUB_PROBE & 16)) {
            asm volatile("s_nop 15\n\ts_nop 7" : HUB_ACC_ALL);           // the last MFMA's passes before its accumulator may be read
            const HubTile* tl = p.tiles + tq[0];
            const int32_t c_row = tl->c_row[wr], mt = tl->mt[wr];
            if (flags & STEP_SPLIT) {
                // image of (sub-tile wr, 128-column slab 2 slab + wc): [register 0..31][4 waves x 32 columns][64 lanes], registers 0..15 rows 0..31, 16..31 rows 32..63
                float* img = p.ws + (int64_t)(2 * slab + wc) * p.ws_slab_stride + (int64_t)(sq[0] + wr) * SK_SLOT_FLOATS;
                const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(img, 0, SK_SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
                for (int rt = 0; rt < 2; rt++)
#pragma unroll
                    for (int ct = 0; ct < 4; ct++)
#pragma unroll
                        for (int q = 0; q < 16; q++)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[rt][ct][q]), rW, (uint32_t)(ct * 64 + lane) * 4u, (uint32_t)((16 * rt + q) * kThreads * 4), 0);
            } else if (mt > 0) {
                const int nw = n0 + 128 * wc;
                float* cbase = p.c_row_major ? p.C + (int64_t)c_row * p.ldc + nw : p.C + c_row + (int64_t)nw * p.ldc;
                const uint32_t jstep = p.c_row_major ? 4u : (uint32_t)p.ldc * 4u;
                const uint32_t mistep = p.c_row_major ? (uint32_t)p.ldc * 128u : 128u;
#pragma unroll
                for (int ct = 0; ct < 4; ct++) {                                   // groups of 32 columns, each with its own scalar base (the per-lane offsets span 32 columns of C)
                    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.c_row_major ? cbase + 32 * ct : cbase + (int64_t)(32 * ct) * p.ldc, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) {
                        if (rt * 32 + lm < mt) {
                            float v[16];
#pragma unroll
                            for (int q = 0; q < 16; q++) v[q] = acc[rt][ct][q];
                            if (p.accumulate) {
                                uint32_t old[16];
#pragma unroll
                                for (int q = 0; q < 16; q++) old[q] = __builtin_amdgcn_raw_buffer_load_b32(rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)rt * mistep, 0);
#pragma unroll
                                for (int q = 0; q < 16; q++) v[q] += __uint_as_float(old[q]);
                            }
                            if (p.c_nt) {
#pragma unroll
                                for (int q = 0; q < 16; q++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)rt * mistep, 2);
                            } else {
#pragma unroll
                                for (int q = 0; q < 16; q++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)((q & 3) + 8 * (q >> 2)) * jstep + (uint32_t)rt * mistep, 0);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
                for (int ct = 0; ct < 4; ct++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[rt][ct][r] = 0.0f;
            asm volatile("s_nop 4" : HUB_ACC_ALL);
        }
#pragma unroll
        for (int k = 0; k < AHEAD; k++) { fq[k] = fq[k + 1]; sq[k] = sq[k + 1]; tq[k] = tq[k + 1]; }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    // the loads issued past the end of the range (into LDS nobody reads any more) must land before the workgroup's LDS is handed on
    if constexpr (!LW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef HUB_ACC_ALL
}

template <int KP, int NS, int G, int WPC, bool LW>
void launch_hub_t(bool bf16, bool gathered, dim3 grid, hipStream_t st, const HubParams& p) {
    const dim3 blk(64 * 2 * G * (LW ? 2 : 1));
    if (gathered) {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_hub_kernel<KP, NS, G, true, true, WPC, LW>), grid, blk, 0, st, p);
        else hipLaunchKernelGGL((vbs_spmm_h16_hub_kernel<KP, NS, G, false, true, WPC, LW>), grid, blk, 0, st, p);
    } else {
        if (bf16) hipLaunchKernelGGL((vbs_spmm_h16_hub_kernel<KP, NS, G, true, false, WPC, LW>), grid, blk, 0, st, p);
        else hipLaunchKernelGGL((vbs_spmm_h16_hub_kernel<KP, NS, G, false, false, WPC, LW>), grid, blk, 0, st, p);
    }
}

}  // namespace

namespace sparta_dev {

// variant (developer A/B, scripts/ubench/hub_gemm.hip): G = 2 (128-row group tiles): 0 = KP 64, three stages; 1 = KP 32, three stages, two workgroups per CU;
// 2 = KP 64, two stages; 3 = KP 32, four stages; 4 = variant 0 with four loader waves; 5 = KP 32, six stages, loader waves.
// G = 4 (256-row group tiles, eight waves): 8 = KP 32, five stages (160 KB); 9 = KP 32, four stages; 10 = KP 64, two stages
void launch_h16_hub(int variant, bool bf16, bool gathered, hipStream_t st, const HubParams& p) {
    const dim3 grid((unsigned)(p.n_workers * p.n_slabs));
    switch (variant) {
        case 1: launch_hub_t<32, 3, 2, 2, false>(bf16, gathered, grid, st, p); break;
        case 2: launch_hub_t<64, 2, 2, 1, false>(bf16, gathered, grid, st, p); break;
        case 3: launch_hub_t<32, 4, 2, 1, false>(bf16, gathered, grid, st, p); break;
        case 4: launch_hub_t<64, 3, 2, 1, true>(bf16, gathered, grid, st, p); break;
        case 5: launch_hub_t<32, 6, 2, 1, true>(bf16, gathered, grid, st, p); break;
        case 8: launch_hub_t<32, 5, 4, 1, false>(bf16, gathered, grid, st, p); break;
        case 9: launch_hub_t<32, 4, 4, 1, false>(bf16, gathered, grid, st, p); break;
        case 10: launch_hub_t<64, 2, 4, 1, false>(bf16, gathered, grid, st, p); break;
        default: launch_hub_t<64, 3, 2, 1, false>(bf16, gathered, grid, st, p); break;
    }
}
int hub_variant_kp(int variant) { return (variant == 1 || variant == 3 || variant == 5 || variant == 8 || variant == 9) ? 32 : 64; }
int hub_variant_g(int variant) { return variant >= 8 ? 4 : 2; }

}  // namespace sparta_dev
