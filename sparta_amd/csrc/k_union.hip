// k_union.hip -- the MFMA kernels of the column-compacted ("union-pattern") tiles: vbs_union_f32_kernel (fp32 handles), vbs_union_h16_kernel (16-bit handles).  Part of
// the device side of libsparta_amd.so; see vbs_device.hpp (UnionRec / UnionParams), vbs_union.cpp (the plan) and DESIGN.md section 3.2 (6).
//
// What they multiply.  At small block widths the reference's VBS of a cluster IS a dense (rows x |U|) tile over the union U of the columns its rows touch, stored back
// to back, plus the ascending column list (src/general/vbr.cpp:177-228 at -b 1; its Jaccard distance is defined on those ids: src/general/blocking.cpp:923-994).  The
// hybrid builder keeps such block-rows in exactly that form (vbs_build.cpp, mode 3): tiles of <= 64 consecutive reordered rows, 32 columns of the list per step.
// C[tile rows, :] (+)= Atile . B[U, :] is then a GEMM whose B operand is GATHERED by row: B must be ROW-major here (one contiguous 512-byte piece per list entry and
// 128-column slab; the host transposes the reference's column-major B once per product, or once per sparta_vbs_prepare_b -- the sparse-row kernels read the same copy).
//
// How (fp32).  A workgroup (four waves) owns a tile x one 128-column slab of C; wave v owns columns [32 v, 32 v + 32) x all the tile's rows.  The matrix instruction is
// v_mfma_f32_16x16x4_f32 -- the same rate as the 32 x 32 x 2 form (64 flops per cycle and SIMD) at a row granularity of 16: a tile of mt rows costs ceil(mt / 16) row
// tiles (RT = 1..4, one kernel body each), so a 48-row cluster pays for 48 rows, not 64.  D = Bpanel^T . Atile^T: the "A" operand of the instruction is the panel of B
// (its 16 M rows are 16 columns of C), the "B" operand the slice of A (its 16 N columns are 16 rows of the tile); lane (i = lane & 15, q = lane >> 4) then holds, per
// row tile rt and column tile ct, C[16 rt + i][32 v + 16 ct + 4 q .. + 3] -- four CONSECUTIVE columns of one row.  A step's operands go HBM / L2 -> LDS with LDS-direct
// loads (no staging registers), 1 KB per wave instruction:
//   * the slice of A (16 RT rows x 32 k) is stored in HBM as the LDS image [rt][h][lane][4], lane = 16 kq + i: A[16 rt + i][k = 4 (4 h + e) + kq] -- the values of four
//     consecutive instructions (k-groups 4 h .. 4 h + 3) in one conflict-free ds_read_b128 per lane;
//   * the panel of B is 32 rows of 512 bytes, LDS image Bs[k][128]: wave v fetches rows 8 v .. 8 v + 7 (a row's base is a 64-bit scalar:
//     row id x ldb -- B may be larger than the 4 GB a 32-bit offset spans), two rows per 1 KB piece: lanes 0..31 the even row, lanes 32..63 the odd one (ONE LDS-direct
//     load per piece, per-lane source addresses; it writes lane l at base + 16 l).  An instruction reads, per lane, Bs[4 s + kq][32 v + 16 ct + i]: four rows x 64 bytes, which would sit on
//     the same 16 banks -- so row k is stored with its 16-byte chunks permuted, chunk c at position c ^ (4 (k & 3)) (the swizzle is on the SOURCE side: a lane fetches the
//     chunk that belongs at its position), and the four rows cover all 64 banks;
//   * list positions behind the tile's last column name the tile's first column: a valid row of B against zeros of A (nothing a dense tile does not do anyway).
// One barrier per step, two LDS stages: wait for this wave's loads of step i, barrier, the step's fragments out of LDS, then its MFMAs with the loads of step i + 1 issued
// BETWEEN them (address arithmetic and load issue ride in the shadow of the wave's own matrix instructions).  Measured (profiles/r5/lab_union_stats.txt, lab_union_stages.txt):
// waves neither wait for their loads nor at the barrier -- a step costs its wave the ISSUE of its loads (an LDS-direct load: 100-185 cycles) plus its MFMAs, one after the
// other in the wave's in-order stream; another co-resident workgroup fills the gaps, more LDS stages do not.  Hence three workgroups per CU, as few load instructions per
// step as the bytes allow, and the 16-row instruction.  A worker walks WHOLE tiles (vbs_union.cpp deals them costliest first).  A tile's TAIL -- up to 16 nonzeros per row in
// columns too thinly used for the list (a cluster's rows have a few columns of their own): lane (i, q) holds its row's accumulators of 8 of the wave's 32 columns, so per
// entry it fetches 2 x 16 bytes of ITS row of B and multiplies them in (entry t during the tile's step t + 1, what is left in the last step's epilogue); no sparse-row
// launch, no second pass over the rows of C -- the last step then stores the tile's rows of C (or adds to them: accumulate).  ONE launch carries all tile types: a workgroup runs the body of each type over its tiles of that type, tallest first.
#include "vbs_kernel_common.hpp"

using namespace sparta_dev;

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// developer probes, TIMING ONLY (results wrong), UnionParams::pad = SPARTA_UNION_PROBE read per call: 1 no B loads, 2 no A loads, 8 no tails, 16 no epilogue

// Two LDS stages (one step of loads in flight across the barrier) and THREE workgroups per CU: measured on the benchmark set's clustered family (N = 128, prepared B):
// three stages with two workgroups per CU 172 us, two stages with two 162, two stages with three 160 -- another co-resident workgroup hides a workgroup's waits
// (barrier, tile epilogue with its tail gathers) better than a deeper pipeline of its own (profiles/r5/lab_union_stages.txt).
#ifndef SPARTA_UNION_TAILPIPE
#define SPARTA_UNION_TAILPIPE 1   /* developer A/B: 0 = every tail entry is added in the tile's epilogue */
#endif
#ifndef SPARTA_UNION_STATS
#define SPARTA_UNION_STATS 0      /* lab build: per-wave cycle sums written over the head of C (scripts/lab/r5_union_stats.py) */
#endif
// Two LDS stages (one step of loads in flight across the barrier) and THREE workgroups per CU.  Measured on 2000 clusters x 48 rows (N = 128, prepared B; profiles/r5/lab_union_stages.txt,
// lab_union_16.txt): three workgroups x two stages 113 us, two x three 128, two x two 124, one x six 202 -- and a launch of at most one workgroup per CU (200 tiles) runs its ten steps in
// 23 us with two stages, 31 with six: a step is not waiting for its loads, so more of them in flight buy nothing, and another co-resident workgroup does.
constexpr int kUnionStages = 2;
constexpr int kUnionLds = kUnionStages * (4 * 2048 + kUnionPairFloats * 4 + 32 * 512);      // the tallest type's stages: rows of A, the KB of tail pairs, the panel of B

template <int RT, int NS>
__device__ __forceinline__ void union_body(const UnionParams& p, const UnionSide& sd, const int worker, char* const lds) {
    static_assert(RT >= 1 && RT <= 4, "one to four 16-row MFMA tiles per wave and column tile");
    static_assert(NS == 2, "the step's top wait is vmcnt(0): one step of loads in flight");
    constexpr int A_BYTES = RT * 2048 + kUnionPairFloats * 4, B_BYTES = 32 * 512, STAGE = A_BYTES + B_BYTES;      // (the slice of A + the KB of tail pairs behind it)
    constexpr int NPA = 2 * RT + 1;                      // 1 KB pieces of the slice per step, dealt to the waves round robin
    constexpr int SLICE_F = RT * 512 + kUnionPairFloats; // floats of a step's slice in memory
    constexpr int AHEAD = NS - 1;                        // steps between a step's loads and its MFMAs
    constexpr bool TAILPIPE = SPARTA_UNION_TAILPIPE != 0;
    static_assert(NS * STAGE <= kUnionLds, "LDS");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int n0 = (int)blockIdx.y * kTN;
    const int probe = p.pad;
    const int s_begin = sd.worker_range[2 * worker];
    const int n = sd.worker_range[2 * worker + 1] - s_begin;
    if (n <= 0) return;

    // step records and list entries through the constant address space (scalar loads): per step this wave needs the record (three dwords) and ITS eight row ids
    typedef const __attribute__((address_space(4))) int32_t* cptr_t;
    const cptr_t srec = (cptr_t)(reinterpret_cast<const int32_t*>(sd.rec + s_begin));
    const cptr_t sids = (cptr_t)(sd.ids + (int64_t)s_begin * 32 + 8 * wave);
    struct Rec { int32_t c_row, info, tail_off, id[8]; };
    auto load_rec = [&](int j) __attribute__((always_inline)) -> Rec {
        Rec r;
        r.c_row = srec[(int64_t)j * 4]; r.info = srec[(int64_t)j * 4 + 1]; r.tail_off = srec[(int64_t)j * 4 + 2];
#pragma unroll
        for (int q = 0; q < 8; q++) r.id[q] = sids[(int64_t)j * 32 + q];
        return r;
    };

    // ---- per-lane / per-wave constants ----
    const int ncv = p.n_cols - n0;                                               // columns of this slab that exist (the last slab of a call may be narrower than 128)
    const uint32_t row_bytes = ncv >= kTN ? 512u : (uint32_t)((ncv * 4 + 15) & ~15);   // (whole 16-byte chunks: ldb % 4 == 0, so the chunk is inside the row's allocation)
    // row k = 8 wave + 2 r2 + (lane >> 5) of the panel: the lane at position l of the row fetches chunk l ^ (4 (k & 3)); k & 3 depends on r2 & 1 and the lane's half only
    uint32_t voffB[2];
#pragma unroll
    for (int par = 0; par < 2; par++) voffB[par] = (uint32_t)(((lane & 31) ^ (4 * ((2 * par + (lane >> 5)) & 3))) * 16);
    const uint32_t voffA = (uint32_t)lane * 16u;
    const float* const Bs0 = p.B + n0;
    const float* const A0 = reinterpret_cast<const float*>(sd.A) + (int64_t)s_begin * SLICE_F;
    // fragment reads of B: sub-step s (k = 4 s + kq), column tile ct: column 32 wave + 16 ct + i of row k, i.e. chunk 8 wave + 4 ct + (i >> 2) at position chunk ^ (4 kq)
    uint32_t rdB[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) rdB[ct] = (uint32_t)(kq * 512 + (((8 * wave + 4 * ct + (li >> 2)) ^ (4 * kq)) * 16) + (li & 3) * 4);
    char* const lds0 = lds;

    // the loads of one step (its record already in scalar registers) into `stage`: the wave's pieces of the slice of A, then its eight rows of B two at a time
    auto issue_a = [&](int j, int stage) __attribute__((always_inline)) {
        if (probe & 2) return;
        char* const stp = lds0 + stage * STAGE;
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A0 + (int64_t)j * SLICE_F), 0, A_BYTES, 0x00020000);
#pragma unroll
        for (int t = 0; t < (NPA + 3) / 4; t++) {
            const int q = wave + 4 * t;
            if (q < NPA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(stp + q * 1024), 16, voffA, (uint32_t)(q * 1024), 0, 0);
        }
    };
    // lanes 0..31 fetch the even row of a 1 KB piece, lanes 32..63 the odd one -- ONE LDS-direct load per piece with a per-lane source address (global_load_lds_dwordx4:
    // the row's base is a 64-bit scalar, row id x ldb; the lane picks its half's base and adds its chunk).  An LDS-direct load costs the issuing wave 100-185 cycles
    // beside MFMAs and LDS reads (MI355X_MICROARCH.md, cycle constants) and the wave's MFMAs queue behind it: the earlier form -- two loads per piece under half an exec
    // mask each, every row through its own buffer descriptor -- spent more of a step issuing its eight loads than multiplying (profiles/r5/lab_union_stages.txt).
    // List positions behind the tile's last column name the tile's FIRST column (vbs_union.cpp): a valid row of B against zeros of A.
    const bool full_rows = row_bytes == 512u;                    // (the last slab of a ragged N: lanes behind the row's last chunk load nothing)
    typedef const __attribute__((address_space(1))) void* gsrc_t;
    auto issue_b = [&](const Rec& rec, int stage, int r2) __attribute__((always_inline)) {
        if (probe & 1) return;
        char* const stp = lds0 + stage * STAGE;
        const char* const be = reinterpret_cast<const char*>(Bs0 + (int64_t)rec.id[2 * r2] * p.ldb);
        const char* const bo = reinterpret_cast<const char*>(Bs0 + (int64_t)rec.id[2 * r2 + 1] * p.ldb);
        const char* const src = (lane < 32 ? be : bo) + voffB[r2 & 1];
        const lds_ptr_t dst = (lds_ptr_t)(stp + A_BYTES + (4 * wave + r2) * 1024);
        if (full_rows || voffB[r2 & 1] < row_bytes) __builtin_amdgcn_global_load_lds((gsrc_t)src, dst, 16, 0, 0);
    };

    f32x4 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) acc[rt][ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // records of the steps in flight: info / c_row of step i are needed at its epilogue, long after its loads were issued
    int32_t iq[NS], cq[NS], tq[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) { iq[k] = 0; cq[k] = 0; tq[k] = 0; }
#pragma unroll
    for (int k = 0; k < AHEAD; k++) {                    // prologue: steps 0 .. AHEAD - 1
        const Rec r = load_rec(k);
        iq[k] = r.info; cq[k] = r.c_row; tq[k] = r.tail_off;
        issue_a(k, k);
#pragma unroll
        for (int r2 = 0; r2 < 4; r2++) issue_b(r, k, r2);
    }
    Rec nxt = load_rec(AHEAD);                           // the record of the step whose loads the next iteration issues

    // C: lane (i, q) owns row 16 rt + i, columns 32 wave + 16 ct + 4 q + r
    const uint32_t voffC = p.c_row_major ? (uint32_t)((li * p.ldc + 4 * kq) * 4) : (uint32_t)((li + (4 * kq) * p.ldc) * 4);
    const uint32_t jstep = p.c_row_major ? 4u : (uint32_t)p.ldc * 4u;            // bytes per output column
    const uint32_t rstep = p.c_row_major ? (uint32_t)p.ldc * 64u : 64u;          // bytes per 16 rows
    const int ncw = ncv - 32 * wave;                                             // columns of this wave that exist

    // tails in the pipeline (two-stage form): the tile's step t requests, per lane, the pieces of B of tail entry t (2 x 16 bytes of ITS row's column; the entry's (column,
    // value) pairs came with the step's slice of A) ahead of the step's panel loads, so they do not queue behind them; step t + 1 multiplies entry t in before its MFMAs
    // (the step's top wait covers them).  A round trip per entry then hides behind a step instead of standing between the tile's last MFMA and its stores; what the steps cannot carry (entries
    // S - 1 .. of a tile of S steps) is added in the epilogue.
    int tstep = 0;                                       // the current step's index inside its tile
    bool tb_live = false;                                // an entry's pieces were requested by the previous step
    float tv[RT];
    f32x4 tb[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        tv[rt] = 0.0f;
#pragma unroll
        for (int ct = 0; ct < 2; ct++) tb[rt][ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    const float* const brow_t = p.B + n0 + 32 * wave + 4 * kq;
#if SPARTA_UNION_STATS
    // lab build only (results WRONG: the sums overwrite the head of C): cycles this wave spent waiting for its loads / at the barrier / in the multiply phase / in epilogues
    uint64_t st_wait = 0, st_bar = 0, st_mul = 0, st_epi = 0;
    const uint64_t st_begin = __builtin_readcyclecounter();
#define UNION_STAMP(var) const uint64_t var = __builtin_readcyclecounter()
#else
#define UNION_STAMP(var)
#endif
    int stage = 0;
    for (int i = 0; i < n; i++) {
        UNION_STAMP(st0);
        // (1) this wave's loads of step i have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UNION_STAMP(st1);
        // (2) everybody's have, and everybody is done with the stage of step i - 1
        __builtin_amdgcn_s_barrier();
        UNION_STAMP(st2);
        // (3) the record of step i + AHEAD (its loads go into the stage step i - 1 has left); the record of the step behind it is requested further down, for the next iteration
        const Rec rec = nxt;
        const int jstage = stage ^ 1;
        iq[AHEAD] = rec.info; cq[AHEAD] = rec.c_row; tq[AHEAD] = rec.tail_off;
        if constexpr (TAILPIPE) {
            // (the compiler counts vmcnt for the loads it emitted and does not see the wait above: without telling it that these registers have landed it waits for them
            // in the MIDDLE of the requests below -- with loads under branches in between it cannot count, so it drains: a round trip per step, exposed)
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) asm volatile("" : : "v"(tb[rt][ct]));        // (a USE, not a redefinition: loop-carried registers that an asm redefines get copied at the back edge -- behind a wait)
            const int32_t inf = iq[0];
            if (tb_live) {                                 // the entry the previous step requested
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int ct = 0; ct < 2; ct++)
#pragma unroll
                        for (int i2 = 0; i2 < 4; i2++) acc[rt][ct][i2] = __builtin_fmaf(tv[rt], tb[rt][ct][i2], acc[rt][ct][i2]);
            }
            tb_live = !(probe & 8) && !(inf & UREC_LAST) && tstep < ((inf >> UREC_TAIL_SHIFT) & 31);
            if (tb_live) {
                // the entry's (column, value) pairs came with the step's slice: [rt][row] uint2 behind the rows of A
                const uint2* const pairs = reinterpret_cast<const uint2*>(lds0 + stage * STAGE + RT * 2048) + li;
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const uint2 cv = pairs[rt * 16];
                    tv[rt] = __uint_as_float(cv.y);
                    const float* bp = brow_t + (int64_t)cv.x * p.ldb;
#pragma unroll
                    for (int ct = 0; ct < 2; ct++) tb[rt][ct] = 16 * ct + 4 * kq < ncw ? *reinterpret_cast<const f32x4*>(bp + 16 * ct) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
        }
        // (4) multiply step i: the fragments first (the LDS reads of the whole step), then 16 RT matrix instructions with the loads of step i + 1 issued between them
        {
            const char* const sa = lds0 + stage * STAGE;
            const char* const sb = sa + A_BYTES;
            f32x4 af[RT][2];
            float bq[8][2];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int h = 0; h < 2; h++) af[rt][h] = *reinterpret_cast<const f32x4*>(sa + (rt * 2 + h) * 1024 + lane * 16);
#pragma unroll
            for (int s = 0; s < 8; s++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) bq[s][ct] = *reinterpret_cast<const float*>(sb + rdB[ct] + s * 2048);
            // every fragment has landed before the first MFMA, and the compiler is told so: the scalar loads of the next record go out right behind, and a later partial
            // wait for a fragment (lgkmcnt counts LDS reads and scalar loads alike; with a scalar load outstanding every wait is a wait for all) would stand through them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int h = 0; h < 2; h++) asm volatile("" : : "v"(af[rt][h]));
#pragma unroll
            for (int s = 0; s < 8; s++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) asm volatile("" : : "v"(bq[s][ct]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; s++) {
#pragma unroll
                for (int ct = 0; ct < 2; ct++) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[s][ct], af[rt][s >> 2][s & 3], acc[rt][ct], 0, 0, 0);
                    const int slot = 2 * s + ct;
                    // the next record's scalar loads go out BEHIND the wait for this step's fragments: lgkmcnt counts LDS reads and scalar loads alike, and a wait with a scalar
                    // load outstanding is a wait for everything -- requested ahead of the fragment reads, every step stood through a scalar-cache miss
                    if (slot == 0) { nxt = load_rec(i + AHEAD + 1); issue_a(i + AHEAD, jstage); }
                    if (slot == 2 || slot == 5 || slot == 8 || slot == 11) issue_b(rec, jstage, (slot - 2) / 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        UNION_STAMP(st3);
        const int32_t info = iq[0];
        if ((info & UREC_LAST) && !(probe & 16)) {
            const int mt = info & 127;
            const int64_t c_row = cq[0];
            // the tail: per entry and row 2 x 16 bytes of the row of B -- the 8 columns whose accumulators this lane holds
            const int tail_e = (info >> UREC_TAIL_SHIFT) & 31;
            if (tail_e > 0 && !(probe & 8)) {
                const uint2* tp = sd.tail + tq[0] + li;
                const float* brow = p.B + n0 + 32 * wave + 4 * kq;
                // two entries at a time: their (column, value) pairs in one round trip, their 2 x RT x 2 pieces of B in a second one (an entry at a time is two round
                // trips per entry with the matrix pipe idle; more at a time -- or two for the tallest type -- costs the third workgroup per CU its registers)
                constexpr int CH = RT == 4 ? 1 : 2;
                const int e_first = TAILPIPE ? (tstep < tail_e ? tstep : tail_e) : 0;          // entries 0 .. S - 2 rode in the steps
                for (int e0 = e_first; e0 < tail_e; e0 += CH) {
                    uint2 cv[CH][RT];
#pragma unroll
                    for (int c = 0; c < CH; c++)
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) cv[c][rt] = e0 + c < tail_e ? tp[((e0 + c) * RT + rt) * 16] : uint2{0u, 0u};      // (behind the last entry: nothing is fetched)
                    f32x4 bv[CH][RT][2];
#pragma unroll
                    for (int c = 0; c < CH; c++)
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) {
                            const float* bp = brow + (int64_t)cv[c][rt].x * p.ldb;
#pragma unroll
                            for (int ct = 0; ct < 2; ct++) bv[c][rt][ct] = (e0 + c < tail_e && 16 * ct + 4 * kq < ncw) ? *reinterpret_cast<const f32x4*>(bp + 16 * ct) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        }
#pragma unroll
                    for (int c = 0; c < CH; c++)
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) {
                            const float av = __uint_as_float(cv[c][rt].y);
#pragma unroll
                            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                                for (int i2 = 0; i2 < 4; i2++) acc[rt][ct][i2] = __builtin_fmaf(av, bv[c][rt][ct][i2], acc[rt][ct][i2]);
                        }
                }
            }
            float* cbase = p.c_row_major ? p.C + c_row * p.ldc + (n0 + 32 * wave) : p.C + c_row + (int64_t)(n0 + 32 * wave) * p.ldc;
            const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                if (rt * 16 + li < mt) {
                    float v[2][4];
#pragma unroll
                    for (int ct = 0; ct < 2; ct++)
#pragma unroll
                        for (int r = 0; r < 4; r++) v[ct][r] = acc[rt][ct][r];
                    if (p.accumulate) {
                        uint32_t old[2][4];
#pragma unroll
                        for (int ct = 0; ct < 2; ct++)
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                old[ct][r] = 16 * ct + 4 * kq + r < ncw ? __builtin_amdgcn_raw_buffer_load_b32(rC, voffC, (uint32_t)(16 * ct + r) * jstep + (uint32_t)rt * rstep, 0) : 0u;
#pragma unroll
                        for (int ct = 0; ct < 2; ct++)
#pragma unroll
                            for (int r = 0; r < 4; r++) v[ct][r] += __uint_as_float(old[ct][r]);
                    }
#pragma unroll
                    for (int ct = 0; ct < 2; ct++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if (16 * ct + 4 * kq + r < ncw) {
                                if (sd.c_nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[ct][r]), rC, voffC, (uint32_t)(16 * ct + r) * jstep + (uint32_t)rt * rstep, 2);
                                else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[ct][r]), rC, voffC, (uint32_t)(16 * ct + r) * jstep + (uint32_t)rt * rstep, 0);
                            }
                        }
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) acc[rt][ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int k = 0; k < AHEAD; k++) { iq[k] = iq[k + 1]; cq[k] = cq[k + 1]; tq[k] = tq[k + 1]; }
        stage ^= 1;
        if constexpr (TAILPIPE) tstep = (info & UREC_LAST) ? 0 : tstep + 1;
#if SPARTA_UNION_STATS
        const uint64_t st4 = __builtin_readcyclecounter();
        st_wait += st1 - st0; st_bar += st2 - st1; st_mul += st3 - st2; st_epi += st4 - st3;
#endif
    }
#if SPARTA_UNION_STATS
    if (lane == 0) {
        float* o = p.C + ((int64_t)(blockIdx.x * 4 + wave)) * 8;
        o[0] = (float)st_wait; o[1] = (float)st_bar; o[2] = (float)st_mul; o[3] = (float)st_epi; o[4] = (float)(__builtin_readcyclecounter() - st_begin); o[5] = (float)n;
    }
#endif
    // the loads issued past the end of the range (into LDS nobody reads any more) must land before the workgroup's LDS is handed on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// =====================================================================================================================================================================
// 16-bit handles (fp16 / bf16 storage of A and B, fp32 accumulation and C): the same tiles, the same pipeline, the 16-bit matrix instruction.
//   * the panel of B is 32 rows of 256 bytes (128 columns x 2 bytes) of the ROW-major 16-bit copy of B, LDS image Bs[k][128]; wave v fetches rows 8 v .. 8 v + 7, four rows per
//     1 KB piece: lanes 16 r' .. 16 r' + 15 the piece's row r' (ONE LDS-direct load per piece, per-lane source addresses);
//   * `v_mfma_f32_32x32x16_{f16,bf16}` wants, per lane, 8 consecutive k of ITS column of the panel -- a column of a k-major image.  gfx950's `ds_read_b64_tr_b16` is that transpose:
//     per group of 16 lanes it reads a block of 4 rows x 16 columns and hands lane i column i of the 4 rows (lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3);
//     two such reads per MFMA and lane.  Bank-conflict-free with 16-byte chunk ch of row r stored at position ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) (cdna_hip_programming.md
//     T10, image (b)); the LDS-direct load writes lane-linear, so the swizzle is on the SOURCE side: lane l of a row's 16 fetches chunk l ^ s(r);
//   * the slice of A (32 MI rows x 32 k) is stored in HBM as the LDS image [rt][m][kg][row][8] = A[32 rt + row][k = 16 m + 8 kg + e]: one `ds_read_b128` per MFMA and lane;
//   * the tails multiply in fp32: value (the rounded 16-bit value of A, kept as fp32) x the 16-bit entries of the row of B widened in registers -- products of two 16-bit
//     values are exact in fp32, as in the matrix instruction.
// Two LDS stages of 12 KB (4 KB of A -- two of them unused by the <= 32-row tiles -- + 8 KB of B), three workgroups per CU, the tails in the pipeline (measured, N = 128 / 512 on
// 2000 clusters: three stages with every tail entry in the epilogue 67.7 / 242.5 us, two stages with the tail pipeline 61.9 / 221.5: profiles/r5/lab_union16.txt).
#ifndef SPARTA_UNION16_STAGES
#define SPARTA_UNION16_STAGES 2     /* developer A/B (with SPARTA_UNION16_WPC = workgroups per CU of the launch bound; the plan's workers: SPARTA_UNION_WPC at create time); the tail pipeline needs 2 */
#endif
#ifndef SPARTA_UNION16_WPC
#define SPARTA_UNION16_WPC 3
#endif
constexpr int kUnion16Stages = SPARTA_UNION16_STAGES;
constexpr int kUnion16Lds = kUnion16Stages * (4096 + 32 * 256);
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

template <int MI, bool BF16>
__device__ __forceinline__ void union16_body(const UnionParams& p, const UnionSide& sd, const int worker, char* const lds) {
    constexpr int NS = kUnion16Stages;
    constexpr int A_BYTES = 4096, B_BYTES = 32 * 256, STAGE = A_BYTES + B_BYTES;
    constexpr int LPS = 2 + 1;                           // vector-memory instructions per wave and step: its two pieces of B (four rows each), one piece of A (waves behind the slice's pieces: zeros)
    constexpr int AHEAD = NS - 1;
    static_assert((AHEAD - 1) * LPS <= 63, "vmcnt holds 6 bits");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, g = lane >> 5;
    const int n0 = (int)blockIdx.y * kTN;
    const int probe = p.pad;
    const int s_begin = sd.worker_range[2 * worker];
    const int n = sd.worker_range[2 * worker + 1] - s_begin;
    if (n <= 0) return;

    typedef const __attribute__((address_space(4))) int32_t* cptr_t;
    const cptr_t srec = (cptr_t)(reinterpret_cast<const int32_t*>(sd.rec + s_begin));
    const cptr_t sids = (cptr_t)(sd.ids + (int64_t)s_begin * 32 + 8 * wave);
    struct Rec { int32_t c_row, info, tail_off, id[8]; };
    auto load_rec = [&](int j) __attribute__((always_inline)) -> Rec {
        Rec r;
        r.c_row = srec[(int64_t)j * 4]; r.info = srec[(int64_t)j * 4 + 1]; r.tail_off = srec[(int64_t)j * 4 + 2];
#pragma unroll
        for (int q = 0; q < 8; q++) r.id[q] = sids[(int64_t)j * 32 + q];
        return r;
    };

    const int ncv = p.n_cols - n0;
    const uint32_t row_bytes = ncv >= kTN ? 256u : (uint32_t)((ncv * 2 + 15) & ~15);        // (whole 16-byte chunks: ldb % 8 == 0)
    const uint16_t* const B16 = reinterpret_cast<const uint16_t*>(p.B);
    const uint16_t* const Bs0 = B16 + n0;
    const uint16_t* const A0 = reinterpret_cast<const uint16_t*>(sd.A) + (int64_t)s_begin * (MI * 1024);
    // B loads: piece pc of this wave's two holds the panel rows k = 8 wave + 4 pc + q, q = lane >> 4; lane l of a row's 16 fetches source chunk l ^ s(k)
    uint32_t voffB[2];
#pragma unroll
    for (int pc = 0; pc < 2; pc++) { const int k = 8 * wave + 4 * pc + (lane >> 4); voffB[pc] = (uint32_t)((((lane & 15) ^ (((k & 3) << 2) | ((k >> 2) & 3)))) * 16); }
    const uint32_t voffA = (uint32_t)lane * 16u;
    // transposed fragment reads: MFMA m (k = 16 m ..), half h (rows + 4 h): the block of rows r0 = 16 m + 8 g + 4 h .. + 3, columns 32 wave + 16 gq .. + 15 (chunks c0, c0 + 1);
    // lane 4 q + pp of the group: row r0 + q, chunk c0 + (pp >> 1), + 8 (pp & 1) bytes
    uint32_t rdB[2][2];
    {
        const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, gq = (lane >> 4) & 1;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int row = 16 * m + 8 * g + 4 * h + q, ch = 4 * wave + 2 * gq + (pp >> 1);
                rdB[m][h] = (uint32_t)(256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 8 * (pp & 1));
            }
    }
    char* const lds0 = lds;

    auto issue = [&](const Rec& rec, int j, int stage) __attribute__((always_inline)) {
        char* const stp = lds0 + stage * STAGE;
        if (!(probe & 2)) {
            // piece `wave` of the slice (MI x 2 pieces of 1 KB; behind them: past the descriptor's end -- zeros in LDS, no access; the A area holds four pieces whatever MI)
            const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A0 + (int64_t)j * (MI * 1024)), 0, MI * 2048, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(stp + wave * 1024), 16, voffA, (uint32_t)(wave * 1024), 0, 0);
        }
        if (!(probe & 1)) {
            // two pieces of four rows, ONE LDS-direct load each: lanes [16 q, 16 q + 16) fetch the piece's row q through a per-lane source address (the row's base is a 64-bit
            // scalar, row id x ldb; the lane picks its row's).  An LDS-direct load costs the issuing wave 100-185 cycles: four quarter-wave descriptor loads per piece spent
            // more of a step issuing loads than anything else.  List positions behind the tile's last column name its first column (vbs_union.cpp).
            typedef const __attribute__((address_space(1))) void* gsrc_t;
            const int q = lane >> 4;
#pragma unroll
            for (int pc = 0; pc < 2; pc++) {
                const char* b0 = reinterpret_cast<const char*>(Bs0 + (int64_t)rec.id[4 * pc] * p.ldb);
                const char* b1 = reinterpret_cast<const char*>(Bs0 + (int64_t)rec.id[4 * pc + 1] * p.ldb);
                const char* b2 = reinterpret_cast<const char*>(Bs0 + (int64_t)rec.id[4 * pc + 2] * p.ldb);
                const char* b3 = reinterpret_cast<const char*>(Bs0 + (int64_t)rec.id[4 * pc + 3] * p.ldb);
                const char* const src = (q == 0 ? b0 : (q == 1 ? b1 : (q == 2 ? b2 : b3))) + voffB[pc];
                const lds_ptr_t dst = (lds_ptr_t)(stp + A_BYTES + (2 * wave + pc) * 1024);
                if (row_bytes == 256u || voffB[pc] < row_bytes) __builtin_amdgcn_global_load_lds((gsrc_t)src, dst, 16, 0, 0);
            }
        }
    };

    f32x16 acc[MI];
#pragma unroll
    for (int rt = 0; rt < MI; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[rt][r] = 0.0f;

    int32_t iq[NS], cq[NS], tq[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) { iq[k] = 0; cq[k] = 0; tq[k] = 0; }
#pragma unroll
    for (int k = 0; k < AHEAD; k++) {
        const Rec r = load_rec(k);
        iq[k] = r.info; cq[k] = r.c_row; tq[k] = r.tail_off;
        issue(r, k, k);
    }
    Rec nxt = load_rec(AHEAD);

    const uint32_t voffC = p.c_row_major ? (uint32_t)((lm * p.ldc + 4 * g) * 4) : (uint32_t)((lm + (4 * g) * p.ldc) * 4);
    const uint32_t jstep = p.c_row_major ? 4u : (uint32_t)p.ldc * 4u;
    const uint32_t mistep = p.c_row_major ? (uint32_t)p.ldc * 128u : 128u;
    const int ncw = ncv - 32 * wave;
    auto widen = [](uint32_t u16) __attribute__((always_inline)) -> float {            // one 16-bit element of B -> fp32
        if constexpr (BF16) return __uint_as_float(u16 << 16);
        else { const uint16_t h = (uint16_t)u16; _Float16 x; __builtin_memcpy(&x, &h, 2); return (float)x; }
    };

    // tails in the pipeline (two-stage build only: the step's top wait is then vmcnt(0) and covers them), as in the fp32 kernel: step t of a tile requests the chunks of B of tail
    // entry t and the (column, value) pair of entry t + 1 ahead of the step's panel loads; step t + 1 trades halves and multiplies entry t in
    constexpr bool TAILPIPE = NS == 2 && SPARTA_UNION_TAILPIPE != 0;
    int tstep = 0;
    bool tb_live = false;
    uint2 cvn[MI];
    float tv[MI];
    u32x4 tbc[MI][2];
#pragma unroll
    for (int rt = 0; rt < MI; rt++) {
        cvn[rt] = uint2{0u, 0u}; tv[rt] = 0.0f;
#pragma unroll
        for (int hh = 0; hh < 2; hh++) tbc[rt][hh] = u32x4{0u, 0u, 0u, 0u};
    }
    const uint16_t* const brow16_t = B16 + n0 + 32 * wave + 8 * g;
    if (TAILPIPE && !(probe & 8) && !(iq[0] & UREC_LAST) && ((iq[0] >> UREC_TAIL_SHIFT) & 31) > 0) {
#pragma unroll
        for (int rt = 0; rt < MI; rt++) cvn[rt] = sd.tail[tq[0] + rt * 32 + lm];
    }
    // one tail entry's chunks (lane g holds chunks g and 2 + g of its row's four) -> the lane's four pieces of four columns (halves traded with v_permlane32_swap) -> accumulators
    auto tail_fma = [&](const float (&av)[MI], const u32x4 (&chunks)[MI][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < MI; rt++)
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(chunks[rt][hh][0], chunks[rt][hh][2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(chunks[rt][hh][1], chunks[rt][hh][3], false, false);
                const uint2 w[2] = {uint2{s0[0], s1[0]}, uint2{s0[1], s1[1]}};
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int qq = 2 * hh + u;
                    acc[rt][4 * qq + 0] = __builtin_fmaf(av[rt], widen(w[u].x & 0xffffu), acc[rt][4 * qq + 0]);
                    acc[rt][4 * qq + 1] = __builtin_fmaf(av[rt], widen(w[u].x >> 16), acc[rt][4 * qq + 1]);
                    acc[rt][4 * qq + 2] = __builtin_fmaf(av[rt], widen(w[u].y & 0xffffu), acc[rt][4 * qq + 2]);
                    acc[rt][4 * qq + 3] = __builtin_fmaf(av[rt], widen(w[u].y >> 16), acc[rt][4 * qq + 3]);
                }
            }
    };

    int stage = 0;
    for (int i = 0; i < n; i++) {
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((AHEAD - 1) * LPS) : "memory");
        __builtin_amdgcn_s_barrier();
        const Rec rec = nxt;
        int jstage = stage + AHEAD; if (jstage >= NS) jstage -= NS;
        iq[AHEAD] = rec.info; cq[AHEAD] = rec.c_row; tq[AHEAD] = rec.tail_off;
        if constexpr (TAILPIPE) {
            // (the compiler is shown that these registers have landed -- it does not see the wait above and would otherwise drain in the middle of the requests below)
#pragma unroll
            for (int rt = 0; rt < MI; rt++) {
                asm volatile("" : : "v"(cvn[rt].x), "v"(cvn[rt].y));
#pragma unroll
                for (int hh = 0; hh < 2; hh++) asm volatile("" : : "v"(tbc[rt][hh]));
            }
            const int32_t inf = iq[0], inf1 = iq[1];
            if (tb_live) tail_fma(tv, tbc);
            tb_live = !(probe & 8) && !(inf & UREC_LAST) && tstep < ((inf >> UREC_TAIL_SHIFT) & 31);
            if (tb_live) {
#pragma unroll
                for (int rt = 0; rt < MI; rt++) {
                    tv[rt] = __uint_as_float(cvn[rt].y);
                    const uint16_t* bp = brow16_t + (int64_t)cvn[rt].x * p.ldb;
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) tbc[rt][hh] = 8 * (2 * hh + g) < ncw ? *reinterpret_cast<const u32x4*>(bp + 16 * hh) : u32x4{0u, 0u, 0u, 0u};
                }
            }
            const int tn = (inf & UREC_LAST) ? 0 : tstep + 1;
            if (!(probe & 8) && !(inf1 & UREC_LAST) && tn < ((inf1 >> UREC_TAIL_SHIFT) & 31)) {
#pragma unroll
                for (int rt = 0; rt < MI; rt++) cvn[rt] = sd.tail[tq[1] + (tn * MI + rt) * 32 + lm];
            }
        }
        // the step's fragments first, and all of them landed (lgkmcnt counts LDS reads and scalar loads alike: with the next record's scalar loads outstanding, the wait for a
        // fragment would be a wait for them too -- a scalar-cache miss per step); then the next record's scalar loads, this step's requests, and the MFMAs
        s16x8 bf[2], af[MI][2];
        {
            const char* const sa = lds0 + stage * STAGE;
            typedef __attribute__((address_space(3))) s16x4* tr_ptr_t;
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(lds_ptr_t)(sa + A_BYTES + rdB[m][0]));
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(lds_ptr_t)(sa + A_BYTES + rdB[m][1]));
                bf[m] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int rt = 0; rt < MI; rt++) af[rt][m] = *reinterpret_cast<const s16x8*>(sa + (rt * 2 + m) * 1024 + lane * 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int m = 0; m < 2; m++) {
                asm volatile("" : : "v"(bf[m]));
#pragma unroll
                for (int rt = 0; rt < MI; rt++) asm volatile("" : : "v"(af[rt][m]));
            }
        }
        nxt = load_rec(i + AHEAD + 1);
        issue(rec, i + AHEAD, jstage);
        if (!(probe & 4)) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int rt = 0; rt < MI; rt++) {
                    if constexpr (BF16) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bf[m]), __builtin_bit_cast(bf16x8_t, af[rt][m]), acc[rt], 0, 0, 0);
                    else acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, bf[m]), __builtin_bit_cast(f16x8_t, af[rt][m]), acc[rt], 0, 0, 0);
                }
        }
        const int32_t info = iq[0];
        if ((info & UREC_LAST) && !(probe & 16)) {
            const int mt = info & 127;
            const int64_t c_row = cq[0];
            const int tail_e = (info >> UREC_TAIL_SHIFT) & 31;
            if (tail_e > 0 && !(probe & 8)) {
                const uint2* tp = sd.tail + tq[0] + lm;
                // the lane wants four pieces of four consecutive 16-bit columns (8 bytes at 16 qq + 8 g of the wave's 64 bytes of the row).  Fetched as such, an instruction
                // touches 32 rows for 16 bytes each.  Instead the two lanes of a row (lane, lane + 32) fetch whole 16-byte chunks -- lane g chunk 2 hh + g -- and trade
                // halves with v_permlane32_swap (lanes 32..63 of one register <-> lanes 0..31 of another): half the vector-memory instructions, 32 bytes per row each
                constexpr int CH = 2;
                const int e_first = TAILPIPE ? (tstep < tail_e ? tstep : tail_e) : 0;          // entries 0 .. S - 2 rode in the steps
                for (int e0 = e_first; e0 < tail_e; e0 += CH) {
                    uint2 cv[CH][MI];
#pragma unroll
                    for (int c = 0; c < CH; c++)
#pragma unroll
                        for (int rt = 0; rt < MI; rt++) cv[c][rt] = e0 + c < tail_e ? tp[((e0 + c) * MI + rt) * 32] : uint2{0u, 0u};
                    u32x4 ch[CH][MI][2];
                    float av[CH][MI];
#pragma unroll
                    for (int c = 0; c < CH; c++)
#pragma unroll
                        for (int rt = 0; rt < MI; rt++) {
                            av[c][rt] = __uint_as_float(cv[c][rt].y);
                            const uint16_t* bp = brow16_t + (int64_t)cv[c][rt].x * p.ldb;
#pragma unroll
                            for (int hh = 0; hh < 2; hh++)
                                ch[c][rt][hh] = (e0 + c < tail_e && 8 * (2 * hh + g) < ncw) ? *reinterpret_cast<const u32x4*>(bp + 16 * hh) : u32x4{0u, 0u, 0u, 0u};
                        }
#pragma unroll
                    for (int c = 0; c < CH; c++) tail_fma(av[c], ch[c]);
                }
            }
            float* cbase = p.c_row_major ? p.C + c_row * p.ldc + (n0 + 32 * wave) : p.C + c_row + (int64_t)(n0 + 32 * wave) * p.ldc;
            const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
            for (int rt = 0; rt < MI; rt++) {
                if (rt * 32 + lm < mt) {
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) v[q] = acc[rt][q];
                    if (p.accumulate) {
                        uint32_t old[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) {
                            const int col = (q & 3) + 8 * (q >> 2);
                            old[q] = col + 4 * g < ncw ? __builtin_amdgcn_raw_buffer_load_b32(rC, voffC, (uint32_t)col * jstep + (uint32_t)rt * mistep, 0) : 0u;
                        }
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] += __uint_as_float(old[q]);
                    }
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int col = (q & 3) + 8 * (q >> 2);
                        if (col + 4 * g < ncw) {
                            if (sd.c_nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)col * jstep + (uint32_t)rt * mistep, 2);
                            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), rC, voffC, (uint32_t)col * jstep + (uint32_t)rt * mistep, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < MI; rt++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[rt][r] = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < AHEAD; k++) { iq[k] = iq[k + 1]; cq[k] = cq[k + 1]; tq[k] = tq[k + 1]; }
        stage = stage + 1 == NS ? 0 : stage + 1;
        if constexpr (TAILPIPE) tstep = (info & UREC_LAST) ? 0 : tstep + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool BF16>
__global__ __launch_bounds__(256, SPARTA_UNION16_WPC) void vbs_union_h16_kernel(const UnionParams p) {
    static_assert(kUnion16Lds * SPARTA_UNION16_WPC <= 160 * 1024, "LDS per CU");
    __shared__ __attribute__((aligned(1024))) char lds[kUnion16Lds];
    const int b = (int)blockIdx.x;
    if (b < p.side[1].n_workers) union16_body<2, BF16>(p, p.side[1], b, lds);
    __syncthreads();
    if (b < p.side[0].n_workers) union16_body<1, BF16>(p, p.side[0], b, lds);
}

__global__ __launch_bounds__(256, 3) void vbs_union_f32_kernel(const UnionParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[kUnionLds];
    static_assert(kUnionLds * 3 <= 160 * 1024, "three workgroups per CU");
    // every workgroup walks ITS tiles of each type, tallest first (vbs_union.cpp balances the workers over all types together); between two bodies every wave must be
    // done with the stages of the first
    const int b = (int)blockIdx.x;
    if (b < p.side[3].n_workers) union_body<4, kUnionStages>(p, p.side[3], b, lds);
    __syncthreads();
    if (b < p.side[2].n_workers) union_body<3, kUnionStages>(p, p.side[2], b, lds);
    __syncthreads();
    if (b < p.side[1].n_workers) union_body<2, kUnionStages>(p, p.side[1], b, lds);
    __syncthreads();
    if (b < p.side[0].n_workers) union_body<1, kUnionStages>(p, p.side[0], b, lds);
}

}  // namespace

namespace sparta_dev {

void launch_union_f32(unsigned n_slabs, hipStream_t st, const UnionParams& p) {
    int W = 0;
    for (int ty = 0; ty < kUnionTypes; ty++) W = p.side[ty].n_workers > W ? p.side[ty].n_workers : W;
    hipLaunchKernelGGL(vbs_union_f32_kernel, dim3((unsigned)W, n_slabs), dim3(256), 0, st, p);
}
// the same for a 16-bit handle: UnionParams::B is the ROW-major 16-bit copy of B (ld a multiple of 8 elements), UnionSide::A the 16-bit slices
void launch_union_h16(bool bf16, unsigned n_slabs, hipStream_t st, const UnionParams& p) {
    const dim3 grid((unsigned)(p.side[0].n_workers > p.side[1].n_workers ? p.side[0].n_workers : p.side[1].n_workers), n_slabs);
    if (bf16) hipLaunchKernelGGL(vbs_union_h16_kernel<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(vbs_union_h16_kernel<false>, grid, dim3(256), 0, st, p);
}

}  // namespace sparta_dev
