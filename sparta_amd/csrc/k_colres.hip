// k_colres.hip -- the resident-column product: a SMALL, very sparse A (every block-row on the sparse-row path, at most 40 960 rows and columns)
// times a WIDE column-major B -- the reference's real matrices at its operand widths (8-22 k rows, B_COLs = 1024 / 8192:
// /root/reference/src/scripts/run_multiplication_experiments_fixed_cluster.sh:6-7).  Part of the device side of libsparta_amd.so; see vbs_device.hpp
// (ColresParams) and DESIGN.md section 14.
//
// The row gather of k_sparse.hip moves one N-wide row of B per NONZERO (d x |B| through the caches for d nonzeros per row, plus 2 x |B| for the transpose of
// the reference's column-major B, plus partial rows and a reduction for the long rows: three launches).  Here the roles are swapped: a workgroup owns NC
// COLUMNS of B / C, copies them whole into LDS (a column of the reference's layout is contiguous: no transpose), and streams A -- a few hundred KB, resident
// in every L2 -- past them: lane = row, one (column, value) entry and one LDS read of NC values per nonzero.  B is read from HBM once, C written once,
// both coalesced; what is re-read is A, N / NC times, from L2.  One launch, no scratch.
//
// A's layout (built at create time, vbs_capi.cpp: build_colres): "slots" = rows, the long ones cut into chunks of at most Lmax nonzeros (a row of 10^4 nonzeros on one lane
// would be the whole kernel's critical path), sorted by length and packed 64 to a slice of a multiple of 4 steps.  The slices of wave w (w, w + 16, ... of the sorted list: equally long
// streams; 16 waves per workgroup) lie back to back, in BATCHES of 4 steps: per batch and lane four 16-bit columns (8 bytes; 512 per wave) and, unless every value is 1.0f -- the reference's
// experiments read their matrices pattern-only, `-P 1`, run_multiplication_experiments_fixed_cluster.sh:27 -- four fp32 values (16 bytes).  The stream of A is bound by what the L2s
// deliver to 256 CUs at once (~15 TB/s: measured 12 us per workgroup with 8-byte (column, value) entries whatever the arithmetic around them), hence the narrow entries.
// A slot accumulates its nonzeros in ascending column order -- the order of the reference's CSR::multiply (/root/reference/src/general/csr.cpp:49-65) and, zeros of the blocks
// aside, of VBR::multiply -- into registers; every slot then writes its sum to its `dest` in a staging image that REPLACES the columns of B in LDS (first chunk of a row: the
// row of C; later chunks: extra cells behind the rows), the owner of a long row adds its extra cells in chunk order (fixed order: bit-reproducible), and the staging image goes
// to C in whole lines.
#include "vbs_kernel_common.hpp"

using namespace sparta_dev;

namespace {

typedef float cr_f2 __attribute__((ext_vector_type(2)));
typedef float cr_f4 __attribute__((ext_vector_type(4)));
typedef unsigned cr_u2 __attribute__((ext_vector_type(2)));

// NC values per column index c.  NC = 3 keeps a pair plane and a single plane (a 12-byte LDS read needs 16-byte alignment and takes 8 cycles: MI355X_MICROARCH.md, LDS)
template <int NC> __device__ __forceinline__ void cr_put(float* lds, int span, int c, const float (&v)[NC]) {
    if constexpr (NC == 1) lds[c] = v[0];
    else if constexpr (NC == 2) *reinterpret_cast<cr_f2*>(lds + 2 * c) = cr_f2{v[0], v[1]};
    else if constexpr (NC == 3) { *reinterpret_cast<cr_f2*>(lds + 2 * c) = cr_f2{v[0], v[1]}; lds[2 * span + c] = v[2]; }
    else *reinterpret_cast<cr_f4*>(lds + 4 * c) = cr_f4{v[0], v[1], v[2], v[3]};
}
template <int NC> __device__ __forceinline__ void cr_get(const float* lds, int span, int c, float (&v)[NC]) {
    if constexpr (NC == 1) v[0] = lds[c];
    else if constexpr (NC == 2) { const cr_f2 t = *reinterpret_cast<const cr_f2*>(lds + 2 * c); v[0] = t.x; v[1] = t.y; }
    else if constexpr (NC == 3) { const cr_f2 t = *reinterpret_cast<const cr_f2*>(lds + 2 * c); v[0] = t.x; v[1] = t.y; v[2] = lds[2 * span + c]; }
    else { const cr_f4 t = *reinterpret_cast<const cr_f4*>(lds + 4 * c); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
}
// rows c .. c + 3 of the NC columns (x[jj] = four consecutive rows of column jj; c a multiple of 4) -> LDS in whole 16-byte writes
template <int NC> __device__ __forceinline__ void cr_put4(float* lds, int span, int c, const cr_f4 (&x)[NC]) {
    if constexpr (NC == 1) *reinterpret_cast<cr_f4*>(lds + c) = x[0];
    else if constexpr (NC == 2 || NC == 3) {
        *reinterpret_cast<cr_f4*>(lds + 2 * c) = cr_f4{x[0].x, x[1].x, x[0].y, x[1].y};
        *reinterpret_cast<cr_f4*>(lds + 2 * c + 4) = cr_f4{x[0].z, x[1].z, x[0].w, x[1].w};
        if constexpr (NC == 3) *reinterpret_cast<cr_f4*>(lds + 2 * span + c) = x[2];
    } else {
        *reinterpret_cast<cr_f4*>(lds + 4 * c) = cr_f4{x[0].x, x[1].x, x[2].x, x[3].x};
        *reinterpret_cast<cr_f4*>(lds + 4 * c + 4) = cr_f4{x[0].y, x[1].y, x[2].y, x[3].y};
        *reinterpret_cast<cr_f4*>(lds + 4 * c + 8) = cr_f4{x[0].z, x[1].z, x[2].z, x[3].z};
        *reinterpret_cast<cr_f4*>(lds + 4 * c + 12) = cr_f4{x[0].w, x[1].w, x[2].w, x[3].w};
    }
}

// 1024 threads, one workgroup per CU.  (512 threads with two or three workgroups sharing a CU -- one's loads of B / stores of C under another's stream of A -- was built and
// measured: the phases' times still ADD UP, start offsets between the co-resident workgroups or not, and the shorter workgroups hide less latency: bcsstk18 at N = 8192
// 0.32-0.37 ms against 0.26, wiki-Vote 0.245 against 0.217; profiles/r4/lab_colres_stagger.txt.)
constexpr int kCrWaves = kColresWaves, kCrThreads = 64 * kCrWaves;

// SL: slices a wave may own (slice s of the length-sorted list belongs to wave s % 16; the sums of all of them stay in registers until the columns of B are no longer needed)
// UNIT: every stored value is 1.0f (no value array: the sums are sums of elements of B)
template <int NC, int SL, bool UNIT>
__global__ __launch_bounds__(kCrThreads) void colres_kernel(const ColresParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = (int)blockIdx.x * NC;
    const int nc = p.N - j0 < NC ? p.N - j0 : NC;                 // columns of this workgroup that exist (the last workgroup of a ragged N)
    // the part of the rows of C this workgroup owns (grid y); its tables
    const ColresPartDev* pdp = p.parts + blockIdx.y;
    const int part_r0 = __builtin_amdgcn_readfirstlane(pdp->r0), part_rows = __builtin_amdgcn_readfirstlane(pdp->rows);
    const int n_slices = __builtin_amdgcn_readfirstlane(pdp->n_slices), n_long = __builtin_amdgcn_readfirstlane(pdp->n_long), P = __builtin_amdgcn_readfirstlane(pdp->plane);
    const bool all_store = __builtin_amdgcn_readfirstlane(pdp->all_store) != 0;
    const int32_t* wslice = p.meta + __builtin_amdgcn_readfirstlane(pdp->meta);
    const int32_t* woff = wslice + 17;
    const int32_t* bnd_all = woff + p.n_ranges * 17;
    const int32_t* dest = p.dest + __builtin_amdgcn_readfirstlane(pdp->dest);
    const ColresLong* longs = p.longs + __builtin_amdgcn_readfirstlane(pdp->longs);

    // Every CU starts its first workgroup at the same time, and all of them then load B together, stream A together, store C together: HBM idles while the LDS works and the
    // other way round.  The CUs of the first dispatch round are therefore started in p.share groups, a share of a workgroup's time apart (developer knob SPARTA_COLRES_STAGGER_US).
    {
        const int wg = (int)(blockIdx.x + blockIdx.y * gridDim.x);
        if (p.stagger_ticks > 0 && wg < p.n_cus) {
            const int slot = (wg >> 3) % p.share;                                      // (consecutive workgroups go to the 8 XCDs in turn: >> 3 = the CU within its XCD)
            if (slot > 0) {
                const long long t_end = wall_clock64() + (long long)slot * p.stagger_ticks;      // 100 MHz
                while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(8);
            }
        }
    }

    // the sums of the slices of this wave (slice s of the part's length-sorted list belongs to wave s % 16): in registers from the first K range to the last
    float acc[SL][NC];
#pragma unroll
    for (int i = 0; i < SL; i++) {
#pragma unroll
        for (int jj = 0; jj < NC; jj++) acc[i][jj] = 0.0f;
    }
    const int sl0 = __builtin_amdgcn_readfirstlane(wslice[wave]);
    const int n_w = __builtin_amdgcn_readfirstlane(wslice[wave + 1]) - sl0;       // slices of this wave (<= SL: checked by the host)

    int dst[SL];                                                    // where this wave's sums go in the end: requested now, needed behind the stream of A
#pragma unroll
    for (int I = 0; I < SL; I++) dst[I] = dest[(sl0 + (I < n_w ? I : 0)) * 64 + lane];

    for (int rg = 0; rg < p.n_ranges; rg++) {
        const int k0 = p.krange[rg], klen = p.krange[rg + 1] - k0;               // this range's rows of B
        const int span = (klen + 4) & ~3;                             // cells per column of B: the rows + the zero cell the padding of A points at
        if (rg > 0) __syncthreads();                                  // the last wave is done with the previous range

        // ---- 1. rows k0 .. k0 + klen of the NC columns of B -> LDS, interleaved per row of B ---------------------------------------------------------------
        // (branch-free: a lane past the last row reads the last row(s) and stores nothing; a column past the last one repeats the last one and is never written to C)
        if (!(p.probe & 1)) {
            const float* Bj = p.B + (int64_t)j0 * p.ldb + k0;
            if (p.vec_in) {                                           // 16-byte aligned columns: four rows per lane and load, whole 16-byte LDS writes, every load of a round in flight at once
                constexpr int R = 3;                                  // 4096-row rounds in flight (12 288 rows: what NC >= 3 can hold)
                const int full = klen >> 2;                           // whole groups of four rows
                for (int q0 = tid; q0 < full; q0 += R * kCrThreads) {
                    cr_f4 x[R][NC];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int q = q0 + r * kCrThreads, qq = q < full ? q : full - 1;
#pragma unroll
                        for (int jj = 0; jj < NC; jj++) x[r][jj] = *reinterpret_cast<const cr_f4*>(Bj + (int64_t)(jj < nc ? jj : nc - 1) * p.ldb + 4 * qq);
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int q = q0 + r * kCrThreads;
                        if (q < full) cr_put4<NC>(lds, span, 4 * q, x[r]);
                    }
                }
                for (int c = 4 * full + tid; c < klen; c += kCrThreads) {              // the last klen % 4 rows
                    float v[NC];
#pragma unroll
                    for (int jj = 0; jj < NC; jj++) v[jj] = Bj[(int64_t)(jj < nc ? jj : nc - 1) * p.ldb + c];
                    cr_put<NC>(lds, span, c, v);
                }
            } else {
                constexpr int U = 4;
                for (int c0 = tid; c0 < klen; c0 += U * kCrThreads) {
                    float v[U][NC];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int c = c0 + u * kCrThreads, cc = c < klen ? c : klen - 1;
#pragma unroll
                        for (int jj = 0; jj < NC; jj++) v[u][jj] = Bj[(int64_t)(jj < nc ? jj : nc - 1) * p.ldb + cc];
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int c = c0 + u * kCrThreads;
                        if (c < klen) cr_put<NC>(lds, span, c, v[u]);
                    }
                }
            }
        }
        if (tid == 0) {                                               // the cell behind the last row: what a padding entry of A (column = klen) reads
            const float z[NC] = {};
            cr_put<NC>(lds, span, klen, z);
        }

        // ---- 2. A streams past: lane = slot, a batch (4 steps) of the wave's stream is one 512-byte line of columns (+ 1 KB of values) -------------------------
        // ONE stream per wave and range, read D batches ahead whatever slice they belong to (a loop per slice exposed a scalar load, a global load and an LDS read, one after
        // the other, at every one of the 12 slices of a wave).  Where a slice ends (wave-uniform: a scalar compare per batch) the running sums are parked in the registers of
        // that slice -- they cannot go to LDS before the last wave is done with the columns of B -- and the next slice's are taken up (zeros in the first range).
        const int off = __builtin_amdgcn_readfirstlane(woff[rg * 17 + wave]);        // in batches
        const int T = (p.probe & 2) ? 0 : __builtin_amdgcn_readfirstlane(woff[rg * 17 + wave + 1]) - off;
        const cr_u2* ec = reinterpret_cast<const cr_u2*>(p.col4) + (size_t)off * 64 + lane;
        const cr_f4* ev = UNIT ? nullptr : reinterpret_cast<const cr_f4*>(p.val4) + (size_t)off * 64 + lane;
        const int32_t* bnd = bnd_all + rg * n_slices + sl0;        // bnd[i]: the batch behind the last one of the wave's i-th slice
        // all of the wave's boundaries in ONE load (lane i holds bnd[i]; SL <= 64): read at every slice end they were a dependent round trip each -- a global load and the wait for
        // it, 3 to 12 times per wave and range -- in the middle of the stream
        static_assert(SL <= 64, "a lane per slice of the wave");
        const int bnd_v = lane < n_w ? bnd[lane] : -1;
        int i = 0;
        int nb = T > 0 ? __builtin_amdgcn_readlane(bnd_v, 0) : -1;
        float cur[NC];
#pragma unroll
        for (int jj = 0; jj < NC; jj++) cur[jj] = acc[0][jj];
        constexpr int D = 4;                                       // batches in flight (an L2 hit is ~700 cycles away): the loop body is written D times, each copy consuming the
        cr_u2 rc[D];                                               // register buffer it then refills for D batches later
        cr_f4 rv[D];
#pragma unroll
        for (int q = 0; q < D; q++) {
            const int t = q < T ? q : (T > 0 ? T - 1 : 0);         // (past the end: the last line again, never used)
            rc[q] = ec[(size_t)t * 64];
            if constexpr (!UNIT) rv[q] = ev[(size_t)t * 64];
            else rv[q] = cr_f4{1.0f, 1.0f, 1.0f, 1.0f};
            __builtin_amdgcn_sched_barrier(0);                     // issue order = consumption order (vmcnt counts loads in order).  NOT an asm memory clobber: behind one the
        }                                                          // slice boundaries are no longer scalar loads, and their vector loads drain the ring at every slice end
        __syncthreads();                                           // the columns of B are in LDS (the first batches of A were requested in front of this wait)
        auto batch = [&](cr_u2& bc, cr_f4& bv, int t0) __attribute__((always_inline)) {
            const unsigned w0 = bc.x, w1 = bc.y;
            const int c[4] = {(int)(w0 & 0xffffu), (int)(w0 >> 16), (int)(w1 & 0xffffu), (int)(w1 >> 16)};
            float b[4][NC];
#pragma unroll
            for (int u = 0; u < 4; u++) cr_get<NC>(lds, span, c[u], b[u]);
            const float a[4] = {bv.x, bv.y, bv.z, bv.w};
            __builtin_amdgcn_sched_barrier(0);
            {
                const int t = t0 + D < T ? t0 + D : T - 1;         // refilled behind its last use: no copies
                bc = ec[(size_t)t * 64];
                if constexpr (!UNIT) bv = ev[(size_t)t * 64];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int jj = 0; jj < NC; jj++) cur[jj] = UNIT ? cur[jj] + b[u][jj] : __builtin_fmaf(a[u], b[u][jj], cur[jj]);
            }
            if (t0 + 1 == nb) {                                    // (wave-uniform) the slice ends with this batch: park its sums in ITS registers, take up the next slice's --
                i++;                                               // scalar branches and 2 NC moves (plain conditional stores: the chain becomes ONE indexed store and the
#pragma unroll                                                     // array moves to scratch)
                for (int I = 0; I < SL; I++) {
                    if (i == I + 1) {
#pragma unroll
                        for (int jj = 0; jj < NC; jj++) asm volatile("v_mov_b32 %0, %1" : "=v"(acc[I][jj]) : "v"(cur[jj]));
                        if (I + 1 < SL) {
#pragma unroll
                            for (int jj = 0; jj < NC; jj++) asm volatile("v_mov_b32 %0, %1" : "=v"(cur[jj]) : "v"(acc[I + 1 < SL ? I + 1 : I][jj]));
                        }
                    }
                }
                nb = i < n_w ? __builtin_amdgcn_readlane(bnd_v, i) : -1;
            }
        };
        for (int t0 = 0; t0 < T; t0 += D) {
            static_for<0, D>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                if (t0 + q < T) batch(rc[q], rv[q], t0 + q);
            });
        }
    }
    __syncthreads();                                              // nobody reads the columns of B any more

    // ---- 3. sums -> staging image (NC planes of P cells), long rows add their extra cells in chunk order ---------------------------------------------
    {
#pragma unroll
        for (int I = 0; I < SL; I++) {
            if (I < n_w && dst[I] >= 0) {
#pragma unroll
                for (int jj = 0; jj < NC; jj++) lds[jj * P + dst[I]] = acc[I][jj];
            }
        }
    }
    __syncthreads();
    if (n_long > 0) {
        for (int t = tid; t < n_long * NC; t += kCrThreads) {
            const int jj = t / n_long, q = t - jj * n_long;
            const ColresLong lr = longs[q];
            float* cell = lds + jj * P;
            float sum = cell[lr.row];
            const float* x = cell + lr.first;
            int i = 0;
            for (; i + 8 <= lr.n; i += 8) {
                float t8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t8[u] = x[i + u];
#pragma unroll
                for (int u = 0; u < 8; u++) sum += t8[u];
            }
            for (; i < lr.n; i++) sum += x[i];
            cell[lr.row] = sum;
        }
        __syncthreads();
    }

    // ---- 4. staging image -> the part's rows of the NC columns of C, whole lines ----------------------------------------------------------------------
    float* Cj = p.C + (int64_t)j0 * p.ldc + part_r0;
    if (p.probe & 4) return;
    const uint8_t* mode = p.mode + part_r0;                       // per row: 0 not ours (a block-row of tiles: left alone), 1 store, 2 add to what the tile launches stored
    if (p.vec_out) {
        const int r4 = part_rows >> 2;
        // the modes of this thread's groups of four rows, all requested before the first is needed (a load -> wait -> store chain per group cost 3-5 us per workgroup); a part
        // whose rows are ALL stored (no tiles in the handle) reads none
        constexpr int MQ = kColresCellsPerPlane / (4 * kCrThreads);            // groups per thread at most
        unsigned m4v[MQ];
#pragma unroll
        for (int k = 0; k < MQ; k++) {
            const int q = tid + k * kCrThreads;
            m4v[k] = all_store ? 0x01010101u : (q < r4 ? *reinterpret_cast<const unsigned*>(mode + 4 * q) : 0u);
        }
        for (int jj = 0; jj < nc; jj++) {
            float* Cc = Cj + (int64_t)jj * p.ldc;
            const float* src = lds + jj * P;
#pragma unroll
            for (int k = 0; k < MQ; k++) {
                const int q = tid + k * kCrThreads;
                const unsigned m4 = m4v[k];
                if (q >= r4 || m4 == 0u) continue;
                cr_f4 x = *reinterpret_cast<const cr_f4*>(src + 4 * q);
                if (m4 == 0x01010101u && !p.accumulate) { *reinterpret_cast<cr_f4*>(Cc + 4 * q) = x; continue; }
                const cr_f4 c = *reinterpret_cast<const cr_f4*>(Cc + 4 * q);
                const bool acc_all = p.accumulate != 0;
                x.x = (m4 & 0xffu) == 0u ? c.x : ((m4 & 0xffu) == 2u || acc_all ? c.x + x.x : x.x);
                x.y = ((m4 >> 8) & 0xffu) == 0u ? c.y : (((m4 >> 8) & 0xffu) == 2u || acc_all ? c.y + x.y : x.y);
                x.z = ((m4 >> 16) & 0xffu) == 0u ? c.z : (((m4 >> 16) & 0xffu) == 2u || acc_all ? c.z + x.z : x.z);
                x.w = (m4 >> 24) == 0u ? c.w : ((m4 >> 24) == 2u || acc_all ? c.w + x.w : x.w);
                *reinterpret_cast<cr_f4*>(Cc + 4 * q) = x;
            }
            for (int r = 4 * r4 + tid; r < part_rows; r += kCrThreads) {
                const unsigned m = all_store ? 1u : mode[r];
                if (m != 0u) Cc[r] = (p.accumulate || m == 2u) ? Cc[r] + src[r] : src[r];
            }
        }
    } else {
        for (int jj = 0; jj < nc; jj++) {
            float* Cc = Cj + (int64_t)jj * p.ldc;
            const float* src = lds + jj * P;
            for (int r = tid; r < part_rows; r += kCrThreads) {
                const unsigned m = all_store ? 1u : mode[r];
                if (m != 0u) Cc[r] = (p.accumulate || m == 2u) ? Cc[r] + src[r] : src[r];
            }
        }
    }
}

template <int NC, int SL, bool UNIT>
int colres_launch(const ColresParams& p, size_t lds_bytes, hipStream_t st) {
    static int attr_rc[64];                                       // once per instantiation and device: allow more than 64 KB of dynamic LDS (0: not yet, 1: done)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (attr_rc[dev] != 1) {
        const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&colres_kernel<NC, SL, UNIT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (rc != 0) return rc;
        attr_rc[dev] = 1;
    }
    const unsigned grid = (unsigned)((p.N + NC - 1) / NC);
    hipLaunchKernelGGL((colres_kernel<NC, SL, UNIT>), dim3(grid, (unsigned)p.n_parts), dim3(kCrThreads), lds_bytes, st, p);
    return 0;
}

}  // namespace

namespace sparta_dev {

int colres_max_slices(int nc) { return (40 / nc) * kCrWaves; }

int launch_colres(int nc, const ColresParams& p, size_t lds_bytes, hipStream_t st) {
    const bool unit = p.val4 == nullptr;
    switch (nc) {
        case 1: return unit ? colres_launch<1, 40, true>(p, lds_bytes, st) : colres_launch<1, 40, false>(p, lds_bytes, st);
        case 2: return unit ? colres_launch<2, 20, true>(p, lds_bytes, st) : colres_launch<2, 20, false>(p, lds_bytes, st);
        case 3: return unit ? colres_launch<3, 13, true>(p, lds_bytes, st) : colres_launch<3, 13, false>(p, lds_bytes, st);
        case 4: return unit ? colres_launch<4, 10, true>(p, lds_bytes, st) : colres_launch<4, 10, false>(p, lds_bytes, st);
    }
    return -1;
}

}  // namespace sparta_dev
