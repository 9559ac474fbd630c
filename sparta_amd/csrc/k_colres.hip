// k_colres.hip -- the resident-column product: a SMALL, very sparse A (every block-row on the sparse-row path, at most 40 960 rows and columns)
// times a WIDE column-major B -- the reference's real matrices at its operand widths (8-22 k rows, B_COLs = 1024 / 8192:
// /root/reference/src/scripts/run_multiplication_experiments_fixed_cluster.sh:6-7).  Part of the device side of libsparta_amd.so; see vbs_device.hpp
// (ColresParams) and DESIGN.md section 14.
//
// The row gather of k_sparse.hip moves one N-wide row of B per NONZERO (d x |B| through the caches for d nonzeros per row, plus 2 x |B| for the transpose of
// the reference's column-major B, plus partial rows and a reduction for the long rows: three launches).  Here the roles are swapped: a workgroup owns NC
// COLUMNS of B / C, copies them whole into LDS (a column of the reference's layout is contiguous: no transpose), and streams A -- a few hundred KB, resident
// in every L2 -- past them: lane = row, one 8-byte (column, value) entry and one LDS read of NC values per nonzero.  B is read from HBM once, C written once,
// both coalesced; what is re-read is A, N / NC times, from L2.  One launch, no scratch.
//
// A's layout (built at create time, vbs_capi.cpp: build_colres): "slots" = rows, the long ones cut into chunks of at most Lmax nonzeros (a row of 10^4 nonzeros on one lane
// would be the whole kernel's critical path), sorted by length and packed 64 to a slice, entry k of the slice's 64 slots contiguous (sliced ELLPACK: a wave's
// load of step k is one 512-byte line).  A slot accumulates its nonzeros in ascending column order -- the order of the reference's CSR::multiply
// (/root/reference/src/general/csr.cpp:49-65) and, zeros of the blocks aside, of VBR::multiply -- into registers; every slot then writes its sum to its
// `dest` in a staging image that REPLACES the columns of B in LDS (first chunk of a row: the row of C; later chunks: extra cells behind the rows), the owner of a long row adds
// its extra cells in chunk order (fixed order: bit-reproducible), and the staging image goes to C in whole lines.
#include "vbs_kernel_common.hpp"

using namespace sparta_dev;

namespace {

typedef float cr_f2 __attribute__((ext_vector_type(2)));
typedef float cr_f4 __attribute__((ext_vector_type(4)));

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

constexpr int kCrThreads = 1024, kCrWaves = kCrThreads / 64;

// SL: slices a wave may own (slice s belongs to wave s % 16; the sums of all of them stay in registers until the columns of B are no longer needed)
template <int NC, int SL>
__global__ __launch_bounds__(kCrThreads) void colres_kernel(const ColresParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = (int)blockIdx.x * NC;
    const int nc = p.N - j0 < NC ? p.N - j0 : NC;                 // columns of this workgroup that exist (the last workgroup of a ragged N)
    const int span = (p.cols + 1) & ~1;

    // ---- 1. the NC columns of B -> LDS, interleaved per row of B -------------------------------------------------------------------------------
    {
        // (branch-free: a lane past the last row reads the last row and stores nothing; a column past the last one repeats the last one and is never written to C)
        const float* Bj = p.B + (int64_t)j0 * p.ldb;
        constexpr int U = 4;
        for (int c0 = tid; c0 < p.cols; c0 += U * kCrThreads) {
            float v[U][NC];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int c = c0 + u * kCrThreads, cc = c < p.cols ? c : p.cols - 1;
#pragma unroll
                for (int jj = 0; jj < NC; jj++) v[u][jj] = Bj[(int64_t)(jj < nc ? jj : nc - 1) * p.ldb + cc];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int c = c0 + u * kCrThreads;
                if (c < p.cols) cr_put<NC>(lds, span, c, v[u]);
            }
        }
    }
    __syncthreads();

    // ---- 2. A streams past: lane = slot, entry k of the slice's 64 slots is one 512-byte line ------------------------------------------------------
    float acc[SL][NC];
#pragma unroll
    for (int i = 0; i < SL; i++) {
#pragma unroll
        for (int jj = 0; jj < NC; jj++) acc[i][jj] = 0.0f;
    }
    constexpr int U = NC >= 3 ? 4 : 8;
    static_for<0, SL>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int s = wave + i * kCrWaves;
        if (s < p.n_slices) {
            const int off = __builtin_amdgcn_readfirstlane(p.soff[s]);
            const int wd = __builtin_amdgcn_readfirstlane(p.soff[s + 1] - off) >> 6;       // entries per slot of this slice
            const int2* e = p.ent + off + lane;
            // (a slot shorter than its slice is padded with (a column of its own row, 0.0f): no per-lane condition; the steps past the slice's end re-read its last
            // entry with the value replaced by zero -- one scalar-conditioned select per entry instead of a tail loop)
            for (int k0 = 0; k0 < wd; k0 += U) {
                int2 cv[U];
#pragma unroll
                for (int u = 0; u < U; u++) { const int k = k0 + u < wd ? k0 + u : wd - 1; cv[u] = e[k * 64]; }
                float b[U][NC];
#pragma unroll
                for (int u = 0; u < U; u++) cr_get<NC>(lds, span, cv[u].x, b[u]);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const float a = k0 + u < wd ? __builtin_bit_cast(float, cv[u].y) : 0.0f;
#pragma unroll
                    for (int jj = 0; jj < NC; jj++) acc[i][jj] = __builtin_fmaf(a, b[u][jj], acc[i][jj]);
                }
            }
        }
    });
    __syncthreads();                                              // nobody reads the columns of B any more

    // ---- 3. sums -> staging image (NC planes of P cells), long rows add their extra cells in chunk order ---------------------------------------------
    const int P = p.plane;
    static_for<0, SL>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int s = wave + i * kCrWaves;
        if (s < p.n_slices) {
            const int d = p.dest[s * 64 + lane];
            if (d >= 0) {
#pragma unroll
                for (int jj = 0; jj < NC; jj++) lds[jj * P + d] = acc[i][jj];
            }
        }
    });
    __syncthreads();
    if (p.n_long > 0) {
        for (int t = tid; t < p.n_long * NC; t += kCrThreads) {
            const int jj = t / p.n_long, q = t - jj * p.n_long;
            const ColresLong lr = p.longs[q];
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

    // ---- 4. staging image -> the NC columns of C, whole lines --------------------------------------------------------------------------------------
    float* Cj = p.C + (int64_t)j0 * p.ldc;
    if (p.vec_out) {
        const int r4 = p.rows >> 2;
        for (int jj = 0; jj < nc; jj++) {
            float* Cc = Cj + (int64_t)jj * p.ldc;
            const float* src = lds + jj * P;
            for (int q = tid; q < r4; q += kCrThreads) {
                cr_f4 x = *reinterpret_cast<const cr_f4*>(src + 4 * q);
                if (p.accumulate) x += *reinterpret_cast<const cr_f4*>(Cc + 4 * q);
                *reinterpret_cast<cr_f4*>(Cc + 4 * q) = x;
            }
            for (int r = 4 * r4 + tid; r < p.rows; r += kCrThreads) Cc[r] = p.accumulate ? Cc[r] + src[r] : src[r];
        }
    } else {
        for (int jj = 0; jj < nc; jj++) {
            float* Cc = Cj + (int64_t)jj * p.ldc;
            const float* src = lds + jj * P;
            for (int r = tid; r < p.rows; r += kCrThreads) Cc[r] = p.accumulate ? Cc[r] + src[r] : src[r];
        }
    }
}

template <int NC, int SL>
int colres_launch(const ColresParams& p, size_t lds_bytes, hipStream_t st) {
    static int attr_rc = -1;                                      // once per instantiation: allow more than 64 KB of dynamic LDS
    if (attr_rc != 0) {
        attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&colres_kernel<NC, SL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr_rc != 0) return attr_rc;
    }
    const unsigned grid = (unsigned)((p.N + NC - 1) / NC);
    hipLaunchKernelGGL((colres_kernel<NC, SL>), dim3(grid), dim3(kCrThreads), lds_bytes, st, p);
    return 0;
}

}  // namespace

namespace sparta_dev {

int colres_max_slices(int nc) { return (nc == 1 ? 40 : nc == 2 ? 20 : nc == 3 ? 14 : 10) * kCrWaves; }

int launch_colres(int nc, const ColresParams& p, size_t lds_bytes, hipStream_t st) {
    switch (nc) {
        case 1: return colres_launch<1, 40>(p, lds_bytes, st);
        case 2: return colres_launch<2, 20>(p, lds_bytes, st);
        case 3: return colres_launch<3, 14>(p, lds_bytes, st);
        case 4: return colres_launch<4, 10>(p, lds_bytes, st);
    }
    return -1;
}

}  // namespace sparta_dev
