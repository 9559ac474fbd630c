// k_sparse.hip -- the HBM-bound leg of the path: sparse-row kernels for nearly empty block-rows ("wavefront-level partial sums
// for thin / ragged blocks"), the layout transposes they need for the reference's column-major B / C, and the row-block pack of
// the multi-GPU exchange.  Part of the device side of libsparta_amd.so; see vbs_device.hpp and DESIGN.md section 3.2 (4).
#include "vbs_kernel_common.hpp"

using namespace sparta_dev;

namespace {

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

// lane t < SP_BATCH takes nonzero q + t of [q, p1) (column 0 / value 0 beyond the end: a valid row of B, never used)
__device__ __forceinline__ void sp_batch_load(const SparseParams& p, int64_t q, int64_t p1, int lane, int& cl, float& vl) {
    cl = 0; vl = 0.0f;
    if (lane < SP_BATCH && q + lane < p1) { cl = p.col[q + lane]; vl = p.val[q + lane]; }
}

// (cl, vl) = the first batch of the row, already requested by the caller (sp_batch_load at p0): the (column, value) pairs of batch
// i + 1 are requested BEFORE the rows of B of batch i, so a batch costs one memory latency, not two in a row
template <int VEC, int BK>
__device__ __forceinline__ typename SpVec<VEC>::T sparse_row_partial(const SparseParams& p, int64_t p0, int64_t p1, int n0, int lane, int cl, float vl) {
    typedef typename SpVec<VEC>::T V;
    typedef typename std::conditional<BK == 0, float, unsigned short>::type E;
    typedef typename std::conditional<BK == 0, V, typename SpRaw16<VEC>::T>::type L;
    V acc = (V)(0.0f);
    const bool in = VEC > 1 || n0 < p.N;
    const E* Bl = (const E*)p.B + (in ? n0 : 0);
    for (int64_t q = p0; q < p1; q += SP_BATCH) {
        const int n = (int)(p1 - q < SP_BATCH ? p1 - q : SP_BATCH);            // wave-uniform
        int cn;
        float vn;
        sp_batch_load(p, q + SP_BATCH, p1, lane, cn, vn);
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
                const int c = __builtin_amdgcn_readlane(cl, t);  // lanes >= n hold column 0: a valid row, never used.  (Issuing only the quarters of a
                                                                 // batch that hold nonzeros -- wave-uniform branches around groups of 4 loads -- measured 18 % SLOWER:
                                                                 // the waits behind the joins lose the overlap inside the batch.)
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
        cl = cn; vl = vn;
    }
    return acc;
}

// Round 4: the same sum for a ROW-major B with the bookkeeping moved off the vector pipe.  Measured on the hub parts of the power-law configs the gather above spent 10-12
// vector instructions per nonzero on top of its load -- two v_readlane, a 64-bit vector add for the address, a VECTOR compare + exec mask + branch for "t < n" (n was
// wave-uniform but lived in VGPRs), four conversions and two packed FMAs -- 24 cycles per nonzero and CU against the 8 the load path needs for 512 bytes.  Here the
// (column, value) pairs come through SCALAR loads (constant address space: the range is wave-uniform, so a batch is two s_load_dwordx16 and no readlane), the row of B
// is addressed as scalar base + one per-lane byte offset (global_load ... v_off, s[base:base+1]), a full batch runs without any condition and only the last, short
// batch of a range takes wave-uniform scalar branches.  The order of the additions is the one of sparse_row_partial: bit-identical results.
__device__ __forceinline__ int64_t sp_uniform64(int64_t x) {
    return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x));
}

template <int VEC, int BK>
__device__ __forceinline__ typename SpVec<VEC>::T sparse_row_partial_s(const SparseParams& p, int64_t p0v, int64_t p1v, int n0) {
    typedef typename SpVec<VEC>::T V;
    typedef typename std::conditional<BK == 0, float, unsigned short>::type E;
    typedef typename std::conditional<BK == 0, V, typename SpRaw16<VEC>::T>::type L;
    typedef const __attribute__((address_space(4))) int32_t* ccol_t;
    typedef const __attribute__((address_space(4))) float* cval_t;
    const int64_t p0 = sp_uniform64(p0v), p1 = sp_uniform64(p1v);
    const ccol_t colc = (ccol_t)p.col;
    const cval_t valc = (cval_t)p.val;
    V acc = (V)(0.0f);
    const bool in = VEC > 1 || n0 < p.N;
    const uint32_t loff = (uint32_t)(in ? n0 : 0) * (uint32_t)sizeof(E);       // this lane's columns inside a row of B
    const char* Bb = (const char*)p.B;
    const uint32_t ldbb32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p.ldb * (int64_t)sizeof(E)));
    // row c of B as a SCALAR base (kept in SGPRs: the empty asm stops the compiler from folding the per-lane offset into a 64-bit vector address per nonzero)
    typedef const __attribute__((address_space(1))) char* gptr_t;
    typedef const __attribute__((address_space(1))) L* gl_t;
    auto row_of = [&](int col) __attribute__((always_inline)) -> gptr_t {
        uint64_t r = (uint64_t)Bb + (uint64_t)(uint32_t)col * (uint64_t)ldbb32;     // (a row of B is shorter than 4 GB: launch_sparse_rows checks)
        asm("" : "+s"(r));
        return (gptr_t)r;
    };
    auto fma_row = [&](float v, const L& b) __attribute__((always_inline)) {
        if constexpr (BK == 0) {
            if constexpr (VEC == 1) acc = __builtin_fmaf(v, b, acc);
            else {
#pragma unroll
                for (int e = 0; e < VEC; e++) acc[e] = __builtin_fmaf(v, b[e], acc[e]);
            }
        } else if constexpr (BK == 1 && VEC > 1) {
            // fp16: v_fma_mix_f32 widens the half inside the FMA (one instruction per element instead of a conversion and half a packed FMA); the same fused product-sum
            const auto w = __builtin_bit_cast(typename std::conditional<VEC == 2, uint32_t, uint32_t __attribute__((ext_vector_type(2)))>::type, b);
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                uint32_t word;
                if constexpr (VEC == 2) word = w; else word = w[e >> 1];
                float a = acc[e];
                if (e & 1) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(a) : "s"(v), "v"(word));
                else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(a) : "s"(v), "v"(word));
                acc[e] = a;
            }
        } else if constexpr (VEC == 1) acc = __builtin_fmaf(v, sp_widen(b, BK == 2), acc);
        else {
#pragma unroll
            for (int e = 0; e < VEC; e++) acc[e] = __builtin_fmaf(v, sp_widen(b[e], BK == 2), acc[e]);
        }
    };
    int c[SP_BATCH];
    float v[SP_BATCH];
#pragma unroll
    for (int t = 0; t < SP_BATCH; t++) { c[t] = colc[p0 + t]; v[t] = valc[p0 + t]; }          // (the arrays end in 64 zero entries: vbs_capi.cpp)
#pragma unroll 1
    for (int64_t q = p0; q < p1; q += SP_BATCH) {
        const int n = (int)(p1 - q < SP_BATCH ? p1 - q : SP_BATCH);
        L b[SP_BATCH];
        if (n == SP_BATCH) {
#pragma unroll
            for (int t = 0; t < SP_BATCH; t++) b[t] = *(gl_t)(row_of(c[t]) + loff);
            float vc[SP_BATCH];
#pragma unroll
            for (int t = 0; t < SP_BATCH; t++) vc[t] = v[t];
#pragma unroll
            for (int t = 0; t < SP_BATCH; t++) { c[t] = colc[q + SP_BATCH + t]; v[t] = valc[q + SP_BATCH + t]; }      // the next batch's pairs, behind this batch's loads
#pragma unroll
            for (int t = 0; t < SP_BATCH; t++) fma_row(vc[t], b[t]);
        } else {
#pragma unroll
            for (int t = 0; t < SP_BATCH; t++)
                if (t < n) b[t] = *(gl_t)(row_of(c[t]) + loff);
#pragma unroll
            for (int t = 0; t < SP_BATCH; t++)
                if (t < n) fma_row(v[t], b[t]);
        }
    }
    return acc;
}

// the gather of one range of nonzeros: the scalar-batch version for a row-major B, the in-place one for a column-major B
template <int VEC, int BK>
__device__ __forceinline__ typename SpVec<VEC>::T sparse_range(const SparseParams& p, int64_t p0, int64_t p1, int n0, int lane) {
    if (p.scalar_gather && !(VEC == 1 && p.b_col_stride > 0)) return sparse_row_partial_s<VEC, BK>(p, p0, p1, n0);
    int cl;
    float vl;
    sp_batch_load(p, p0, p1, lane, cl, vl);
    return sparse_row_partial<VEC, BK>(p, p0, p1, n0, lane, cl, vl);
}

template <int VEC>
__device__ __forceinline__ void sparse_row_store(const SparseParams& p, int ord, typename SpVec<VEC>::T acc, int n0) {
    typedef typename SpVec<VEC>::T V;
    if (VEC == 1 && n0 >= p.N) return;
    const int32_t cr = p.crow[ord];                              // bit 31: the row also has MFMA tiles -- add to what they stored
    float* o = p.out + (int64_t)(cr & 0x7fffffff) * p.ldo + n0;
    if (p.accumulate || cr < 0) acc += *reinterpret_cast<const V*>(o);
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
    sparse_row_store<VEC>(p, ord, sparse_range<VEC, BK>(p, p0, p1, n0, lane), n0);
}

// kCmRows partial rows of one chunk of columns meet in LDS (element (row j, column l * VEC + e) at ((e * kCmRows + j) * 65 + l)) and leave
// as pieces of kCmRows consecutive rows of one column of the column-major C: thread (j, cg) stores row j of columns cg, cg + CG, ...
template <int VEC, int kCmRows, int NT = kThreads>
__device__ __forceinline__ void sp_cm_flush(const SparseParams& p, const float* tile, bool valid, int32_t crow) {
    constexpr int W = 64 * VEC;
    constexpr int CG = NT / kCmRows;                             // columns written at a time (one per group of kCmRows threads)
    constexpr int NQ = W / CG < 8 ? W / CG : 8;                  // stores of a thread in flight
    static_assert(W % (CG * NQ) == 0, "chunk width must be a whole number of store rounds");
    const int j = threadIdx.x & (kCmRows - 1), cg = threadIdx.x / kCmRows;
    if (!valid) return;
    const int cbase = blockIdx.y * W;
    const bool add = p.accumulate || crow < 0;                   // bit 31 of a crow entry: the row also has MFMA tiles -- add to what they stored
    float* o = p.out + (crow & 0x7fffffff) + (int64_t)cbase * p.ldo;
#pragma unroll 1
    for (int c0 = 0; c0 < W; c0 += NQ * CG) {
        float x[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int c = c0 + cg + CG * q;
            x[q] = tile[((c % VEC) * kCmRows + j) * 65 + c / VEC];
        }
        if (add) {
            float old[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int c = c0 + cg + CG * q;
                old[q] = o[(int64_t)(cbase + c < p.N ? c : 0) * p.ldo];
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) x[q] += old[q];
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int c = c0 + cg + CG * q;
            if (VEC > 1 || cbase + c < p.N) o[(int64_t)c * p.ldo] = x[q];
        }
    }
}

// the same for a COLUMN-major C (the reference's layout): a workgroup takes kCmRows consecutive rows of the list (mostly consecutive rows
// of C), each wave walks kCmRows / 4 of them one after the other, the partial rows meet in LDS and leave as pieces of kCmRows consecutive
// rows of one column (kCmRows * 4 contiguous bytes where the C rows are consecutive).  Replaces "row-major scratch + scatter launch": the product
// is written once instead of written, read back and written again.
template <int VEC, int BK, int kCmRows, int NW>
__device__ __forceinline__ void sparse_rows_cm_body(const SparseParams& p, float* tile, int bx) {      // tile: VEC * kCmRows * 65 floats of LDS
    typedef typename SpVec<VEC>::T V;
    constexpr int RPW = kCmRows / NW;                            // rows a wave walks one after the other
    static_assert(RPW >= 1 && RPW * NW == kCmRows, "rows per workgroup must be a multiple of the waves");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot0 = bx * kCmRows;
    const int n0 = (blockIdx.y * 64 + lane) * VEC;
    // the wave's rows: lane i holds the nonzero range of row j = wave + NW i (one chain of dependent loads for all of them, not one per row)
    int mlo = 0, mhi = 0, mcnt = 0;
    if (lane < RPW && slot0 + wave + NW * lane < p.n_list) {
        const int ord = p.list[slot0 + wave + NW * lane];
        const int64_t a = p.rowptr[ord];
        mlo = (int)(uint32_t)a; mhi = (int)(a >> 32); mcnt = (int)(p.rowptr[ord + 1] - a);
    }
    auto row_p0 = [&](int i) {
        return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(mhi, i) << 32) | (uint32_t)__builtin_amdgcn_readlane(mlo, i));
    };
    const bool scalar_path = p.scalar_gather && !(VEC == 1 && p.b_col_stride > 0);
    int cl = 0;
    float vl = 0.0f;
    if (!scalar_path) {
        const int64_t a = row_p0(0);
        sp_batch_load(p, a, a + __builtin_amdgcn_readlane(mcnt, 0), lane, cl, vl);
    }
#pragma unroll 1
    for (int i = 0; i < RPW; i++) {
        const int j = wave + NW * i;
        const int64_t p0 = row_p0(i), p1 = p0 + __builtin_amdgcn_readlane(mcnt, i);
        V acc;
        if (scalar_path) acc = sparse_row_partial_s<VEC, BK>(p, p0, p1, n0);         // an empty range (slot past the list) gives zeros
        else {
            int cn = 0;
            float vn = 0.0f;
            if (i + 1 < RPW) {                                   // first batch of the next row, requested before this row's rows of B
                const int64_t a = row_p0(i + 1);
                sp_batch_load(p, a, a + __builtin_amdgcn_readlane(mcnt, i + 1), lane, cn, vn);
            }
            acc = sparse_row_partial<VEC, BK>(p, p0, p1, n0, lane, cl, vl);
            cl = cn; vl = vn;
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            float x;
            if constexpr (VEC == 1) x = acc; else x = acc[e];
            tile[(e * kCmRows + j) * 65 + lane] = x;
        }
    }
    __syncthreads();
    const int fj = threadIdx.x & (kCmRows - 1);
    const bool valid = slot0 + fj < p.n_list;
    sp_cm_flush<VEC, kCmRows, 64 * NW>(p, tile, valid, valid ? p.crow[p.list[slot0 + fj]] : 0);
}

template <int VEC, int BK, int kCmRows, int NW>
__global__ __launch_bounds__(64 * NW) void sparse_rows_cm_kernel(SparseParams p) {
    __shared__ float tile[VEC * kCmRows * 65];                   // 64 * VEC columns of kCmRows rows, layout in sp_cm_flush
    sparse_rows_cm_body<VEC, BK, kCmRows, NW>(p, tile, blockIdx.x);
}

// long rows (hubs): cut into segments of <= kSpSeg nonzeros, one wave per segment writes a partial row; a second launch adds
// the partial rows of every long row in segment order (deterministic) and stores the row

template <int VEC, int BK>
__device__ __forceinline__ void sparse_segments_body(const SparseParams& p, const SpSegRec* segs, int32_t n_segs, float* part, int bx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = bx * 4 + wave;
    if (slot >= n_segs) return;
    const SpSegRec sg = segs[slot];
    const int n0 = (blockIdx.y * 64 + lane) * VEC;
    typename SpVec<VEC>::T acc = sparse_range<VEC, BK>(p, sg.p0, sg.p0 + sg.cnt, n0, lane);
    if (VEC == 1 && n0 >= p.N) return;
    *reinterpret_cast<typename SpVec<VEC>::T*>(part + (int64_t)sg.pad * p.N + n0) = acc;
}

template <int VEC, int BK>
__global__ __launch_bounds__(kThreads) void sparse_segments_kernel(SparseParams p, const SpSegRec* segs, int32_t n_segs, float* part) {
    sparse_segments_body<VEC, BK>(p, segs, n_segs, part, blockIdx.x);
}

// XCD-affine order (round 4).  With all workgroups walking the same column window, the eight L2s of the chip hold eight copies of the same few MB of B.  Here the
// segment list is eight STREAMS (vbs_capi.cpp: windows dealt to streams by nonzeros, each stream in column order), and workgroup b -- which the hardware places on XCD
// b % 8 -- takes its segments from stream b % 8: every L2 holds its own window, eight different windows are cache-resident at a time.
template <int VEC, int BK>
__global__ __launch_bounds__(kThreads) void sparse_segments_xcd_kernel(SparseParams p, const SpSegRec* segs, const int32_t* stream_begin, float* part) {
    const int stream = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int b0 = stream_begin[stream], b1 = stream_begin[stream + 1];
    if (b0 + k * 4 >= b1) return;
    sparse_segments_body<VEC, BK>(p, segs + b0, b1 - b0, part, k);
}

// small sparse parts, column-major C: the rows of ordinary length (4 or 16 per workgroup, as in the separate launch) and the segments of the long rows in ONE launch - workgroups
// [0, n_row_wgs) take rows, the rest segments (a launch costs 5-8 us whatever it does: wiki-Vote 33.7 us = rows 13.0 + segments 6.7 + reduce 10.7 + transpose 5.8)
template <int VEC, int BK, int kCmRows>
__global__ __launch_bounds__(kThreads) void sparse_small_cm_kernel(SparseParams p, int32_t n_row_wgs, const SpSegRec* segs, int32_t n_segs, float* part) {
    __shared__ float tile[VEC * kCmRows * 65];
    if ((int)blockIdx.x < n_row_wgs) sparse_rows_cm_body<VEC, BK, kCmRows, 4>(p, tile, blockIdx.x);       // (workgroup-uniform branch: the barrier inside is reached by all or none)
    else sparse_segments_body<VEC, BK>(p, segs, n_segs, part, (int)blockIdx.x - n_row_wgs);
}

// the partial rows of one long row, added in segment order - eight loads in flight at a time (the order of the additions is what fixes the bits, not the
// order of the loads: one load + one add per iteration made the reduction a chain of memory latencies, 52 us for the hub rows of ia-wikiquote)
template <int VEC>
__device__ __forceinline__ typename SpVec<VEC>::T sp_sum_partials(const float* first, int n_seg, int64_t stride) {
    typedef typename SpVec<VEC>::T V;
    V acc = (V)(0.0f);
    int sgi = 0;
    for (; sgi + 8 <= n_seg; sgi += 8) {
        V t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = *reinterpret_cast<const V*>(first + (int64_t)(sgi + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; u++) acc += t[u];
    }
    for (; sgi < n_seg; sgi++) acc += *reinterpret_cast<const V*>(first + (int64_t)sgi * stride);
    return acc;
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
    const V acc = sp_sum_partials<VEC>(part + (int64_t)r.seg_begin * p.N + n0, r.n_seg, p.N);
    sparse_row_store<VEC>(p, r.ord, acc, n0);
}

// the same for a column-major C: kCmRows long rows per workgroup through LDS (where most rows are long - dense power-law inputs - this
// kernel writes most of C)
template <int VEC, int kCmRows>
__global__ __launch_bounds__(kThreads) void sparse_reduce_cm_kernel(SparseParams p, const SpLongRec* rows, int32_t n_rows, const float* part) {
    typedef typename SpVec<VEC>::T V;
    __shared__ float tile[VEC * kCmRows * 65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot0 = blockIdx.x * kCmRows;
    const int n0 = (blockIdx.y * 64 + lane) * VEC;
    const bool in = VEC > 1 || n0 < p.N;
#pragma unroll 1
    for (int i = 0; i < kCmRows / 4; i++) {
        const int j = wave + 4 * i;
        V acc = (V)(0.0f);
        if (slot0 + j < n_rows && in) {
            const SpLongRec r = rows[slot0 + j];
            acc = sp_sum_partials<VEC>(part + (int64_t)r.seg_begin * p.N + n0, r.n_seg, p.N);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            float x;
            if constexpr (VEC == 1) x = acc; else x = acc[e];
            tile[(e * kCmRows + j) * 65 + lane] = x;
        }
    }
    __syncthreads();
    const int fj = threadIdx.x & (kCmRows - 1);
    const bool valid = slot0 + fj < n_rows;
    sp_cm_flush<VEC, kCmRows>(p, tile, valid, valid ? p.crow[rows[slot0 + fj].ord] : 0);
}

// B (column-major, ld = ldb, or the gathered slabs) -> row-major rows x N (ld = N); 64 x 64 tiles through LDS: a wave reads 64
// consecutive rows of one column (256 contiguous bytes) and writes 64 consecutive columns of one row (256 contiguous bytes)
template <class E>
__global__ __launch_bounds__(kThreads) void b_to_row_major_kernel(const E* __restrict__ B, int64_t ldb, int64_t shard_rows, int64_t shard_stride,
                                                                  int64_t rows, int N, E* __restrict__ out, int64_t ld_out) {
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
        if (rr < rows && n < N) out[rr * ld_out + n] = tile[lane][j];
    }
}

// Measured and dropped (round 2): a 16-bit variant with 8-byte loads (4 rows of a column per thread) and 16-byte stores (16 columns of a row) --
// 4 + 2 accesses per thread instead of 16 + 16 two-byte ones.  R-MAT 2^20 bf16 N = 512: 4.36 / 4.37 ms with it, 4.33 ms without; ogbn-shaped: 10.73 /
// 10.70 against 10.59 ms.  The transpose is not bound by the width of its accesses.

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

}  // namespace

namespace sparta_dev {

namespace {
template <int VEC, int BK>
void launch_sparse_t(SparseParams q, unsigned gy, hipStream_t st, const int32_t* list, int64_t n_short, const SpSegRec* segs, int64_t n_segs,
                     const SpLongRec* longs, int64_t n_long, float* part, const int32_t* stream_begin, int64_t max_stream) {
    const bool fused_small = q.out_is_c == 2 && n_short > 0 && n_short < 262144 && n_long > 0 && n_segs > 0 && n_segs < 65536;
    if (fused_small) {
        q.list = list; q.n_list = (int32_t)n_short;
        if (n_short < 16384) {
            const int32_t n_row_wgs = (int32_t)((n_short + 3) / 4);
            hipLaunchKernelGGL((sparse_small_cm_kernel<VEC, BK, 4>), dim3((unsigned)(n_row_wgs + (n_segs + 3) / 4), gy), dim3(kThreads), 0, st, q, n_row_wgs, segs, (int32_t)n_segs, part);
        } else {
            const int32_t n_row_wgs = (int32_t)((n_short + 15) / 16);
            hipLaunchKernelGGL((sparse_small_cm_kernel<VEC, BK, 16>), dim3((unsigned)(n_row_wgs + (n_segs + 3) / 4), gy), dim3(kThreads), 0, st, q, n_row_wgs, segs, (int32_t)n_segs, part);
        }
    } else if (n_short > 0) {
        q.list = list; q.n_list = (int32_t)n_short;
        // column-major C: 16 rows per workgroup (64-byte pieces of a column; 32 rows = 128-byte pieces but half the workgroups: measured 0-25 %
        // slower from R-MAT 2^14 to 2^20), 4 waves walking 4 rows each (8 / 16 waves with 2 / 1 rows each: 15-45 % slower on R-MAT 2^20; 1 wave
        // with 16 rows: 10-30 % slower)
        // ... on LARGE sparse parts.  A small one (about as many rows as the GPU has wave slots: the smaller of the reference's real inputs, 10^4 rows; at 5 x 10^4 rows the 16-row form is ahead again) is a
        // latency chain, not a bandwidth problem: one row per wave, 4 rows per workgroup, so that no wave walks four rows one after the other
        if (q.out_is_c == 2 && n_short < 16384) hipLaunchKernelGGL((sparse_rows_cm_kernel<VEC, BK, 4, 4>), dim3((unsigned)((n_short + 3) / 4), gy), dim3(256), 0, st, q);
        else if (q.out_is_c == 2) hipLaunchKernelGGL((sparse_rows_cm_kernel<VEC, BK, 16, 4>), dim3((unsigned)((n_short + 15) / 16), gy), dim3(256), 0, st, q);
        else hipLaunchKernelGGL((sparse_rows_kernel<VEC, BK>), dim3((unsigned)((n_short + 3) / 4), gy), dim3(kThreads), 0, st, q);
    }
    if (n_long > 0) {
        if (!fused_small && stream_begin)
            hipLaunchKernelGGL((sparse_segments_xcd_kernel<VEC, BK>), dim3((unsigned)(8 * ((max_stream + 3) / 4)), gy), dim3(kThreads), 0, st, q, segs, stream_begin, part);
        else if (!fused_small) hipLaunchKernelGGL((sparse_segments_kernel<VEC, BK>), dim3((unsigned)((n_segs + 3) / 4), gy), dim3(kThreads), 0, st, q, segs, (int32_t)n_segs, part);
        // (a few hundred long rows -- the hyper-sparse real inputs -- are a latency chain like their short rows above: one row per wave instead of four one after the other;
        // social_location, 453 long rows: 9-14 us for this launch with 16 rows per workgroup)
        if (q.out_is_c == 2 && n_long < 16384)
            hipLaunchKernelGGL((sparse_reduce_cm_kernel<VEC, 4>), dim3((unsigned)((n_long + 3) / 4), gy), dim3(kThreads), 0, st, q, longs, (int32_t)n_long, (const float*)part);
        else if (q.out_is_c == 2) hipLaunchKernelGGL((sparse_reduce_cm_kernel<VEC, 16>), dim3((unsigned)((n_long + 15) / 16), gy), dim3(kThreads), 0, st, q, longs, (int32_t)n_long, (const float*)part);
        else hipLaunchKernelGGL(sparse_reduce_kernel<VEC>, dim3((unsigned)((n_long + 3) / 4), gy), dim3(kThreads), 0, st, q, longs, (int32_t)n_long, (const float*)part);
    }
}
}  // namespace

void launch_sparse_kernels(int vec, int bk, const SparseParams& q, unsigned gy, hipStream_t st, const int32_t* list, int64_t n_short, const SpSegRec* segs,
                           int64_t n_segs, const SpLongRec* longs, int64_t n_long, float* part, const int32_t* stream_begin, int64_t max_stream) {
#define SPARTA_SP_DISPATCH(V_)                                                                                         \
    do {                                                                                                               \
        if (bk == 0) launch_sparse_t<V_, 0>(q, gy, st, list, n_short, segs, n_segs, longs, n_long, part, stream_begin, max_stream);              \
        else if (bk == 1) launch_sparse_t<V_, 1>(q, gy, st, list, n_short, segs, n_segs, longs, n_long, part, stream_begin, max_stream);         \
        else launch_sparse_t<V_, 2>(q, gy, st, list, n_short, segs, n_segs, longs, n_long, part, stream_begin, max_stream);                      \
    } while (0)
    if (vec == 4) SPARTA_SP_DISPATCH(4); else if (vec == 2) SPARTA_SP_DISPATCH(2); else SPARTA_SP_DISPATCH(1);
#undef SPARTA_SP_DISPATCH
}

void launch_b_to_row_major(bool is16, unsigned grid, hipStream_t st, const void* B, int64_t ldb, int64_t shard_rows, int64_t shard_stride, int64_t rows, int N,
                           void* out, int64_t ld_out) {
    if (!is16) hipLaunchKernelGGL(b_to_row_major_kernel<float>, dim3(grid), dim3(kThreads), 0, st, (const float*)B, ldb, shard_rows, shard_stride, rows, N, (float*)out, ld_out);
    else hipLaunchKernelGGL(b_to_row_major_kernel<unsigned short>, dim3(grid), dim3(kThreads), 0, st, (const unsigned short*)B, ldb, shard_rows, shard_stride, rows, N, (unsigned short*)out, ld_out);
}

void launch_pack_blocks(dim3 grid, hipStream_t st, const void* src, const int32_t* ids, void* dst, int64_t block_vec) {
    hipLaunchKernelGGL(pack_blocks_kernel, grid, dim3(kThreads), 0, st, (const u32x4*)src, ids, (u32x4*)dst, block_vec);
}

}  // namespace sparta_dev
