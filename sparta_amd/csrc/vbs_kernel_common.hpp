// vbs_kernel_common.hpp -- device-side helpers shared by the kernel translation units (clock probe, developer timeline
// stamps, accumulator types, step-record field access, the accumulator-image -> C store).
#pragma once
#include "vbs_device.hpp"

namespace sparta_dev {

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

// Cache policy of the stream kernels' C stores, chosen per launch (StreamParams::c_nt, a scalar branch around the 16 store
// instructions -- stores define no register, so the join costs nothing): NON-TEMPORAL where tiles are long (>= 6 steps per tile on average) --
// C is written once and never read back, with the default policy the 32 MB of C of the flagship push panels of B out of the L2s (fp32
// flagship, same box, interleaved runs: 54.2 -> 53.2 us) -- and the DEFAULT policy where tiles are short: a step's loads wait (vmcnt is one
// in-order counter on gfx9) for the stores of the tile that ended two steps earlier, and non-temporal stores take longer to complete
// (banded 200k, 1.8 steps per tile: 78-79 us non-temporal, 56-68 default; with every tile storing to the same rows of C: 37).

// f(integral_constant<0>) ... f(integral_constant<N - 1>): a loop whose index is a compile-time constant in every iteration
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

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

__device__ __forceinline__ int32_t sk_field(int vrec0, int vrec1, int s, int f) {   // field f of step s's record (see the kernels)
    const int ln = ((s & 7) << 3) + f;
    const int32_t x0 = __builtin_amdgcn_readlane(vrec0, ln), x1 = __builtin_amdgcn_readlane(vrec1, ln);
    return ((s >> 3) & 1) ? x1 : x0;
}

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

// ---- the C ring of the no-barrier kernels (CSTAGE: column-major C, tiles of <= 32 rows of arbitrary height) -------------------------------
// A wave does not store a finished tile straight away but parks it in a wave-private ring of 64 rows x its 32 columns in LDS (row of the ring
// = row of C & 63) and stores ALIGNED blocks of 32 rows once the tiles that follow -- a worker walks vertically adjacent tiles -- have
// completed them: whole 128-byte pieces of a column instead of the 28-row pieces of two neighbouring tiles, which the memory side has to
// read, merge and write back and acknowledges late (gfx9 has one in-order counter for loads and stores: the loads of step i + 4 wait for the
// stores of the tile that ended at step i; banded 200k).  The flush is lazy -- at the next tile's epilogue, one site in a
// scalar loop; the block a worker's range starts or ends in, and blocks around a gap between tiles (an empty block-row, sparse rows, a split
// tile), are stored partially, as before.  All state is wave-uniform (scalar registers).
constexpr int kCRingStride = 65;                                 // floats per column of the ring (64 rows + 1)
constexpr int kCRingFloats = 32 * kCRingStride;                  // per wave
struct CRing {
    float* ring;                                                 // element (column j, row r) at j * kCRingStride + (r & 63)
    int32_t win_base = 0, pend_lo = 0, pend_hi = 0;              // rows [pend_lo, pend_hi) of C are parked; win_base = first aligned block not yet stored

    // stores rows [max(b, lo), min(b + 32, hi)) of the aligned block b (lane lm = row b + lm), the way the direct epilogue stores a tile
    __device__ __forceinline__ void flush_block(const StreamParams& p, int n0, int lm, int g, uint32_t voffC, int32_t b, int32_t lo, int32_t hi) {
        float* cbase = p.C + (int64_t)b + (int64_t)n0 * p.ldc;
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, 0x7ffffff0, 0x00020000);
        const uint32_t jstep = (uint32_t)p.ldc * 4u;
        const int32_t row = b + lm;
        if (row >= lo && row < hi) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = ring[((q & 3) + 8 * (q >> 2) + 4 * g) * kCRingStride + (row & 63)];
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
    // stores what can be stored: everything parked (`all`: the next tile is not adjacent, or the range ends), else the blocks that are complete
    __device__ __forceinline__ void flush(const StreamParams& p, int n0, int lm, int g, uint32_t voffC, bool all) {
        while (pend_lo < pend_hi && (all || pend_hi - win_base >= 32)) {
            flush_block(p, n0, lm, g, voffC, win_base, pend_lo, pend_hi);
            win_base += 32;
            pend_lo = win_base < pend_hi ? win_base : pend_hi;
        }
    }
    // a finished tile (rows [c_row, c_row + mt) of C, mt <= 32, accumulator image `acc` of this wave's 32 columns) goes into the ring
    __device__ __forceinline__ void park(const StreamParams& p, int n0, int lm, int g, uint32_t voffC, const f32x16& acc, int32_t c_row, int32_t mt) {
        flush(p, n0, lm, g, voffC, pend_lo < pend_hi && c_row != pend_hi);          // (a gap: drain first, the ring restarts at this tile)
        if (pend_lo >= pend_hi) { win_base = c_row & ~31; pend_lo = c_row; }
        if (lm < mt) {
#pragma unroll
            for (int q = 0; q < 16; q++) ring[((q & 3) + 8 * (q >> 2) + 4 * g) * kCRingStride + ((c_row + lm) & 63)] = acc[q];
        }
        pend_hi = c_row + mt;
    }
};

}  // namespace sparta_dev
