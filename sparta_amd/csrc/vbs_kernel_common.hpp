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

}  // namespace sparta_dev
