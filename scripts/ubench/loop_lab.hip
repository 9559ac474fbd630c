// Lab for the stream kernel's inner loop: which ingredient keeps two co-resident waves from filling the fp32 MFMA pipe?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef RANDOM_DATA
#define RANDOM_DATA 1
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1.f; } } while (0)

// FLAGS: 1 = LDS fragment reads, 2 = LDS writes, 4 = barrier per step, 8 = pin order with sched_barrier, 16 = global loads,
// 32 = record/cursor control emulation, 64 = not-taken uniform branches, 128 / 256 = 1 / 2 more TAKEN branches per round
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void k(float* out, int steps, const f32x4* __restrict__ gsrc, long gmask, long long* clk) {
    long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
    constexpr int STAGE = 128 * 36 + 32 * 64;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 31, g = lane >> 5;
    for (int i = tid; i < 2 * STAGE; i += 256) { unsigned h = (unsigned)(i * 2654435761u + blockIdx.x * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; lds[i] = ((float)(h & 0xffffff) / 16777216.0f - 0.5f) * (RANDOM_DATA ? 1.0f : 0.0f) + (RANDOM_DATA ? 0.0f : 0.01f); }
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    f32x4 w0 = {1.f, 2.f, 3.f, 4.f};
    f32x4 ga[6], gb[6];
    long gpos = ((long)blockIdx.x * 9973 + tid) & gmask;
    for (int q = 0; q < 6; q++) { ga[q] = w0; gb[q] = w0; }
    const float* aF = lds + 128 * 36 + 4 * g * 64 + lm;
    const float* bF = lds + (32 * wave + lm) * 36 + 4 * g;
    float* wB = lds + ((tid >> 3) * 36 + (tid & 7) * 4);
    float* wA = lds + 128 * 36 + (tid >> 4) * 64 + (tid & 15) * 4;
    int vrec0 = tid * 7 + 3, vrec1 = tid * 5 + 1;   // pretend record batches
    long cursor = 0; int hsc = 64;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(gsrc), 0, 0x7ffffff0, 0x00020000);
    const unsigned smask = (unsigned)(((gmask + 1) * 16 - 1) & 0x1ffff000);      // stay inside the allocation (<= 512 MB), 4 KB granules
    unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(blockIdx.x * 9973u * 4096u)) & smask;
    auto field = [&](int st, int f) -> int { const int ln = ((st & 7) << 3) + f; const int x0 = __builtin_amdgcn_readlane(vrec0, ln), x1 = __builtin_amdgcn_readlane(vrec1, ln); return ((st >> 3) & 1) ? x1 : x0; };
    auto step = [&](int s, f32x4 (&src)[6]) {
        int flags = 64;
        if (FLAGS & 32) { flags = field(s, 5); if ((flags & 0xffff) > 100000) w0.x += 1.f; }
        const int cur = (s & 1) * STAGE, nxt = ((s + 1) & 1) * STAGE;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float a0[4] = {1.f, 1.f, 1.f, 1.f}, a1[4] = {2.f, 2.f, 2.f, 2.f};
            f32x4 b = w0;
            if (FLAGS & 1) {
#pragma unroll
                for (int m = 0; m < 4; m++) { a0[m] = aF[cur + (8 * r + m) * 64]; if (!(FLAGS & 8192)) a1[m] = aF[cur + (8 * r + m) * 64 + 32]; }
                b = *reinterpret_cast<const f32x4*>(bF + cur + 8 * r);
            }
            if (FLAGS & 2) {
                const bool g = (FLAGS & 16) != 0;
                if (r < 2) { *reinterpret_cast<f32x4*>(wB + nxt + (64 * r) * 36) = g ? src[2 * r] : w0; *reinterpret_cast<f32x4*>(wB + nxt + (64 * r + 32) * 36) = g ? src[2 * r + 1] : w0; }
                if (r == 2) { *reinterpret_cast<f32x4*>(wA + nxt) = g ? src[4] : w0; *reinterpret_cast<f32x4*>(wA + nxt + 16 * 64) = g ? src[5] : w0; }
            }
            if ((FLAGS & 32) && r == 3) {
                const int f3 = field(s + 3, 5);
                if (f3 & (1 << 16)) { cursor = (long)(unsigned)field(s + 3, 0) | ((long)field(s + 3, 1) << 32); hsc = field(s + 3, 3); } else cursor += 32L * hsc;
                gpos = (gpos + (cursor & 1) + (field(s + 3, 2) & 1)) & gmask;
            }
            auto ld = [&](int q) {
                if (FLAGS & 2048) {                               // buffer load, scalar offset: no VALU address math (as in the real kernel)
                    src[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)tid * 16u, soff, 0));
                    soff = (soff + 4096u * 37u) & smask;
                } else { src[q] = gsrc[gpos]; gpos = (gpos + 256 * 37) & gmask; }
            };
            if ((FLAGS & 16) && !(FLAGS & 512) && r == 3) {
#pragma unroll
                for (int q = 0; q < 6; q++) ld(q);   // refilled two steps ahead of its next use
            }
            if ((FLAGS & 16) && (FLAGS & 512) && r >= 1) {                                               // SPREAD: the pair written in round r-1 is refilled in round r
#pragma unroll
                for (int q = 2 * (r - 1); q < 2 * r; q++) ld(q);
            }
            if (FLAGS & 64) { if (gmask == 77 + r) { acc0[r] += 1.f; } if ((gmask & (1L << (40 + r))) != 0) { out[r] = acc1[r]; } }
            if (FLAGS & 128) asm volatile("s_branch .Ltb%=\n\t.rept 16\n\ts_nop 0\n\t.endr\n.Ltb%=:" ::: "memory");                  // one TAKEN branch per round
            if (FLAGS & 256) { asm volatile("s_branch .Ltc%=\n\t.rept 16\n\ts_nop 0\n\t.endr\n.Ltc%=:" ::: "memory"); asm volatile("s_branch .Ltd%=\n\t.rept 16\n\ts_nop 0\n\t.endr\n.Ltd%=:" ::: "memory"); }
            if (FLAGS & 8) __builtin_amdgcn_sched_barrier(0);
            if (FLAGS & 16384) {                                  // no MFMA at all: what the staging path alone can move
                acc0[r] += b[0] + a0[0];
            } else if (FLAGS & 8192) {                            // MI1 emulation: one accumulator, 16 MFMAs per step
#pragma unroll
                for (int m = 0; m < 4; m++) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[m], a0[m], acc0, 0, 0, 0);
            } else if (FLAGS & 4096) {                                   // 4 dependent MFMAs on acc0, then 4 on acc1 (behind a uniform branch)
#pragma unroll
                for (int m = 0; m < 4; m++) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[m], a0[m], acc0, 0, 0, 0);
                if (gmask != 12345) {
#pragma unroll
                    for (int m = 0; m < 4; m++) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[m], a1[m], acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[m], a0[m], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[m], a1[m], acc1, 0, 0, 0);
            }
            }
            if (FLAGS & 8) __builtin_amdgcn_sched_barrier(0);
            if ((FLAGS & 1024) && r >= 1) {                     // one load behind each of the first two MFMA pairs of the round
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        if ((FLAGS & 32) && (flags & (1 << 17))) { out[tid] = acc0[1]; acc0 = (f32x16){0}; acc1 = (f32x16){0}; }
        if (FLAGS & 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int s = 0; s < steps; s += 2) { step(s, ga); step(s + 1, gb); }
    out[blockIdx.x * 256 + tid] = acc0[0] + acc1[3] + lds[tid];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t0; clk[1] = __builtin_readcyclecounter(); clk[2] = r0; clk[3] = wall_clock64(); }
}

__global__ void fill_random(float* p, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) { unsigned h = (unsigned)(i * 2654435761u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; p[i] = (float)(h & 0xffffff) / 16777216.0f - 0.5f; }
}
static long long* g_clk = nullptr;   // [0..1] s_memtime start/end, [2..3] s_memrealtime start/end of block 0
static double g_mhz = 0;
template <int FLAGS>
float run(int blocks, int steps, long gmask_override = 0) {
    float* d; CK(hipMalloc(&d, blocks * 256 * sizeof(float)));
    static f32x4* g = nullptr; const long gelems = 1L << 25;   // 512 MB
    if (!g) { CK(hipMalloc(&g, gelems * sizeof(f32x4))); if (getenv("LAB_ZERO")) { CK(hipMemset(g, 0, gelems * sizeof(f32x4))); } else { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float*)g, gelems * 4); } CK(hipMalloc(&g_clk, 4 * sizeof(long long))); }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<FLAGS>, dim3(blocks), dim3(256), 0, 0, d, steps, g, gmask_override ? gmask_override : gelems - 1, g_clk);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<FLAGS>, dim3(blocks), dim3(256), 0, 0, d, steps, g, gmask_override ? gmask_override : gelems - 1, g_clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipFree(d));
    long long c[4]; CK(hipMemcpy(c, g_clk, sizeof(c), hipMemcpyDeviceToHost));
    g_mhz = (double)(c[1] - c[0]) / (double)(c[3] - c[2]) * 100.0;       // s_memtime ticks per 100 MHz real-time tick
    printf("[%4.0f MHz-equivalent s_memtime] ", g_mhz);
    return ms;
}
int main() {
    const int steps = 4000;
    for (int blocks : {256, 512}) {
        const double ideal = (double)steps * 32 * 64 * (blocks / 256) / 2.4e9 * 1e3;
        printf("blocks=%d (%d/CU), %d steps of 32 MFMA; ideal %.3f ms @2.4GHz\n", blocks, blocks / 256, steps, ideal);
        printf("  mfma only              %.3f\n", run<0>(blocks, steps));
        printf("  + frag reads           %.3f\n", run<1>(blocks, steps));
        printf("  + frag reads pinned    %.3f\n", run<1 | 8>(blocks, steps));
        printf("  + writes               %.3f\n", run<2>(blocks, steps));
        printf("  + barrier              %.3f\n", run<4>(blocks, steps));
        printf("  reads+writes           %.3f\n", run<3>(blocks, steps));
        printf("  reads+writes+barrier   %.3f\n", run<7>(blocks, steps));
        printf("  all, pinned            %.3f\n", run<15>(blocks, steps));
        printf("  reads+writes+barrier+global loads (6 x 16B/lane/step) %.3f\n", run<7 | 16>(blocks, steps));
        printf("  reads+writes+barrier+control emulation %.3f\n", run<7 | 32>(blocks, steps));
        printf("  reads+writes+barrier+loads+control      %.3f\n", run<7 | 16 | 32>(blocks, steps));
        printf("  mfma only + 8 uniform not-taken branches/step %.3f\n", run<64>(blocks, steps));
        printf("  r+w+barrier+loads SPREAD by pairs       %.3f\n", run<7 | 16 | 512>(blocks, steps));
        printf("  r+w+barrier+loads SPREAD + group-barrier %.3f\n", run<7 | 16 | 512 | 1024>(blocks, steps));
        printf("  mfma only, 4xacc0 then branch 4xacc1    %.3f\n", run<4096>(blocks, steps));
        printf("  r+w+barrier+BUFFER loads, 4+4 order     %.3f\n", run<7 | 16 | 2048 | 4096>(blocks, steps));
        printf("  MI1 (16 MFMA/step) mfma only  [ideal = half] %.3f\n", run<8192>(blocks, steps));
        printf("  MI1 r+w+barrier                          %.3f\n", run<7 | 8192>(blocks, steps));
        printf("  MI1 r+w+barrier+BUFFER loads             %.3f\n", run<7 | 16 | 2048 | 8192>(blocks, steps));
        printf("  MI1 r+w+barrier+BUFFER loads, 1 MB set   %.3f\n", run<7 | 16 | 2048 | 8192>(blocks, steps, (1L << 16) - 1));
        printf("  MI1 r+w+barrier+BUFFER loads, 32 MB set  %.3f\n", run<7 | 16 | 2048 | 8192>(blocks, steps, (1L << 21) - 1));
        printf("  MI1 r+w+barrier+BUFFER loads, 128 MB set %.3f\n", run<7 | 16 | 2048 | 8192>(blocks, steps, (1L << 23) - 1));
        printf("  MI2 r+w+barrier+BUFFER loads, 32 MB set  %.3f\n", run<7 | 16 | 2048>(blocks, steps, (1L << 21) - 1));
        for (long set : {1L << 16, 1L << 21, 1L << 25}) {
            const float ms = run<7 | 16 | 2048 | 16384>(blocks, steps, set - 1);
            printf("  NO MFMA: r+w+barrier+BUFFER loads, %4ld MB set: %.3f ms = %.1f B/clk/CU @2.4GHz, %.2f TB/s\n", set * 16 >> 20, ms, (double)blocks * steps * 24576.0 / (ms * 1e-3) / 256 / 2.4e9, (double)blocks * steps * 24576.0 / (ms * 1e-3) / 1e12);
        }
        printf("  MI1 r+w+barrier+BUFFER loads+control+br  %.3f\n", run<7 | 16 | 32 | 64 | 2048 | 8192>(blocks, steps));
        printf("  r+w+barrier+BUFFER loads clumped        %.3f\n", run<7 | 16 | 2048>(blocks, steps));
        printf("  r+w+barrier+BUFFER loads spread         %.3f\n", run<7 | 16 | 512 | 2048>(blocks, steps));
        printf("  r+w+barrier+BUFFER loads spread+grpbar  %.3f\n", run<7 | 16 | 512 | 1024 | 2048>(blocks, steps));
        printf("  r+w+barrier+BUFFER loads clumped, L2-hot %.3f\n", run<7 | 16 | 2048>(blocks, steps, (1L << 16) - 1));
        printf("  mfma only + loads clumped               %.3f\n", run<16>(blocks, steps));
        printf("  mfma only + loads spread                %.3f\n", run<16 | 512>(blocks, steps));
        printf("  mfma only + loads spread + group-barrier %.3f\n", run<16 | 512 | 1024>(blocks, steps));
        printf("  mfma only + 4 TAKEN branches/step  %.3f\n", run<128>(blocks, steps));
        printf("  mfma only + 12 TAKEN branches/step %.3f\n", run<128 | 256>(blocks, steps));
        printf("  reads+writes+barrier + 4 taken     %.3f\n", run<7 | 128>(blocks, steps));
        printf("  reads+writes+barrier + 12 taken    %.3f\n", run<7 | 128 | 256>(blocks, steps));
        printf("  reads+writes+barrier+loads+control+branches %.3f\n", run<7 | 16 | 32 | 64>(blocks, steps));
        printf("  same, 1 MB working set (L2-hot)  %.3f\n", run<7 | 16>(blocks, steps, (1L << 16) - 1));
        printf("  same, 64 MB working set (MALL)   %.3f\n", run<7 | 16>(blocks, steps, (1L << 22) - 1));
    }
    return 0;
}
