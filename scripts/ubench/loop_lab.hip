// Lab for the stream kernel's inner loop: which ingredient keeps two co-resident waves from filling the fp32 MFMA pipe?
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef RANDOM_DATA
#define RANDOM_DATA 1
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1.f; } } while (0)

// FLAGS: 1 = LDS fragment reads, 2 = LDS writes, 4 = barrier per step, 8 = pin order with sched_barrier
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void k(float* out, int steps) {
    constexpr int STAGE = 128 * 36 + 32 * 64;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 31, g = lane >> 5;
    for (int i = tid; i < 2 * STAGE; i += 256) { unsigned h = (unsigned)(i * 2654435761u + blockIdx.x * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; lds[i] = ((float)(h & 0xffffff) / 16777216.0f - 0.5f) * (RANDOM_DATA ? 1.0f : 0.0f) + (RANDOM_DATA ? 0.0f : 0.01f); }
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    f32x4 w0 = {1.f, 2.f, 3.f, 4.f};
    const float* aF = lds + 128 * 36 + 4 * g * 64 + lm;
    const float* bF = lds + (32 * wave + lm) * 36 + 4 * g;
    float* wB = lds + ((tid >> 3) * 36 + (tid & 7) * 4);
    float* wA = lds + 128 * 36 + (tid >> 4) * 64 + (tid & 15) * 4;
    for (int s = 0; s < steps; s++) {
        const int cur = (s & 1) * STAGE, nxt = ((s + 1) & 1) * STAGE;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float a0[4] = {1.f, 1.f, 1.f, 1.f}, a1[4] = {2.f, 2.f, 2.f, 2.f};
            f32x4 b = w0;
            if (FLAGS & 1) {
#pragma unroll
                for (int m = 0; m < 4; m++) { a0[m] = aF[cur + (8 * r + m) * 64]; a1[m] = aF[cur + (8 * r + m) * 64 + 32]; }
                b = *reinterpret_cast<const f32x4*>(bF + cur + 8 * r);
            }
            if (FLAGS & 2) {
                if (r < 2) { *reinterpret_cast<f32x4*>(wB + nxt + (64 * r) * 36) = w0; *reinterpret_cast<f32x4*>(wB + nxt + (64 * r + 32) * 36) = w0; }
                if (r == 2) { *reinterpret_cast<f32x4*>(wA + nxt) = w0; *reinterpret_cast<f32x4*>(wA + nxt + 16 * 64) = w0; }
            }
            if (FLAGS & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[m], a0[m], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[m], a1[m], acc1, 0, 0, 0);
            }
            if (FLAGS & 8) __builtin_amdgcn_sched_barrier(0);
        }
        if (FLAGS & 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    out[blockIdx.x * 256 + tid] = acc0[0] + acc1[3] + lds[tid];
}

template <int FLAGS>
float run(int blocks, int steps) {
    float* d; CK(hipMalloc(&d, blocks * 256 * sizeof(float)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<FLAGS>, dim3(blocks), dim3(256), 0, 0, d, steps);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<FLAGS>, dim3(blocks), dim3(256), 0, 0, d, steps);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipFree(d));
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
    }
    return 0;
}
