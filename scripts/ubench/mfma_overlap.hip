// Micro-benchmark: does v_mfma_f32_32x32x2_f32 overlap with LDS / VALU / VMEM issue of the SAME wave and of a co-resident wave?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: mfma only, 1: lds reads only, 2: interleaved mfma+ds_read_b128, 3: interleaved mfma + ds_write_b128, 4: mfma + valu, 5: mfma + 2 ds_read_b32
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) lds[i] = (float)(i & 7);
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
    float s = 0.f;
    const float* rp = lds + (tid & 63) * 4;
    float* wp = lds + 4096 + tid * 4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE != 1) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, v.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, v.w, acc1, 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) {
                f32x4 t = *reinterpret_cast<const f32x4*>(rp + ((u * 64) & 1023));
                s += t.x;
            }
            if (MODE == 3) { *reinterpret_cast<f32x4*>(wp) = v; }
            if (MODE == 4) {
#pragma unroll
                for (int q = 0; q < 8; q++) s = s * 1.0001f + (float)q;
            }
            if (MODE == 5) { s += rp[(u * 8) & 255] + rp[256 + ((u * 8) & 255)]; }
        }
    }
    out[blockIdx.x * 256 + tid] = acc0[0] + acc1[3] + s + lds[4096 + tid];
}

template <int MODE>
float run(int blocks, int iters) {
    float* d; hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    return ms;
}
int main() {
    const int iters = 2000;
    for (int blocks : {256, 512}) {
        printf("blocks=%d (%d per CU): 32 mfma per unrolled body x %d iters; ideal mfma-only = %.3f ms at 2.4 GHz\n", blocks, blocks / 256, iters,
               (double)iters * 32 * 64 * (blocks / 256) / 2.4e9 * 1e3);
        printf("  mfma only            %.3f ms\n", run<0>(blocks, iters));
        printf("  ds_read_b128 only    %.3f ms\n", run<1>(blocks, iters));
        printf("  mfma + ds_read_b128  %.3f ms\n", run<2>(blocks, iters));
        printf("  mfma + ds_write_b128 %.3f ms\n", run<3>(blocks, iters));
        printf("  mfma + 8 valu fma    %.3f ms\n", run<4>(blocks, iters));
        printf("  mfma + 2 ds_read_b32 %.3f ms\n", run<5>(blocks, iters));
    }
    return 0;
}
