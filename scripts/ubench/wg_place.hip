// ubench (round 5): where do the workgroups of a persistent launch land?  grid = 3 x CUs workgroups of 256 threads with 48 KB of LDS each (three per CU, as vbs_union_f32_kernel);
// every workgroup records its XCC id and the CU it runs on (HW_REG_HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13).  Prints how many of the trios (b, b + CUs, b + 2 CUs) share a CU.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/wg_place.hip -o /tmp/wg_place && /tmp/wg_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 3) void where(unsigned* out, int spin) {
    __shared__ char lds[49152];
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);          // stay resident until every workgroup has started
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = (xcc & 0xf) | ((unsigned)lds[5] << 16); }
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, W = 3 * cus;
    unsigned* d; hipMalloc(&d, W * 8);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(where, dim3(W), dim3(256), 0, 0, d, 20000);       // 200 us at 100 MHz
        std::vector<unsigned> h(2 * W); hipMemcpy(h.data(), d, W * 8, hipMemcpyDeviceToHost);
        auto cu_of = [&](int b) { const unsigned hw = h[2 * b], x = h[2 * b + 1] & 0xf; return (int)((x << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)); };
        std::map<int, int> per_cu;
        for (int b = 0; b < W; b++) per_cu[cu_of(b)]++;
        int same3 = 0, same2 = 0;
        for (int b = 0; b < cus; b++) { const int a = cu_of(b), c = cu_of(b + cus), e = cu_of(b + 2 * cus); same3 += (a == c && c == e); same2 += (a == c) + (c == e) + (a == e); }
        int mn = 1 << 30, mx = 0; for (auto& kv : per_cu) { mn = kv.second < mn ? kv.second : mn; mx = kv.second > mx ? kv.second : mx; }
        printf("launch %d: %d CUs seen, workgroups per CU min %d max %d; trios (b, b + %d, b + %d) on one CU: %d of %d (pairs %d of %d)\n", rep, (int)per_cu.size(), mn, mx, cus, 2 * cus, same3, cus, same2, 3 * cus);
        if (rep == 0) { printf("first 24 workgroups (xcc:se.sh.cu):"); for (int b = 0; b < 24; b++) { const unsigned hw = h[2 * b]; printf(" %u:%u.%u.%u", h[2 * b + 1] & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf); } printf("\n");
            printf("workgroups 256..279:"); for (int b = 256; b < 280; b++) { const unsigned hw = h[2 * b]; printf(" %u:%u.%u.%u", h[2 * b + 1] & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf); } printf("\n"); }
    }
    return 0;
}
