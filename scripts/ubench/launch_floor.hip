// ubench (round 5): what does a launch cost whatever the kernel does?  Back-to-back launches on one stream of a kernel that touches nothing, by workgroup size and dynamic LDS:
// microseconds per launch (HIP events around 2000 launches).   hipcc --offload-arch=gfx950 -O2 scripts/ubench/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void nothing(int* p) { extern __shared__ char lds[]; if (p && threadIdx.x == 4096) { lds[0] = 1; p[0] = lds[0]; } }
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&nothing), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfg[][3] = {{256, 64, 0}, {256, 256, 0}, {256, 1024, 0}, {256, 1024, 100 * 1024}, {256, 1024, 150 * 1024}, {768, 256, 49152}, {2048, 256, 0}, {4096, 1024, 150 * 1024}};
    for (auto& c : cfg) {
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(nothing, dim3(c[0]), dim3(c[1]), c[2], 0, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(nothing, dim3(c[0]), dim3(c[1]), c[2], 0, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d x %4d threads, %6d B of LDS: %.2f us per launch\n", c[0], c[1], c[2], ms / 2000 * 1e3);
    }
    return 0;
}
