// glds_probe.hip -- what buffer_load_dwordx4 ... lds does on this GPU: lane -> LDS address, M0 above 64 KB, a descriptor of zero records, soffset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ __launch_bounds__(256, 1) void probe(const uint32_t* src, uint32_t* out, int n_rec_zero) {
    __shared__ __attribute__((aligned(1024))) char lds[147456];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 147456 / 4; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, 65536, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, n_rec_zero, 0x00020000);
    // wave w: 1 KB piece w of src -> LDS offset 2048 w (low), soffset form -> 100000 + 2048 w (above 64 KB), zero-record descriptor -> 140000 + 1024 w
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + 2048 * wave), 16, lane * 16 + wave * 1024, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + 100352 + 2048 * wave), 16, lane * 16, wave * 1024 + 4096, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (lds_ptr_t)(lds + 139264 + 1024 * wave), 16, lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 147456 / 4; i += 256) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}
int main() {
    uint32_t *dsrc, *dout;
    std::vector<uint32_t> h(16384), o(147456 / 4);
    for (int i = 0; i < 16384; i++) h[i] = 0x10000000u + i;
    hipMalloc(&dsrc, 65536); hipMalloc(&dout, 147456);
    hipMemcpy(dsrc, h.data(), 65536, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, dsrc, dout, 0);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    hipMemcpy(o.data(), dout, 147456, hipMemcpyDeviceToHost);
    int bad_lo = 0, bad_hi = 0, zero_written = 0, untouched = 0;
    for (int w = 0; w < 4; w++)
        for (int i = 0; i < 256; i++) {
            if (o[(2048 * w) / 4 + i] != 0x10000000u + 256 * w + i) bad_lo++;
            if (o[(100352 + 2048 * w) / 4 + i] != 0x10000000u + 1024 + 256 * w + i) bad_hi++;
            const uint32_t z = o[(139264 + 1024 * w) / 4 + i];
            if (z == 0) zero_written++; else if (z == 0xdeadbeefu) untouched++;
        }
    int stray = 0;
    for (int i = 0; i < 147456 / 4; i++) {
        const int b = i * 4;
        const bool expect = (b < 8192 && (b % 2048) < 1024) || (b >= 100352 && b < 100352 + 8192 && ((b - 100352) % 2048) < 1024) || (b >= 139264 && b < 139264 + 4096);
        if (!expect && o[i] != 0xdeadbeefu) { if (stray < 8) printf("  stray write at LDS byte %d: %08x\n", b, o[i]); stray++; }
    }
    printf("glds probe: low image bad %d / 1024, image above 64 KB bad %d / 1024, zero-record loads: %d words zeroed, %d untouched; stray %d\n", bad_lo, bad_hi, zero_written, untouched, stray);
    if (bad_hi) { printf("  first words at 100352: %08x %08x %08x %08x; at 100352 - 65536 = %d: %08x\n", o[100352 / 4], o[100352 / 4 + 1], o[100352 / 4 + 2], o[100352 / 4 + 3], 100352 - 65536, o[(100352 - 65536) / 4]); }
    return 0;
}
