// Test helper (not product code): K workgroups that each hold a CU for a given number of 100 MHz ticks (s_memrealtime) -- 512 threads and
// 80 KB of LDS, the footprint class of a collective's channel kernel -- so that a persistent kernel's workgroup (90-152 KB of
// LDS, every register of the CU) cannot be placed beside it.   hipcc -O3 --offload-arch=gfx950 -shared -fPIC
// scripts/hold_cus.hip -o scripts/libhold_cus.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(512) void hold_kernel(long long cycles, int* sink) {
    __shared__ int lds[20480];                       // 80 KB
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
    if (lds[(threadIdx.x + 1) & 511] == -1) *sink = 1;      // keeps the LDS array alive
}

extern "C" int hold_cus(int nblocks, long long cycles, int* sink, void* stream) {
    hipLaunchKernelGGL(hold_kernel, dim3(nblocks), dim3(512), 0, (hipStream_t)stream, cycles, sink);
    return (int)hipGetLastError();
}
