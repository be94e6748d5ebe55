// Per-direction HBM rates of plain streaming kernels ("HBM has a ceiling per direction"): write-only, read-only and
// copy over a 2 GiB buffer, 16 bytes per lane, in three grid shapes.  Build: hipcc --offload-arch=gfx950 -O3 -o
// scripts/hbm_dir_probe.bin scripts/hbm_dir_probe.hip ; run on the GPU box: ./scripts/hbm_dir_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// grid-stride over 16-byte words: consecutive lanes write consecutive words (1 KB per wave-instruction)
template <int NT> __global__ __launch_bounds__(256) void fill_k(u32x4* __restrict__ dst, long long n16, unsigned v) {
    const u32x4 val = {v, v + 1, v + 2, v + 3};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(val, dst + i); else dst[i] = val;
    }
}
// every block owns one contiguous chunk (the shape of a tile walk: 256 CUs in 256 distant places)
__global__ __launch_bounds__(256) void fill_chunk_k(u32x4* __restrict__ dst, long long n16, unsigned v) {
    const u32x4 val = {v, v + 1, v + 2, v + 3};
    const long long per = n16 / gridDim.x;
    u32x4* d = dst + per * blockIdx.x;
    for (long long i = threadIdx.x; i < per; i += 256) d[i] = val;
}
__global__ __launch_bounds__(256) void read_k(const u32x4* __restrict__ src, long long n16, unsigned* out) {
    u32x4 acc = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) acc ^= src[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *out = 1;      // (never: keeps the loads alive)
}
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long n16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

template <typename F> static float time_ms(F&& launch, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const long long bytes = 2ll << 30, n16 = bytes / 16;
    u32x4 *a, *b; unsigned* flag;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&flag, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const int grids[] = {256 * 4, 256 * 8, 256 * 32, 1 << 20};
    for (int g : grids) {
        float ms = time_ms([&] { hipLaunchKernelGGL(fill_k<0>, dim3(g), dim3(256), 0, 0, a, n16, 7u); });
        printf("fill      grid %8d  %7.3f ms  %5.2f TB/s written\n", g, ms, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(fill_k<1>, dim3(g), dim3(256), 0, 0, a, n16, 7u); });
        printf("fill nt   grid %8d  %7.3f ms  %5.2f TB/s written\n", g, ms, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(read_k, dim3(g), dim3(256), 0, 0, a, n16, flag); });
        printf("read      grid %8d  %7.3f ms  %5.2f TB/s read\n", g, ms, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(copy_k, dim3(g), dim3(256), 0, 0, a, b, n16); });
        printf("copy      grid %8d  %7.3f ms  %5.2f TB/s read + written\n", g, ms, 2.0 * bytes / ms / 1e9);
    }
    for (int g : {256, 512, 1024, 4096}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(fill_chunk_k, dim3(g), dim3(256), 0, 0, a, n16, 7u); });
        printf("fill chunk grid %7d  %7.3f ms  %5.2f TB/s written\n", g, ms, bytes / ms / 1e9);
    }
    float ms = time_ms([&] { CK(hipMemsetAsync(a, 0, bytes, 0)); });
    printf("hipMemsetAsync           %7.3f ms  %5.2f TB/s written\n", ms, bytes / ms / 1e9);
    CK(hipDeviceSynchronize());
    return 0;
}
