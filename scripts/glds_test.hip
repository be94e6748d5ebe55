// Semantics probe for __builtin_amdgcn_global_load_lds (16 B) on gfx950:
// LDS destination = wave-uniform base + lane*16 ?  per-lane global source.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__global__ void k(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const int* __restrict__ perm) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 256 * 2];   // 4 waves x 1 KB x 2 instrs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // each lane fetches 16 B from row perm[tid]
    const uint32_t* g0 = src + perm[tid] * 4;
    const uint32_t* g1 = src + perm[255 - tid] * 4;
    uint32_t* base = lds + wave * 512;                  // wave-uniform
    __builtin_amdgcn_global_load_lds((gbl_void*)g0, (lds_void*)base, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)g1, (lds_void*)(base + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = tid; i < 2048; i += 256) dst[i] = lds[i];
}

int main() {
    std::vector<uint32_t> h(256 * 4);
    std::vector<int> perm(256);
    for (int i = 0; i < 256; ++i) { perm[i] = (i * 37 + 11) % 256; for (int j = 0; j < 4; ++j) h[i * 4 + j] = i * 16 + j; }
    uint32_t *ds, *dd; int* dp;
    hipMalloc(&ds, 4096); hipMalloc(&dd, 8192); hipMalloc(&dp, 1024);
    hipMemcpy(ds, h.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(dp, perm.data(), 1024, hipMemcpyHostToDevice);
    hipMemset(dd, 0xff, 8192);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, ds, dd, dp);
    std::vector<uint32_t> o(2048);
    hipMemcpy(o.data(), dd, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int tid = w * 64 + l;
                const uint32_t e0 = perm[tid] * 16 + j, e1 = perm[255 - tid] * 16 + j;
                if (o[w * 512 + l * 4 + j] != e0) { if (bad < 5) printf("mismatch0 w%d l%d j%d got %u want %u\n", w, l, j, o[w * 512 + l * 4 + j], e0); ++bad; }
                if (o[w * 512 + 256 + l * 4 + j] != e1) { if (bad < 5) printf("mismatch1 w%d l%d j%d got %u want %u\n", w, l, j, o[w * 512 + 256 + l * 4 + j], e1); ++bad; }
            }
    printf("glds 16B: %s (%d mismatches)\n", bad ? "FAIL" : "OK: dest = uniform base + lane*16, per-lane source", bad);
    return bad != 0;
}
