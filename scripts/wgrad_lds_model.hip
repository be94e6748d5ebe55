// Does the LDS read rate bound the weight-gradient kernel?  Model of its inner loop with no
// global traffic: per 16-pixel k-step, transposing LDS reads (ds_read_b64_tr_b16) feed
// v_mfma_f32_32x32x16_bf16.
//   VAR 0: wgrad2 today   -- wave = 32co x 32ci x 5 taps: 2 + 2*5 reads per 5 MFMA
//   VAR 1: 64co x 32ci x 3 taps: 4 + 2*3 reads per 6 MFMA
//   VAR 2: 64co x 64ci x 2 taps: 4 + 4*2 reads per 8 MFMA
//   VAR 3: no LDS reads (MFMA only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

__device__ __forceinline__ bf16x8 rd(const char* p) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 512));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int VAR, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
    constexpr int LDSB = 64 * 1024;
    __shared__ __attribute__((aligned(16))) char smem[LDSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDSB / 16; i += WAVES * 64) reinterpret_cast<u32x4*>(smem)[i] = src[(blockIdx.x * 131 + i) % 65536];
    __syncthreads();
    constexpr int NM = VAR == 0 ? 1 : 2, NB = VAR == 2 ? 2 : 1, NT = VAR == 0 ? 5 : (VAR == 1 ? 3 : 2);
    f32x16 acc[NM][NB][NT];
    for (int m = 0; m < NM; ++m) for (int b = 0; b < NB; ++b) for (int t = 0; t < NT; ++t) for (int e = 0; e < 16; ++e) acc[m][b][t][e] = 0.f;
    const int base = (lane & 15) * 128 + (lane >> 4) * 8 + wave * 2048;
    bf16x8 A[NM], B[NB][NT];
    for (int it = 0; it < iters; ++it) {
        const int o = (it & 15) * 1024;
        if (VAR != 3 || it == 0) {
#pragma unroll
            for (int m = 0; m < NM; ++m) A[m] = rd(smem + ((base + o + m * 64) & (LDSB - 2048)) + (lane & 3) * 8);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int t = 0; t < NT; ++t) B[b][t] = rd(smem + ((base + o + 16384 + b * 4096 + t * 128) & (LDSB - 2048)) + (lane & 3) * 8);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[m][b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], B[b][t], acc[m][b][t], 0, 0, 0);
    }
    float res = 0.f;
    for (int m = 0; m < NM; ++m) for (int b = 0; b < NB; ++b) for (int t = 0; t < NT; ++t) for (int e = 0; e < 16; ++e) res += acc[m][b][t][e];
    out[blockIdx.x * WAVES * 64 + tid] = res;
}

template <int VAR, int WAVES>
static void run(const u32x4* src, float* out, int iters, const char* name) {
    constexpr int NM = VAR == 0 ? 1 : 2, NB = VAR == 2 ? 2 : 1, NT = VAR == 0 ? 5 : (VAR == 1 ? 3 : 2);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<VAR, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, src, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double fl = 2.0 * 32 * 32 * 16 * NM * NB * NT * (double)iters * WAVES * 256;
    printf("%-44s %.3f ms  %.0f TF/s\n", name, best, fl / best / 1e9);
}

int main() {
    u32x4* src; float* out;
    (void)hipMalloc(&src, 65536 * 16); (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMemset(src, 0x3c, 65536 * 16);
    run<0, 8>(src, out, 4000, "warm");
    run<0, 8>(src, out, 4000, "32x32 x5 taps (today), 8 waves");
    run<1, 8>(src, out, 4000, "64x32 x3 taps, 8 waves");
    run<1, 12>(src, out, 4000, "64x32 x3 taps, 12 waves");
    run<2, 8>(src, out, 4000, "64x64 x2 taps, 8 waves");
    run<3, 8>(src, out, 4000, "no LDS reads, 8 waves");
    return 0;
}
