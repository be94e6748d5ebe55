// Calibration: what does an LDS-fed bf16 MFMA loop with OUR per-wave tile (128 x 64,
// 6 ds_read_b128 per 8 MFMA 32x32x16, or 12 per 32 MFMA 16x16x32) sustain on this
// device, on random data, with no global traffic and no barriers?
//   ./mfma_ceiling.bin [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// VAR 0: all fragments re-read every k-step (24 KB / wave-iteration)
// VAR 1: half of the A fragments re-read per k-step (16 KB / wave-iteration: models reuse of
//        halo rows across the three dy taps)
// VAR 2: no LDS reads inside the loop (sustained MFMA rate at the clocks the device holds)
template <int SHAPE, int WAVES, int VAR = 0>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 4 ? 2 : 2)) void k(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
    constexpr int LDSB = 76 * 1024;
    __shared__ __attribute__((aligned(16))) char smem[LDSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDSB / 16; i += WAVES * 64) reinterpret_cast<u32x4*>(smem)[i] = src[(blockIdx.x * 131 + i) % 65536];
    __syncthreads();
    float res = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4][2];
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
        const int r = lane & 31, h = lane >> 5;
        for (int it = 0; it < iters; ++it) {
            const int tapoff = (it % 9) * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 2 * ks + h;
                u32x4 a[4], b[2];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int row = (wave & 1) * 128 + m * 32 + r + (it % 9);
                    a[m] = *reinterpret_cast<const u32x4*>(smem + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int row = 352 + n * 32 + r;
                    b[n] = *reinterpret_cast<const u32x4*>(smem + row * 128 + ((c ^ ((row >> 1) & 7)) << 4) + tapoff % 1024);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, b[n]), acc[m][n], 0, 0, 0);
            }
        }
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int e = 0; e < 16; ++e) res += acc[m][n][e];
    } else {
        f32x4 acc[8][4];
        for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) for (int e = 0; e < 4; ++e) acc[m][n][e] = 0.f;
        const int r = lane & 15, q = lane >> 4;
        u32x4 a[8], b[4];
        for (int it = 0; it < iters; ++it) {
            const int tapoff = (it % 9) * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int c = 4 * ks + q;
                if (VAR != 2 || it == 0) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (VAR == 1 && it > 0 && (m >> 2) != ks) continue;
                    const int row = (wave & 1) * 128 + m * 16 + r + (it % 9);
                    a[m] = *reinterpret_cast<const u32x4*>(smem + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int row = 352 + n * 16 + r;
                    b[n] = *reinterpret_cast<const u32x4*>(smem + row * 128 + ((c ^ ((row >> 1) & 7)) << 4) + tapoff % 1024);
                }
                }
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, b[n]), acc[m][n], 0, 0, 0);
            }
        }
        for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) for (int e = 0; e < 4; ++e) res += acc[m][n][e];
    }
    out[blockIdx.x * WAVES * 64 + tid] = res;
}

template <int SHAPE, int WAVES, int VAR = 0>
static void run(const u32x4* src, float* out, int iters, int blocks, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, WAVES, VAR>), dim3(blocks), dim3(WAVES * 64), 0, 0, src, out, iters);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, WAVES, VAR>), dim3(blocks), dim3(WAVES * 64), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per wave per iter: 128 x 64 x 64 MACs
    const double fl = 2.0 * 128 * 64 * 64 * (double)iters * WAVES * blocks;
    printf("%-34s blocks=%d iters=%d: %.3f ms  %.0f TF/s\n", name, blocks, iters, best, fl / best / 1e9);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    std::vector<uint16_t> h(65536 * 8);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    u32x4* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 4096 * 512 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    // warm the clocks
    for (int i = 0; i < 3; ++i) run<32, 4>(src, out, iters, 512, "warm");
    run<32, 4>(src, out, iters, 512, "32x32x16  4 waves, 2 blocks/CU");
    run<16, 4>(src, out, iters, 512, "16x16x32  4 waves, 2 blocks/CU");
    run<32, 4>(src, out, iters, 256, "32x32x16  4 waves, 1 block/CU");
    run<16, 4>(src, out, iters, 256, "16x16x32  4 waves, 1 block/CU");
    run<32, 8>(src, out, iters, 256, "32x32x16  8 waves, 1 block/CU");
    run<16, 8>(src, out, iters, 256, "16x16x32  8 waves, 1 block/CU");
    run<16, 4, 1>(src, out, iters, 512, "16x16x32 4w 2blk/CU, A half re-read");
    run<16, 4, 2>(src, out, iters, 512, "16x16x32 4w 2blk/CU, no LDS reads");
    run<16, 4, 2>(src, out, iters, 256, "16x16x32 4w 1blk/CU, no LDS reads");
    return 0;
}
