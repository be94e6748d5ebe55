// Shared device/host helpers for the fea_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include "../../include/fea_unet.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Activations and packed weights are stored as "128-byte channel rows": one
// row holds CK consecutive channels of one pixel (or of one output channel of
// a packed weight).  16-byte chunk = EPC elements.
template <typename T> struct Elem;
template <> struct Elem<float>  { static constexpr int CK = 32; static constexpr int EPC = 4; static constexpr int DT = FEA_F32; };
template <> struct Elem<__bf16> { static constexpr int CK = 64; static constexpr int EPC = 8; static constexpr int DT = FEA_BF16; };

// One "k-step" = the same 16-byte chunk of an A row and a B row in every lane:
// bf16 -> one 32x32x16 MFMA, fp32 -> four exact-f32 32x32x2 MFMAs.  A and B use
// the same lane->k map, so any consistent k order is correct.
__device__ __forceinline__ void mma16(f32x16& acc, const u32x4& a, const u32x4& b, const float*) {
    // NOTE: bit-cast the WHOLE vector, then index.  `__builtin_bit_cast(float, a[t])`
    // on a vector element is miscompiled by hipcc 7.2 (every t reads element 0).
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x16& acc, const u32x4& a, const u32x4& b, const __bf16*) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// 128 zero bytes: the source of LDS-DMA loads for out-of-image (zero padding) rows.
static __device__ __attribute__((aligned(128))) const uint32_t fea_zero_line[32] = {0};

// Sum of x over the four 16-lane rows of the wave (lanes l, l^16, l^32, l^48), in every
// lane, on the VALU (gfx950 v_permlane16/32_swap; no LDS round trip).  Inline asm: the
// builtins mis-handle identical operands in hipcc 7.2.  s_nop 1 = the two wait states
// required between a VALU write of an operand and the swap.
__device__ __forceinline__ float row_sum4(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b; b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// XCD-aware bijective block remap (blocks b and b+8 share an XCD/L2): each XCD
// gets a contiguous range of logical tile ids.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// compile-time unrolled loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

// bf16 conv3x3 with Cout (GEMM N) not a multiple of 128 runs the 64-column kernel, which
// loads its weight fragments global -> VGPR and wants them packed in fragment order
// (igemm.hip DB path, elementwise.hip pack_conv3x3_kernel).  One rule for both sides.
#ifndef IG_DB128
#define IG_DB128 0     // 1: the 128-column 3x3 kernel also takes its weight fragments global -> VGPR (fragment-ordered pack, no weight tile in LDS)
#endif
__host__ __device__ inline bool fea_conv3x3_frag_pack(int gemm_n, int elem_bytes) { return elem_bytes == 2 && (gemm_n % 128 != 0 || IG_DB128); }

// Deterministic sum of the pixel-split slabs of the weight-gradient kernels: out[i] = sum over sp of slab[sp][i], i in float4
// units.  With hundreds of splits and a small weight (64 -> 64: 256 splits of 147 KB) one thread per element is a chain of
// 256 dependent strided loads on 36 workgroups: 0.62 ms for 38 MB.  Here SL lanes share an element: lane j adds splits
// j, j + SL, ... into four independent partial sums, and the SL lane sums are then added in lane order through LDS -- a fixed
// association for a given (splits, SL), so the result is bit-identical from run to run.
template <int SL>
static __global__ __launch_bounds__(256) void fea_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                                     int splits, long long per4) {
    constexpr int EPB = 256 / SL;                       // float4 elements per block
    const int e = threadIdx.x % EPB, j = threadIdx.x / EPB;
    const long long i = (long long)blockIdx.x * EPB + e;
    __shared__ f32x4 part[256];
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (i < per4) {
        const f32x4* s = reinterpret_cast<const f32x4*>(slab) + i;
        int sp = j;
        for (; sp + 3 * SL < splits; sp += 4 * SL) {
            a0 += s[(long long)sp * per4]; a1 += s[(long long)(sp + SL) * per4];
            a2 += s[(long long)(sp + 2 * SL) * per4]; a3 += s[(long long)(sp + 3 * SL) * per4];
        }
        for (; sp < splits; sp += SL) a0 += s[(long long)sp * per4];
    }
    const f32x4 a = (a0 + a1) + (a2 + a3);
    if constexpr (SL == 1) {
        if (i < per4) reinterpret_cast<f32x4*>(out)[i] = a;
    } else {
        part[j * EPB + e] = a;
        __syncthreads();
        if (j == 0 && i < per4) {
            f32x4 r = part[e];
#pragma unroll
            for (int k = 1; k < SL; ++k) r += part[k * EPB + e];
            reinterpret_cast<f32x4*>(out)[i] = r;
        }
    }
}
static inline void fea_launch_slab_reduce(const float* slab, float* out, int splits, long long per4, hipStream_t s) {
    if (splits >= 32)     hipLaunchKernelGGL((fea_slab_reduce_kernel<16>), dim3((unsigned)((per4 + 15) / 16)), dim3(256), 0, s, slab, out, splits, per4);
    else if (splits >= 8) hipLaunchKernelGGL((fea_slab_reduce_kernel<4>), dim3((unsigned)((per4 + 63) / 64)), dim3(256), 0, s, slab, out, splits, per4);
    else                  hipLaunchKernelGGL((fea_slab_reduce_kernel<1>), dim3((unsigned)((per4 + 255) / 256)), dim3(256), 0, s, slab, out, splits, per4);
}

// host side -----------------------------------------------------------------
void fea_set_error(const char* fmt, ...);
int fea_wg_split(void);      // api.hip: fea_set_persistent_split
// first_conv.hip: the first layer on the persistent kernel; false = shape not served (*rc untouched), true = launched (*rc = status)
bool fea_first_conv_p(const float* x, const void* w, const float* bias, void* y, int ldy, float* stats, int relu,
                      int N, int C, int H, int W, int Cout, hipStream_t s, int* rc);
// upgemm.hip: ConvTranspose2d forward on the persistent GEMM; false = shape not served (*rc untouched), true = launched (*rc = status)
bool fea_upg_fwd(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy, int N, int H, int W, int Cin, int Cout,
                 hipStream_t stream, int* rc);
#define FEA_CHECK(cond, code, ...) do { if (!(cond)) { fea_set_error(__VA_ARGS__); return (code); } } while (0)
#define FEA_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    fea_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return FEA_E_LAUNCH; } } while (0)
