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
__host__ __device__ inline bool fea_conv3x3_frag_pack(int gemm_n, int elem_bytes) { return elem_bytes == 2 && gemm_n % 128 != 0; }

// host side -----------------------------------------------------------------
void fea_set_error(const char* fmt, ...);
#define FEA_CHECK(cond, code, ...) do { if (!(cond)) { fea_set_error(__VA_ARGS__); return (code); } } while (0)
#define FEA_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    fea_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return FEA_E_LAUNCH; } } while (0)
