// 1x1 classification head (models.py:23,46) and the build-defined segmentation
// loss / mask / IoU.  All bandwidth-bound.
#include "common.h"
#include <mutex>

// 8 lanes cooperate on one pixel: lane `sub` owns input channels 8*sub..8*sub+7.
template <typename T> __device__ __forceinline__ void load8(const T* p, float* f);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <> __device__ __forceinline__ void load8<__bf16>(const __bf16* p, float* f) {
    const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)b[j];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* f);
template <> __device__ __forceinline__ void store8<float>(float* p, const float* f) {
    f32x4 a, b;
    a.x = f[0]; a.y = f[1]; a.z = f[2]; a.w = f[3]; b.x = f[4]; b.y = f[5]; b.z = f[6]; b.w = f[7];
    *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<__bf16>(__bf16* p, const float* f) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)f[j];
    *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, b);
}

template <typename T, int NC>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ a, int lda, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ logits,
                                                       long long npix, long long HW, int Ccls) {
    __shared__ float ws[NC * 64 + NC];
    for (int i = threadIdx.x; i < NC * 64; i += 256) ws[i] = i < Ccls * 64 ? w[i] : 0.f;
    if (threadIdx.x < NC) ws[NC * 64 + threadIdx.x] = threadIdx.x < Ccls ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const long long stride = (long long)gridDim.x * 32;
    for (long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); p < npix; p += stride) {
        float f[8];
        load8<T>(a + p * lda + sub * 8, f);
        float acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s = __builtin_fmaf(f[j], ws[c * 64 + sub * 8 + j], s);
            acc[c] = s;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[c] += __shfl_xor(acc[c], 1);
            acc[c] += __shfl_xor(acc[c], 2);
            acc[c] += __shfl_xor(acc[c], 4);
        }
        const long long n = p / HW, hw = p - n * HW;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if ((c & 7) == sub && c < Ccls) logits[(n * Ccls + c) * HW + hw] = acc[c] + ws[NC * 64 + c];
    }
}

// Four consecutive pixels per 8-lane group (HW % 4 == 0): four 16-byte loads in flight per
// lane, and after a halving exchange (xor 4, 2, 1: each step a lane keeps half of its
// classes and adds the partner's partial sums of them) lane `sub` holds class `sub` of the
// four pixels and stores them as one float4 -- 128 contiguous bytes per plane per wave.
template <typename T>
__global__ __launch_bounds__(256) void head_fwd4_kernel(const T* __restrict__ a, int lda, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ logits,
                                                        long long nquad, long long HW, int Ccls) {
    __shared__ float ws[8 * 64 + 8];
    for (int i = threadIdx.x; i < 8 * 64; i += 256) ws[i] = i < Ccls * 64 ? w[i] : 0.f;
    if (threadIdx.x < 8) ws[8 * 64 + threadIdx.x] = threadIdx.x < Ccls ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const float bsub = ws[8 * 64 + sub];
    const long long stride = (long long)gridDim.x * 32;
    for (long long q = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); q < nquad; q += stride) {
        const long long p0 = q * 4;
        float f[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) load8<T>(a + (p0 + k) * lda + sub * 8, f[k]);
        float v[8][4];
        int woff = sub * 8;
        asm volatile("" : "+v"(woff));         // opaque: the weight reads stay in the loop (64 fewer live registers)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float wr[8];                       // this lane's 8 weights of class c
#pragma unroll
            for (int j = 0; j < 8; ++j) wr[j] = ws[c * 64 + woff + j];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s_ = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s_ = __builtin_fmaf(f[k][j], wr[j], s_);
                v[c][k] = s_;
            }
        }
        float v4[4][4], v2[2][4], v1[4];
        const bool b4 = sub & 4, b2 = sub & 2, b1 = sub & 1;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) v4[c][k] = (b4 ? v[c + 4][k] : v[c][k]) + __shfl_xor(b4 ? v[c][k] : v[c + 4][k], 4);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) v2[c][k] = (b2 ? v4[c + 2][k] : v4[c][k]) + __shfl_xor(b2 ? v4[c][k] : v4[c + 2][k], 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) v1[k] = (b1 ? v2[1][k] : v2[0][k]) + __shfl_xor(b1 ? v2[0][k] : v2[1][k], 1);
        if (sub < Ccls) {
            const long long n = p0 / HW, hw = p0 - n * HW;
            f32x4 o = {v1[0] + bsub, v1[1] + bsub, v1[2] + bsub, v1[3] + bsub};
            *reinterpret_cast<f32x4*>(logits + (n * Ccls + sub) * HW + hw) = o;
        }
    }
}

// da = W^T dl ; partial dW / db.  part[blk][c*65 + k], k = 64 -> db.
template <typename T, int NC>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dl, const T* __restrict__ a, int lda,
                                                       const float* __restrict__ w, T* __restrict__ da, int ldda,
                                                       float* __restrict__ part, long long npix, long long HW, int Ccls,
                                                       long long pix_per_block) {
    __shared__ float ws[NC * 64];
    __shared__ float red[4][NC * 65];
    for (int i = threadIdx.x; i < NC * 64; i += 256) ws[i] = i < Ccls * 64 ? w[i] : 0.f;
    __syncthreads();
    const int sub = threadIdx.x & 7;
    float gw[NC][8], gb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        gb[c] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) gw[c][j] = 0.f;
    }
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    const long long p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
    for (long long p = p0 + (threadIdx.x >> 3); p < p1; p += 32) {
        const long long n = p / HW, hw = p - n * HW;
        float f[8], o[8];
        load8<T>(a + p * lda + sub * 8, f);
        int woff = sub * 8;
        asm volatile("" : "+v"(woff));         // opaque: the weight reads stay in the loop (64 fewer live registers, 2x the waves)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float d = c < Ccls ? dl[(n * Ccls + c) * HW + hw] : 0.f;
            gb[c] += d;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = __builtin_fmaf(d, ws[c * 64 + woff + j], o[j]);
                gw[c][j] = __builtin_fmaf(d, f[j], gw[c][j]);
            }
        }
        store8<T>(da + p * ldda + sub * 8, o);
    }
    // reduce over the 8 pixel slots of each wave (lanes with equal sub), then over the 4 waves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = gw[c][j];
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            gw[c][j] = v;
        }
        float v = gb[c];
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        gb[c] = v;
    }
    if (lane < 8) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wv][c * 65 + sub * 8 + j] = gw[c][j];
            if (sub == 0) red[wv][c * 65 + 64] = gb[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Ccls * 65; i += 256)
        part[(long long)blockIdx.x * (Ccls * 65) + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

__global__ __launch_bounds__(256) void head_bwd_finalize_kernel(const float* __restrict__ part, int nblk, float* dW, float* db,
                                                                int Ccls) {
    // one block per (c, k) entry
    __shared__ double sh[4];
    const int i = blockIdx.x;                 // c*65 + k
    double s = 0.0;
    for (int t = threadIdx.x; t < nblk; t += 256) s += part[(long long)t * (Ccls * 65) + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = sh[0] + sh[1] + sh[2] + sh[3];
        const int c = i / 65, k = i - c * 65;
        if (k < 64) dW[c * 64 + k] = (float)tot; else db[c] = (float)tot;
    }
}

// ---------------------------------------------------------------------------
// loss: per (n, c) plane, blocks over chunks of HW
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// One element of the loss sums.  e = exp(-|x|) serves both the stable BCE term
// max(x,0) - x t + log1p(e) and the sigmoid (1/(1+e) for x >= 0, e/(1+e) otherwise).
__device__ __forceinline__ void loss_terms(float v, float tt, float& sb, float& si, float& sp, float& st) {
    const float e = expf(-fabsf(v));
    const float r = 1.f / (1.f + e);
    const float p = v >= 0.f ? r : e * r;
    sb += fmaxf(v, 0.f) - v * tt + log1pf(e);
    si += p * tt; sp += p; st += tt;
}

// part[blk][4]; blk = (plane * chunks + chunk); plane = n*C + c.  VEC = 4: 16-byte logit loads
// and 4-byte target loads (HW % 4 == 0 and aligned bases), VEC = 1: any shape.
template <int VEC>
__global__ __launch_bounds__(256) void bce_dice_sums_kernel(const float* __restrict__ x, const uint8_t* __restrict__ t,
                                                            float* __restrict__ part, long long HW, int chunks, long long chunk_len) {
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
    const long long b = (long long)plane * HW;
    const long long i0 = chunk * chunk_len, i1 = i0 + chunk_len < HW ? i0 + chunk_len : HW;
    float sb = 0.f, si = 0.f, sp = 0.f, st = 0.f;
    if constexpr (VEC == 4) {
        for (long long i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + b + i);
            const uint32_t tw = *reinterpret_cast<const uint32_t*>(t + b + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) loss_terms(v[k], ((tw >> (8 * k)) & 0xff) ? 1.f : 0.f, sb, si, sp, st);
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += 256) loss_terms(x[b + i], t[b + i] ? 1.f : 0.f, sb, si, sp, st);
    }
    __shared__ float sh[4][4];
    float vals[4] = {sb, si, sp, st};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = vals[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) part[(long long)blockIdx.x * 4 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// sums[c][4] (double) from part; one block per class
__global__ __launch_bounds__(256) void bce_dice_sums_finalize_kernel(const float* __restrict__ part, double* __restrict__ sums,
                                                                     int N, int C, int chunks) {
    __shared__ double sh[4][4];
    const int c = blockIdx.x;
    double a[4] = {0, 0, 0, 0};
    const int per = N * chunks;
    for (int i = threadIdx.x; i < per; i += 256) {
        const int n = i / chunks, ch = i - n * chunks;
        const float* p = part + ((long long)(n * C + c) * chunks + ch) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += p[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double v = a[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) sums[c * 4 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

template <int VEC>
__global__ __launch_bounds__(256) void bce_dice_grad_kernel(const float* __restrict__ x, const uint8_t* __restrict__ t,
                                                            const double* __restrict__ sums, float* __restrict__ loss,
                                                            float* __restrict__ dx, float gscale, int N, int C, long long HW) {
    const long long total = (long long)N * C * HW;
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss) {
        double bce = 0.0, dice = 0.0;
        for (int c = 0; c < C; ++c) {
            bce += sums[c * 4];
            dice += (2.0 * sums[c * 4 + 1] + 1.0) / (sums[c * 4 + 2] + sums[c * 4 + 3] + 1.0);
        }
        loss[0] = (float)(bce / (double)total + 1.0 - dice / C);
    }
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i >= total || !dx) return;
    const int c = (int)((i / HW) % C);            // VEC elements never straddle a plane (HW % VEC == 0)
    const double I = sums[c * 4 + 1], S = sums[c * 4 + 2] + sums[c * 4 + 3];
    const float inv = (float)(1.0 / (S + 1.0));
    const float k2 = (float)((2.0 * I + 1.0) / (S + 1.0));
    const float itot = 1.f / (float)total, ic = 1.f / (float)C;
    auto grad = [&](float v, float tt) {
        const float e = expf(-fabsf(v));
        const float r = 1.f / (1.f + e);
        const float p = v >= 0.f ? r : e * r;
        // d dice_c / d p = (2 t (S+1) - (2I+1)) / (S+1)^2
        const float ddice = (2.f * tt - k2) * inv;
        return ((p - tt) * itot - ddice * p * (1.f - p) * ic) * gscale;
    };
    if constexpr (VEC == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        const uint32_t tw = *reinterpret_cast<const uint32_t*>(t + i);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = grad(v[k], ((tw >> (8 * k)) & 0xff) ? 1.f : 0.f);
        *reinterpret_cast<f32x4*>(dx + i) = o;
    } else {
        dx[i] = grad(x[i], t[i] ? 1.f : 0.f);
    }
}

__global__ __launch_bounds__(256) void mask_iou_kernel(const float* __restrict__ x, const uint8_t* __restrict__ t,
                                                       uint8_t* __restrict__ mask, unsigned long long* __restrict__ inter,
                                                       unsigned long long* __restrict__ uni, long long HW, int C, int chunks,
                                                       long long chunk_len) {
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
    const int c = plane % C;
    const long long b = (long long)plane * HW;
    const long long i0 = chunk * chunk_len, i1 = i0 + chunk_len < HW ? i0 + chunk_len : HW;
    unsigned ni = 0, nu = 0;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const bool pr = x[b + i] > 0.f;
        const bool tg = t ? t[b + i] != 0 : false;          // no target: the mask alone (inference without ground truth)
        if (mask) mask[b + i] = pr ? 1 : 0;
        ni += (pr && tg) ? 1u : 0u;
        nu += (pr || tg) ? 1u : 0u;
    }
    if (!t) return;                                         // block-uniform: no barrier is skipped by part of a block
    __shared__ unsigned sh[4][2];
    for (int o = 32; o > 0; o >>= 1) { ni += __shfl_down(ni, o); nu += __shfl_down(nu, o); }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = ni; sh[threadIdx.x >> 6][1] = nu; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(inter + c, (unsigned long long)(sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0]));
        atomicAdd(uni + c, (unsigned long long)(sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1]));
    }
}

__global__ void binarise_kernel(const uint8_t* __restrict__ g, uint8_t* __restrict__ o, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = g[i] > 127 ? 1 : 0;
}

// ---------------------------------------------------------------------------
static int head_blocks(long long npix) {
    long long b = (npix + 32 * 16 - 1) / (32 * 16);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}
extern "C" int fea_head_bwd_blocks(int N, int H, int W) { return head_blocks((long long)N * H * W); }

extern "C" int fea_head_fwd(const void* a, int lda, const float* w, const float* bias, float* logits,
                            int N, int H, int W, int Cin, int Ccls, int dtype, void* stream) {
    FEA_CHECK(a && w && bias && logits && N > 0 && H > 0 && W > 0, FEA_E_BADARG, "fea_head_fwd: bad args");
    FEA_CHECK(Cin == 64, FEA_E_UNSUPPORTED_SHAPE, "fea_head_fwd: Cin=%d (the head is Conv2d(64, classes, 1))", Cin);
    FEA_CHECK(Ccls >= 1 && Ccls <= 16, FEA_E_UNSUPPORTED_SHAPE, "fea_head_fwd: %d classes (1..16 supported)", Ccls);
    FEA_CHECK(lda >= 64 && lda % 8 == 0, FEA_E_BADARG, "fea_head_fwd: lda=%d", lda);
    const long long npix = (long long)N * H * W, HW = (long long)H * W;
    const int nblk = head_blocks(npix);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FEA_BF16 && Ccls <= 8 && HW % 4 == 0 && (uintptr_t)logits % 16 == 0) {
        hipLaunchKernelGGL((head_fwd4_kernel<__bf16>), dim3(nblk), dim3(256), 0, s, (const __bf16*)a, lda, w, bias, logits, npix / 4, HW, Ccls);
    } else if (dtype == FEA_BF16) {
        if (Ccls <= 8) hipLaunchKernelGGL((head_fwd_kernel<__bf16, 8>), dim3(nblk), dim3(256), 0, s, (const __bf16*)a, lda, w, bias, logits, npix, HW, Ccls);
        else hipLaunchKernelGGL((head_fwd_kernel<__bf16, 16>), dim3(nblk), dim3(256), 0, s, (const __bf16*)a, lda, w, bias, logits, npix, HW, Ccls);
    } else if (dtype == FEA_F32) {
        if (Ccls <= 8) hipLaunchKernelGGL((head_fwd_kernel<float, 8>), dim3(nblk), dim3(256), 0, s, (const float*)a, lda, w, bias, logits, npix, HW, Ccls);
        else hipLaunchKernelGGL((head_fwd_kernel<float, 16>), dim3(nblk), dim3(256), 0, s, (const float*)a, lda, w, bias, logits, npix, HW, Ccls);
    } else { fea_set_error("fea_head_fwd: bad dtype"); return FEA_E_BADARG; }
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_head_bwd(const float* dlogits, const void* a, int lda, const float* w, void* da, int ldda, float* part,
                            int N, int H, int W, int Cin, int Ccls, int dtype, void* stream) {
    FEA_CHECK(dlogits && a && w && da && part && N > 0 && H > 0 && W > 0, FEA_E_BADARG, "fea_head_bwd: bad args");
    FEA_CHECK(Cin == 64, FEA_E_UNSUPPORTED_SHAPE, "fea_head_bwd: Cin=%d", Cin);
    FEA_CHECK(Ccls >= 1 && Ccls <= 16, FEA_E_UNSUPPORTED_SHAPE, "fea_head_bwd: %d classes (1..16 supported)", Ccls);
    FEA_CHECK(lda >= 64 && lda % 8 == 0 && ldda >= 64 && ldda % 8 == 0, FEA_E_BADARG, "fea_head_bwd: ld");
    const long long npix = (long long)N * H * W, HW = (long long)H * W;
    const int nblk = head_blocks(npix);
    long long ppb = (npix + nblk - 1) / nblk;
    ppb = (ppb + 31) / 32 * 32;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FEA_BF16) {
        if (Ccls <= 8) hipLaunchKernelGGL((head_bwd_kernel<__bf16, 8>), dim3(nblk), dim3(256), 0, s, dlogits, (const __bf16*)a, lda, w, (__bf16*)da, ldda, part, npix, HW, Ccls, ppb);
        else hipLaunchKernelGGL((head_bwd_kernel<__bf16, 16>), dim3(nblk), dim3(256), 0, s, dlogits, (const __bf16*)a, lda, w, (__bf16*)da, ldda, part, npix, HW, Ccls, ppb);
    } else if (dtype == FEA_F32) {
        if (Ccls <= 8) hipLaunchKernelGGL((head_bwd_kernel<float, 8>), dim3(nblk), dim3(256), 0, s, dlogits, (const float*)a, lda, w, (float*)da, ldda, part, npix, HW, Ccls, ppb);
        else hipLaunchKernelGGL((head_bwd_kernel<float, 16>), dim3(nblk), dim3(256), 0, s, dlogits, (const float*)a, lda, w, (float*)da, ldda, part, npix, HW, Ccls, ppb);
    } else { fea_set_error("fea_head_bwd: bad dtype"); return FEA_E_BADARG; }
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_head_bwd_finalize(const float* part, int nblk, float* dW, float* db, int Cin, int Ccls, void* stream) {
    FEA_CHECK(part && dW && db && nblk > 0 && Cin == 64 && Ccls >= 1 && Ccls <= 16, FEA_E_BADARG, "fea_head_bwd_finalize: bad args");
    hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(Ccls * 65), dim3(256), 0, (hipStream_t)stream, part, nblk, dW, db, Ccls);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

static void loss_plan(long long HW, int& chunks, long long& chunk_len) {
    chunk_len = 256 * 32;
    chunks = (int)((HW + chunk_len - 1) / chunk_len);
    if (chunks < 1) chunks = 1;
}
extern "C" int fea_loss_blocks(int N, int Ccls, long long HW) {
    int chunks; long long cl; loss_plan(HW, chunks, cl);
    return N * Ccls * chunks;
}

extern "C" int fea_bce_dice_sums(const float* logits, const uint8_t* target, double* sums, float* part,
                                 int N, int Ccls, long long HW, void* stream) {
    FEA_CHECK(logits && target && sums && part && N > 0 && Ccls > 0 && HW > 0, FEA_E_BADARG, "fea_bce_dice_sums: bad args");
    int chunks; long long cl; loss_plan(HW, chunks, cl);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = HW % 4 == 0 && (uintptr_t)logits % 16 == 0 && (uintptr_t)target % 4 == 0;
    if (vec) hipLaunchKernelGGL(bce_dice_sums_kernel<4>, dim3(N * Ccls * chunks), dim3(256), 0, s, logits, target, part, HW, chunks, cl);
    else     hipLaunchKernelGGL(bce_dice_sums_kernel<1>, dim3(N * Ccls * chunks), dim3(256), 0, s, logits, target, part, HW, chunks, cl);
    FEA_LAUNCH_CHECK();
    hipLaunchKernelGGL(bce_dice_sums_finalize_kernel, dim3(Ccls), dim3(256), 0, s, part, sums, N, Ccls, chunks);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

// ---------------------------------------------------------------------------
// Head + loss fused (training loop): the logits never go to HBM.
//   pass 1 (head_loss_sums_kernel): logits of 4 consecutive pixels in registers -> per-block (bce, p*t, p, t) sums per class
//   pass 2 (head_loss_bwd_kernel):  logits recomputed, dlogits formed in registers from the finalized sums, then exactly
//                                   head_bwd_kernel's work: da = W^T dl and the dW / db partials
// Same arithmetic per element as head_fwd4_kernel + bce_dice_*_kernel + head_bwd_kernel (the sums are accumulated in a
// different order).  Requires HW % 4 == 0, <= 8 classes, bf16 or fp32 activations.
template <typename T>
__global__ __launch_bounds__(256) void head_loss_sums_kernel(const T* __restrict__ a, int lda, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const uint8_t* __restrict__ tgt,
                                                             float* __restrict__ logits, float* __restrict__ part,
                                                             long long nquad, long long HW, int Ccls) {
    __shared__ float ws[8 * 64 + 8];
    __shared__ float red[4][8][4];
    for (int i = threadIdx.x; i < 8 * 64; i += 256) ws[i] = i < Ccls * 64 ? w[i] : 0.f;
    if (threadIdx.x < 8) ws[8 * 64 + threadIdx.x] = threadIdx.x < Ccls ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const float bsub = ws[8 * 64 + sub];
    float sb = 0.f, si = 0.f, sp = 0.f, st = 0.f;
    const long long stride = (long long)gridDim.x * 32;
    for (long long q = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); q < nquad; q += stride) {
        const long long p0 = q * 4;
        float f[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) load8<T>(a + (p0 + k) * lda + sub * 8, f[k]);
        const long long n = p0 / HW, hw = p0 - n * HW;
        uint32_t tw = 0;
        if (sub < Ccls) tw = *reinterpret_cast<const uint32_t*>(tgt + (n * Ccls + sub) * HW + hw);
        float v[8][4];
        int woff = sub * 8;
        asm volatile("" : "+v"(woff));
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float wr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wr[j] = ws[c * 64 + woff + j];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s_ = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s_ = __builtin_fmaf(f[k][j], wr[j], s_);
                v[c][k] = s_;
            }
        }
        float v4[4][4], v2[2][4], v1[4];
        const bool b4 = sub & 4, b2 = sub & 2, b1 = sub & 1;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) v4[c][k] = (b4 ? v[c + 4][k] : v[c][k]) + __shfl_xor(b4 ? v[c][k] : v[c + 4][k], 4);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) v2[c][k] = (b2 ? v4[c + 2][k] : v4[c][k]) + __shfl_xor(b2 ? v4[c][k] : v4[c + 2][k], 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) v1[k] = (b1 ? v2[1][k] : v2[0][k]) + __shfl_xor(b1 ? v2[0][k] : v2[1][k], 1);
        if (sub < Ccls) {
            const f32x4 o = {v1[0] + bsub, v1[1] + bsub, v1[2] + bsub, v1[3] + bsub};
            if (logits) *reinterpret_cast<f32x4*>(logits + (n * Ccls + sub) * HW + hw) = o;
#pragma unroll
            for (int k = 0; k < 4; ++k) loss_terms(o[k], ((tw >> (8 * k)) & 0xff) ? 1.f : 0.f, sb, si, sp, st);
        }
    }
    // lanes with equal sub hold the same class: sum them over the wave, then over the four waves
    float vals[4] = {sb, si, sp, st};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = vals[k];
        x += __shfl_xor(x, 8); x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
        if ((threadIdx.x & 63) < 8) red[threadIdx.x >> 6][sub][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < Ccls * 4) {
        const int c = threadIdx.x >> 2, k = threadIdx.x & 3;
        part[((long long)blockIdx.x * Ccls + c) * 4 + k] = red[0][c][k] + red[1][c][k] + red[2][c][k] + red[3][c][k];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_loss_bwd_kernel(const T* __restrict__ a, int lda, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const uint8_t* __restrict__ tgt,
                                                            const double* __restrict__ sums, float* __restrict__ loss,
                                                            T* __restrict__ da, int ldda, float* __restrict__ part,
                                                            float gscale, long long npix, long long HW, int N, int Ccls,
                                                            long long pix_per_block) {
    constexpr int NC = 8, PP = 2;                     // pixels per 8-lane group and iteration
    __shared__ float ws[NC * 64 + NC];
    __shared__ float cf[NC][2];                       // per class: 1/(S+1), (2I+1)/(S+1)
    __shared__ float red[4][NC * 65];
    for (int i = threadIdx.x; i < NC * 64; i += 256) ws[i] = i < Ccls * 64 ? w[i] : 0.f;
    if (threadIdx.x < NC) {
        const int c = threadIdx.x;
        ws[NC * 64 + c] = c < Ccls ? bias[c] : 0.f;
        const double I = c < Ccls ? sums[c * 4 + 1] : 0.0, S = c < Ccls ? sums[c * 4 + 2] + sums[c * 4 + 3] : 0.0;
        cf[c][0] = (float)(1.0 / (S + 1.0));
        cf[c][1] = (float)((2.0 * I + 1.0) / (S + 1.0));
    }
    const long long total = (long long)N * Ccls * HW;
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss) {
        double bce = 0.0, dice = 0.0;
        for (int c = 0; c < Ccls; ++c) {
            bce += sums[c * 4];
            dice += (2.0 * sums[c * 4 + 1] + 1.0) / (sums[c * 4 + 2] + sums[c * 4 + 3] + 1.0);
        }
        loss[0] = (float)(bce / (double)total + 1.0 - dice / Ccls);
    }
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const float itot = 1.f / (float)total, ic = 1.f / (float)Ccls;
    float gw[NC][8], gb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        gb[c] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) gw[c][j] = 0.f;
    }
    const long long p_begin = (long long)blockIdx.x * pix_per_block;
    const long long p_end = p_begin + pix_per_block < npix ? p_begin + pix_per_block : npix;
    for (long long p0 = p_begin + (threadIdx.x >> 3) * PP; p0 < p_end; p0 += 32 * PP) {
        const long long n = p0 / HW, hw = p0 - n * HW;
        float f[PP][8], o[PP][8];
#pragma unroll
        for (int k = 0; k < PP; ++k) load8<T>(a + (p0 + k) * lda + sub * 8, f[k]);
        int woff = sub * 8;
        asm volatile("" : "+v"(woff));
#pragma unroll
        for (int k = 0; k < PP; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) o[k][j] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c < Ccls) {                                  // (uniform)
                const uint16_t tw = *reinterpret_cast<const uint16_t*>(tgt + (n * Ccls + c) * HW + hw);
                float wr[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wr[j] = ws[c * 64 + woff + j];
#pragma unroll
                for (int k = 0; k < PP; ++k) {
                    float s_ = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) s_ = __builtin_fmaf(f[k][j], wr[j], s_);
                    s_ += __shfl_xor(s_, 1); s_ += __shfl_xor(s_, 2); s_ += __shfl_xor(s_, 4);
                    const float v = s_ + ws[NC * 64 + c];
                    const float tt = ((tw >> (8 * k)) & 0xff) ? 1.f : 0.f;
                    const float e = expf(-fabsf(v));
                    const float r = 1.f / (1.f + e);
                    const float pr = v >= 0.f ? r : e * r;
                    const float ddice = (2.f * tt - cf[c][1]) * cf[c][0];
                    const float d = ((pr - tt) * itot - ddice * pr * (1.f - pr) * ic) * gscale;
                    gb[c] += d;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        o[k][j] = __builtin_fmaf(d, wr[j], o[k][j]);
                        gw[c][j] = __builtin_fmaf(d, f[k][j], gw[c][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PP; ++k) store8<T>(da + (p0 + k) * ldda + sub * 8, o[k]);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = gw[c][j];
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            gw[c][j] = v;
        }
        float v = gb[c];
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        gb[c] = v;
    }
    if (lane < 8) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wv][c * 65 + sub * 8 + j] = gw[c][j];
            if (sub == 0) red[wv][c * 65 + 64] = gb[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Ccls * 65; i += 256)
        part[(long long)blockIdx.x * (Ccls * 65) + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// ---------------------------------------------------------------------------
// Head + loss fused, MFMA form (bf16 activations, H*W % 16 == 0, <= 8 classes): the same two passes as above, with both
// small GEMMs of a 16-pixel tile on the matrix pipe instead of 8-lane dot products:
//   logits^T[class][pixel] = W[class][k] x^T[k][pixel]        A = weight (rows = classes, padded to 16), B = 16 pixels
//   da^T[ch][pixel]        = W^T[ch][class] d[class][pixel]   A = weight^T (four 16-channel row blocks), B = dlogits
// The first product leaves lane (pixel = lane % 16, q = lane / 16) with rows 4q .. 4q+3 of its pixel in the four
// accumulator registers.  Weight row r holds class (r % 4) * 4 + r / 4, so register e of lane group q is class q + 4e: up
// to 8 classes occupy registers 0 and 1 of ALL four lane groups (rows in class order would leave 6 classes in 4 + 2
// registers of two groups, and the loss arithmetic -- which issues per register, not per active lane -- twice as long).
// The second product takes those registers as its B operand as they stand: k-slot (q, j) of the MFMA stands for class
// q + 4j (slots 4..7 are zero on both sides) -- no lane exchange between the two.  The fp32 weight enters as
// hi + lo bf16 pairs (two MFMAs), dlogits likewise (d_hi W_hi + d_lo W_hi + d_hi W_lo), so the results agree with the fp32
// arithmetic of the separate kernels to ~2^-16 before the final bf16 rounding of da.  dW / db stay on the VALU: every lane
// fetches the (<= 8) dlogits of its pixel from the lanes that hold them and multiplies them into its 16 channels.
// Both kernels are streams that were instruction-bound before they were HBM-bound: a wave's tile position (image, pixel
// offset) is wave-uniform 32-bit state advanced by additions (no per-tile 64-bit divisions), the loop has no branches on
// lane predicates (lanes of absent classes compute on zero rows and are dropped at the end), and da leaves as one 16-byte
// store per lane and block pair.
__device__ __forceinline__ uint32_t hl_pack(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
    const f2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2_t));
}
__device__ __forceinline__ float hl_lo(float w) { return w - (float)(__bf16)w; }        // what bf16(w) leaves out

struct HlFrag { u32x4 ah[2], al[2]; float bias[4]; };
// GEMM 1 A operand: lane (row = lane % 16, q): k = q*8 + j  <->  input channel ks*32 + q*8 + j; row r <-> class (r%4)*4 + r/4
__device__ __forceinline__ void hl_weights(HlFrag& f, const float* __restrict__ w, const float* __restrict__ bias, int Ccls, int r16, int q) {
    const int cr = (r16 & 3) * 4 + (r16 >> 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = cr < Ccls ? w[cr * 64 + ks * 32 + q * 8 + j] : 0.f;
        f.ah[ks] = u32x4{hl_pack(v[0], v[1]), hl_pack(v[2], v[3]), hl_pack(v[4], v[5]), hl_pack(v[6], v[7])};
        f.al[ks] = u32x4{hl_pack(hl_lo(v[0]), hl_lo(v[1])), hl_pack(hl_lo(v[2]), hl_lo(v[3])),
                         hl_pack(hl_lo(v[4]), hl_lo(v[5])), hl_pack(hl_lo(v[6]), hl_lo(v[7]))};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) f.bias[e] = q + 4 * e < Ccls ? bias[q + 4 * e] : 0.f;
}
__device__ __forceinline__ f32x4 hl_logits(const HlFrag& f, const u32x4& x0, const u32x4& x1) {
    f32x4 acc = {f.bias[0], f.bias[1], f.bias[2], f.bias[3]};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.ah[0]), __builtin_bit_cast(bf16x8, x0), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.al[0]), __builtin_bit_cast(bf16x8, x0), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.ah[1]), __builtin_bit_cast(bf16x8, x1), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.al[1]), __builtin_bit_cast(bf16x8, x1), acc, 0, 0, 0);
    return acc;
}

// Wave-uniform position of a wave's 16-pixel tile: tile t, image n, pixel offset hw inside the image.  Tiles never straddle
// images (H*W % 16 == 0).  All 32-bit (the launchers check N * H*W * classes < 2^31).
struct HlWalk {
    int t, n, hw, step_t, step_n, step_hw;
    __device__ __forceinline__ void init(int t0, int nw, int HW) {
        t = t0;
        const int p0 = t0 * 16;
        n = p0 / HW; hw = p0 - n * HW;
        step_t = nw;
        const int sp = nw * 16;
        step_n = sp / HW; step_hw = sp - step_n * HW;
    }
    __device__ __forceinline__ void next(int HW) {
        t += step_t; n += step_n; hw += step_hw;
        if (hw >= HW) { hw -= HW; ++n; }
    }
};

// One tile's operands.  Its loads are inline asm: hipcc sizes the vmcnt wait in front of a tile for the loop path with the
// FEWEST loads behind it, so with the read-ahead under a branch every tile waited for the loads just issued; duplicating the
// tile code per path spills.  The compiler does not know these loads are pending: hl_wait() retires them by count (vmcnt
// retires in order, loads and stores alike) and every use of a tile comes after it.
template <int NE> struct HlTile { u32x4 x0, x1; uint32_t t[NE]; };

template <int NE, bool NT>
__device__ __forceinline__ void hl_fetch(HlTile<NE>& d, const __bf16* xp, const uint8_t* tgt, long long toff, const int (&lc)[NE]) {
    // (the target plane offset is wave-uniform; readfirstlane says so to the compiler, which otherwise may hand the
    // scalar-base operand below a VGPR pair)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)toff), hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)toff >> 32));
    const uint8_t* tp = tgt + (long long)(((unsigned long long)hi << 32) | lo);
    if constexpr (NT) {
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d.x0) : "v"(xp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:64 nt" : "=v"(d.x1) : "v"(xp) : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d.x0) : "v"(xp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(d.x1) : "v"(xp) : "memory");
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) asm volatile("global_load_ubyte %0, %1, %2" : "=v"(d.t[e]) : "v"(lc[e]), "s"(tp) : "memory");
}
template <int N, int NE>                          // wait until at most N vector-memory operations newer than this tile's loads are pending
__device__ __forceinline__ void hl_wait(HlTile<NE>& d) {
    if constexpr (NE == 1) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(d.x0), "+v"(d.x1), "+v"(d.t[0]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(d.x0), "+v"(d.x1), "+v"(d.t[0]), "+v"(d.t[1]) : "n"(N) : "memory");
}

// Loss terms on the hardware transcendentals (v_exp_f32 / v_rcp_f32 / v_log_f32, ~1 ulp each): the libm sequences of
// loss_terms() (expf, log1pf, a division: ~80 instructions per element) made these kernels VALU-bound.
// |log(1 + e) - log1p(e)| <= 6e-8 absolute per element.
__device__ __forceinline__ void loss_terms_fast(float v, float tt, float& sb, float& si, float& sp, float& st) {
    const float e = __expf(-fabsf(v));
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    const float p = v >= 0.f ? r : e * r;
    sb += fmaxf(v, 0.f) - v * tt + __logf(1.f + e);
    si += p * tt; sp += p; st += tt;
}

#ifndef HL_SUMS_AHEAD
#define HL_SUMS_AHEAD 1
#endif
#ifndef HL_SUMS_NT
#define HL_SUMS_NT 0              // plain loads: 0.49 ms against 0.58 with nontemporal ones (the activation was written by the kernel before and is read again by the backward kernel)
#endif
// BNA: `a` is the PRE-BatchNorm output of dec1.3 and the activation relu(a * scale + shift), rounded to bf16 as fea_bn_relu_pool
// would have stored it, is formed in registers after the load: the activated tensor is never written (fea_head_loss_*_bn)
struct HlBn { float sc[16], sh[16]; };
template <bool BNA>
__device__ __forceinline__ void hl_bn_load(HlBn& b, const float* __restrict__ sc, const float* __restrict__ sh, int q) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ch = (j >> 3) * 32 + q * 8 + (j & 7);        // x0: channels 8q .. 8q+7, x1: 32 + 8q ..
        b.sc[j] = BNA ? sc[ch] : 0.f; b.sh[j] = BNA ? sh[ch] : 0.f;
    }
}
template <bool BNA>
__device__ __forceinline__ void hl_bn_act(const HlBn& b, u32x4& x0, u32x4& x1) {
    if constexpr (BNA) {
        const bf16x8 y0 = __builtin_bit_cast(bf16x8, x0), y1 = __builtin_bit_cast(bf16x8, x1);
        float v[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = fmaxf(__builtin_fmaf((float)y0[j], b.sc[j], b.sh[j]), 0.f);
            v[8 + j] = fmaxf(__builtin_fmaf((float)y1[j], b.sc[8 + j], b.sh[8 + j]), 0.f);
        }
        x0 = u32x4{hl_pack(v[0], v[1]), hl_pack(v[2], v[3]), hl_pack(v[4], v[5]), hl_pack(v[6], v[7])};
        x1 = u32x4{hl_pack(v[8], v[9]), hl_pack(v[10], v[11]), hl_pack(v[12], v[13]), hl_pack(v[14], v[15])};
    }
}

// the same, keeping the pre-BN values and the activations as floats (the BatchNorm-backward modes of the backward kernel)
__device__ __forceinline__ void hl_bn_act_keep(const HlBn& b, u32x4& x0, u32x4& x1, float (&yf)[16], float (&act)[16]) {
    const bf16x8 y0 = __builtin_bit_cast(bf16x8, x0), y1 = __builtin_bit_cast(bf16x8, x1);
#pragma unroll
    for (int j = 0; j < 8; ++j) { yf[j] = (float)y0[j]; yf[8 + j] = (float)y1[j]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) act[j] = fmaxf(__builtin_fmaf(yf[j], b.sc[j], b.sh[j]), 0.f);
    x0 = u32x4{hl_pack(act[0], act[1]), hl_pack(act[2], act[3]), hl_pack(act[4], act[5]), hl_pack(act[6], act[7])};
    x1 = u32x4{hl_pack(act[8], act[9]), hl_pack(act[10], act[11]), hl_pack(act[12], act[13]), hl_pack(act[14], act[15])};
}

template <int NE, bool BNA>                        // accumulator registers that hold classes: (classes + 3) / 4
__global__ __launch_bounds__(256) void head_loss_sums_mfma_kernel(const __bf16* __restrict__ a, int lda, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, const uint8_t* __restrict__ tgt,
                                                                  float* __restrict__ logits, float* __restrict__ part,
                                                                  int ntile, int HW, int Ccls,
                                                                  const float* __restrict__ bn_sc, const float* __restrict__ bn_sh) {
    __shared__ float red[4][8][4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, q = lane >> 4;
    HlFrag f;
    hl_weights(f, w, bias, Ccls, r16, q);
    HlBn bn;
    hl_bn_load<BNA>(bn, bn_sc, bn_sh, q);
    float sb[NE], si[NE], sp[NE], st[NE];
    int lc[NE];                                    // target byte of (class, pixel) relative to (image, hw); absent classes read class q
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        sb[e] = si[e] = sp[e] = st[e] = 0.f;
        lc[e] = (q + 4 * e < Ccls ? q + 4 * e : q < Ccls ? q : 0) * HW + r16;
    }
    const int nw = gridDim.x * 4;
    HlWalk cur, ahead;                                // the tile being multiplied; the next tile to ask for
    cur.init(blockIdx.x * 4 + wv, nw, HW);
    ahead = cur;
    const __bf16* xc = a + ((long long)cur.t * 16 + r16) * lda + q * 8;      // this lane's row of `cur` ...
    const __bf16* xa = xc;                                                   // ... and of `ahead`
    const long long xstep = (long long)nw * 16 * lda;
    const int CHW = Ccls * HW;
    // D tiles of read-ahead on a ring of D + 1 register sets, no copies.  Past the last tile the fetch asks for the current
    // tile again: every tile then has the same number of loads behind it.
    constexpr int D = HL_SUMS_AHEAD;
    HlTile<NE> ring[D + 1];
    auto fetch = [&](HlTile<NE>& d) {
        const bool more = ahead.t < ntile;
        const HlWalk& k = more ? ahead : cur;
        hl_fetch<NE, HL_SUMS_NT != 0>(d, more ? xa : xc, tgt, (long long)k.n * CHW + k.hw, lc);
        xa += xstep; ahead.next(HW);                  // (past the end neither is dereferenced)
    };
    auto tile = [&](const HlTile<NE>& c) {
        u32x4 x0 = c.x0, x1 = c.x1;
        hl_bn_act<BNA>(bn, x0, x1);
        const f32x4 v = hl_logits(f, x0, x1);
        if (logits) {                                                  // (uniform)
            float* lp = logits + ((long long)cur.n * CHW + cur.hw);
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (q + 4 * e < Ccls) lp[lc[e]] = v[e];
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) loss_terms_fast(v[e], c.t[e] ? 1.f : 0.f, sb[e], si[e], sp[e], st[e]);
    };
    // behind a tile's loads when it is waited for: D fetches of 2 + NE loads (and logits stores, if any: the wait is then
    // stronger than needed)
    if (cur.t < ntile) {
        sfor<0, D>([&](auto ic) { fetch(ring[decltype(ic)::value]); });
        bool done = false;
        while (!done) {
            sfor<0, D + 1>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                if (!done) {
                    fetch(ring[(s_ + D) % (D + 1)]);
                    hl_wait<D * (2 + NE)>(ring[s_]);
                    tile(ring[s_]);
                    cur.next(HW); xc += xstep;
                    done = cur.t >= ntile;
                }
            });
        }
        // the repeated fetches still target these registers: keep them out of other use until they have landed
        sfor<0, D + 1>([&](auto ic) { hl_wait<0>(ring[decltype(ic)::value]); });
    }
    // lanes with equal q hold the same classes: sum over the 16 pixel lanes, then over the four waves
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        float vals[4] = {sb[e], si[e], sp[e], st[e]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = vals[k];
            x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
            if (r16 == 0) red[wv][q + 4 * e][k] = x;
        }
    }
    __syncthreads();
    if (threadIdx.x < Ccls * 4) {
        const int c = threadIdx.x >> 2, k = threadIdx.x & 3;
        part[((long long)blockIdx.x * Ccls + c) * 4 + k] = red[0][c][k] + red[1][c][k] + red[2][c][k] + red[3][c][k];
    }
}

#ifndef HL_BWD_AHEAD
#define HL_BWD_AHEAD 1            // 2 measured the same (0.96 vs 0.95 ms at batch 64 @512^2): the kernel is not waiting for HBM latency
#endif
// MODE 0: da = W^T dlogits is stored, dW / db partial sums go to `part`.
// MODE 1 (BNA): the first pass of dec1.3's BatchNorm backward instead of the da store: g = da * [relu(bn(y)) > 0] (da rounded to
//         bf16 as MODE 0 stores it), per-channel sum g and invstd * sum g * (y - mean) -> bn_part[block][64][2]; dW / db as MODE 0.
//         One wave per SIMD: the dW accumulators and the BatchNorm sums together need ~300 registers.
// MODE 2 (BNA): da is formed again and dec1.3's BatchNorm backward is applied to it: dy = scale * (g - c1 - xhat * c2) is stored
//         where MODE 0 stores da (what fea_bn_relu_bwd_apply would write from da and y); per-channel sum dy -> bn_part[block][64][2]
//         slot 0 (the conv bias gradient); no dW / db.  With MODE 1 and 2 the gradient da (2 GB at batch 64 x 512^2) is never
//         stored: three reads of y and one store of dy replace store da, read da + y twice, store dy.
struct HlBnBwd { const float *mu, *is, *coef; float* part; };
template <int NC, int D, bool BNA, int MODE>       // classes the dW accumulators are sized for (4, 6 or 8); tiles of read-ahead
__global__ __launch_bounds__(256, 2) void head_loss_bwd_mfma_kernel(const __bf16* __restrict__ a, int lda, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, const uint8_t* __restrict__ tgt,
                                                                    const double* __restrict__ sums, float* __restrict__ loss,
                                                                    __bf16* __restrict__ da, int ldda, float* __restrict__ part,
                                                                    float gscale, int ntile, int HW, int N, int Ccls,
                                                                    const float* __restrict__ bn_sc, const float* __restrict__ bn_sh,
                                                                    const HlBnBwd bb) {
    static_assert(MODE == 0 || BNA, "the BatchNorm-backward modes read the pre-BN tensor");
    constexpr bool GW = MODE != 2;                   // dW / db accumulation
    constexpr int NE = (NC + 3) / 4;
    __shared__ float red[4][NC * 65];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, q = lane >> 4;
    const long long total = (long long)N * Ccls * HW;
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss) {
        double bce = 0.0, dice = 0.0;
        for (int c = 0; c < Ccls; ++c) {
            bce += sums[c * 4];
            dice += (2.0 * sums[c * 4 + 1] + 1.0) / (sums[c * 4 + 2] + sums[c * 4 + 3] + 1.0);
        }
        loss[0] = (float)(bce / (double)total + 1.0 - dice / Ccls);
    }
    HlFrag f;
    hl_weights(f, w, bias, Ccls, r16, q);
    HlBn bn;
    hl_bn_load<BNA>(bn, bn_sc, bn_sh, q);
    // BatchNorm-backward modes: this lane's 16 channels (x0: 8q .., x1: 32 + 8q ..)
    float bmu[16], bis[16], bc1[16], bc2[16], s0[16], s1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ch = (j >> 3) * 32 + q * 8 + (j & 7);
        bmu[j] = MODE != 0 ? bb.mu[ch] : 0.f;
        bis[j] = MODE == 2 ? bb.is[ch] : 0.f;
        bc1[j] = MODE == 2 ? bb.coef[2 * ch] : 0.f; bc2[j] = MODE == 2 ? bb.coef[2 * ch + 1] : 0.f;
        s0[j] = 0.f; s1[j] = 0.f;
    }
    // GEMM 2 A operand, row block mb: lane (row = lane % 16, q): k-slot j < 4 <-> class q + 4j, slots 4..7 zero.  Row i of
    // block mb is channel 32 (mb / 2) + 8 (i / 4) + 4 (mb % 2) + i % 4: the accumulators of a lane over a block pair are
    // eight consecutive channels.
    u32x4 w2h[4], w2l[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int ch = 32 * (mb >> 1) + 8 * (r16 >> 2) + 4 * (mb & 1) + (r16 & 3);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = q + 4 * j < Ccls ? w[(q + 4 * j) * 64 + ch] : 0.f;
        w2h[mb] = u32x4{hl_pack(v[0], v[1]), hl_pack(v[2], v[3]), 0u, 0u};
        w2l[mb] = u32x4{hl_pack(hl_lo(v[0]), hl_lo(v[1])), hl_pack(hl_lo(v[2]), hl_lo(v[3])), 0u, 0u};
    }
    float cf0[NE], cf1[NE];                           // per class of this lane: 1/(S+1), (2I+1)/(S+1)
    bool have[NE];
    int lc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int c = q + 4 * e;
        have[e] = c < Ccls;
        const double I = have[e] ? sums[c * 4 + 1] : 0.0, S = have[e] ? sums[c * 4 + 2] + sums[c * 4 + 3] : 0.0;
        cf0[e] = (float)(1.0 / (S + 1.0));
        cf1[e] = (float)((2.0 * I + 1.0) / (S + 1.0));
        lc[e] = (have[e] ? c : q < Ccls ? q : 0) * HW + r16;
    }
    const float itot = gscale / (float)total, ic = gscale / (float)Ccls;
    // dW on the matrix pipe: dW^T[channel][class] = sum over pixels act^T[channel][pixel] d[pixel][class], four 16-channel M
    // tiles, k = the tile's 16 pixels (k-slots 16..31 carry zeros in B).  Both operands need the pixel index as k, i.e. a
    // transpose of what the lanes hold (lane = pixel): the activations go through a per-wave LDS tile and come back through
    // the transposing read, the (<= 8 x 16) dlogits through a small fp32 table.  d enters as hi + lo bf16 like in the da product.
    // (On the VALU -- 16 channels x <= 8 classes of fmas per lane and tile, 96 accumulators -- this was 40 % of the kernel's
    // instructions and kept the BatchNorm-backward mode from fitting two waves per SIMD.)
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
    // Bank layout (round 3: SQ_LDS_BANK_CONFLICT was 77 % of this kernel's LDS cycles, the LDS busy for half of its time, with pixel
    // rows 128 B apart: every row began at bank 0).  Rows are APITCH = 160 B apart: the four rows x 32 B a 16-lane group of the
    // transposing read touches cover all 32 banks once, the eight rows of a ds_write_b128 group meet two by two (an XOR swizzle
    // that clears both costs the kernel a register it does not have: 256, one spilled).  The d table's rows are 20 floats apart,
    // so the eight lanes of its 16-byte reads cover all banks too.
    constexpr int APITCH = 160, DTP = 20;
    __shared__ __attribute__((aligned(128))) char tsp[GW ? 4 : 1][16 * APITCH + 16 * DTP * 4];     // per wave: [16 px][APITCH] activations, [16 classes][DTP] fp32 d
    char* const my = tsp[GW ? wv : 0];
    float* const dt = reinterpret_cast<float*>(my + 16 * APITCH);
    if constexpr (GW) {
        for (int i = lane; i < 16 * DTP; i += 64) dt[i] = 0.f;        // rows of absent classes stay zero
    }
    f32x4 G[4];
    float gb[NE];
#pragma unroll
    for (int t = 0; t < 4; ++t) G[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < NE; ++e) gb[e] = 0.f;

    const int nw = gridDim.x * 4;
    HlWalk cur, ahead;                                // the tile being multiplied; the next tile to ask for
    cur.init(blockIdx.x * 4 + wv, nw, HW);
    ahead = cur;
    const __bf16* xc = a + ((long long)cur.t * 16 + r16) * lda + q * 8;      // this lane's row of `cur` ...
    const __bf16* xa = xc;                                                   // ... and of `ahead`
    __bf16* dp = da + ((long long)cur.t * 16 + r16) * ldda + q * 8;
    const long long xstep = (long long)nw * 16 * lda, dstep = (long long)nw * 16 * ldda;
    const int CHW = Ccls * HW;
    // D tiles of read-ahead (two waves per SIMD do not hide HBM latency) on a ring of D + 1 register sets, no copies.  Past
    // the last tile the fetch asks for the current tile again: every tile then has the same number of loads behind it.
    HlTile<NE> ring[D + 1];
    auto fetch = [&](HlTile<NE>& d) {
        const bool more = ahead.t < ntile;
        const HlWalk& k = more ? ahead : cur;
        hl_fetch<NE, false>(d, more ? xa : xc, tgt, (long long)k.n * CHW + k.hw, lc);
        xa += xstep; ahead.next(HW);                  // (past the end neither is dereferenced)
    };
    // (always_inline: in MODE 1 the body is past the inliner's threshold, and a called lambda keeps every captured array in scratch)
    auto tile = [&](const HlTile<NE>& c) __attribute__((always_inline)) {
        u32x4 x0 = c.x0, x1 = c.x1;
        float yf[16], act[16];
        if constexpr (MODE != 0) hl_bn_act_keep(bn, x0, x1, yf, act);
        else hl_bn_act<BNA>(bn, x0, x1);
        const f32x4 v = hl_logits(f, x0, x1);
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float tt = c.t[e] ? 1.f : 0.f;
            const float ex = __expf(-fabsf(v[e]));
            const float r = __builtin_amdgcn_rcpf(1.f + ex);
            const float pr = v[e] >= 0.f ? r : ex * r;
            const float ddice = (2.f * tt - cf1[e]) * cf0[e];
            const float de = (pr - tt) * itot - ddice * pr * (1.f - pr) * ic;
            d[e] = have[e] ? de : 0.f;
            gb[e] += d[e];
        }
        // da = W^T d on the matrix pipe
        const u32x4 bh = {hl_pack(d[0], d[1]), hl_pack(d[2], d[3]), 0u, 0u};
        const u32x4 bl = {hl_pack(hl_lo(d[0]), hl_lo(d[1])), hl_pack(hl_lo(d[2]), hl_lo(d[3])), 0u, 0u};
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
            f32x4 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mb = 2 * mp + h;
                o[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                o[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2h[mb]), __builtin_bit_cast(bf16x8, bh), o[h], 0, 0, 0);
                o[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2h[mb]), __builtin_bit_cast(bf16x8, bl), o[h], 0, 0, 0);
                o[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2l[mb]), __builtin_bit_cast(bf16x8, bh), o[h], 0, 0, 0);
            }
            const u32x4 dav = u32x4{hl_pack(o[0][0], o[0][1]), hl_pack(o[0][2], o[0][3]), hl_pack(o[1][0], o[1][1]), hl_pack(o[1][2], o[1][3])};
            if constexpr (MODE == 0) *reinterpret_cast<u32x4*>(dp + mp * 32) = dav;
            else {
                // da as MODE 0 would have stored it (bf16), then the BatchNorm backward of bn_relu_bwd_kernel (elementwise.hip)
                const bf16x8 dab = __builtin_bit_cast(bf16x8, dav);
                float dy[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = mp * 8 + e;
                    const float gg = act[j] > 0.f ? (float)dab[e] : 0.f;
                    if constexpr (MODE == 1) { s0[j] += gg; s1[j] = __builtin_fmaf(gg, yf[j] - bmu[j], s1[j]); }
                    else {
                        const float xh = (yf[j] - bmu[j]) * bis[j];
                        dy[e] = bn.sc[j] * (gg - bc1[j] - xh * bc2[j]);
                        s0[j] += dy[e];
                    }
                }
                if constexpr (MODE == 2)
                    *reinterpret_cast<u32x4*>(dp + mp * 32) = u32x4{hl_pack(dy[0], dy[1]), hl_pack(dy[2], dy[3]), hl_pack(dy[4], dy[5]), hl_pack(dy[6], dy[7])};
            }
        }
        dp += dstep;
        if constexpr (!GW) return;
        // dW (see above): lane (pixel r16, q) writes its 16 activations and its dlogits, then reads the transposed operands
        *reinterpret_cast<u32x4*>(my + r16 * APITCH + q * 16) = x0;
        *reinterpret_cast<u32x4*>(my + r16 * APITCH + 64 + q * 16) = x1;
#pragma unroll
        for (int e = 0; e < NE; ++e) dt[(q + 4 * e) * DTP + r16] = d[e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // B: lane (class n = r16, k-group q): d[pixel 8q .. 8q+7][n] for q < 2, zeros for the padding k-slots
        float dv[8];
        {
            const f32x4 lo4 = *reinterpret_cast<const f32x4*>(dt + r16 * DTP + (q & 1) * 8);
            const f32x4 hi4 = *reinterpret_cast<const f32x4*>(dt + r16 * DTP + (q & 1) * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { dv[j] = q < 2 ? lo4[j] : 0.f; dv[4 + j] = q < 2 ? hi4[j] : 0.f; }
        }
        const u32x4 dbh = {hl_pack(dv[0], dv[1]), hl_pack(dv[2], dv[3]), hl_pack(dv[4], dv[5]), hl_pack(dv[6], dv[7])};
        const u32x4 dbl = {hl_pack(hl_lo(dv[0]), hl_lo(dv[1])), hl_pack(hl_lo(dv[2]), hl_lo(dv[3])),
                           hl_pack(hl_lo(dv[4]), hl_lo(dv[5])), hl_pack(hl_lo(dv[6]), hl_lo(dv[7]))};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // A: lane (channel 16 t + r16, k-group q): pixels 8 (q & 1) .. + 7 (lanes of the padding k-slots re-read real pixels:
            // their products meet zeros).  Transposing read: lane fetches 4 channels of pixel 8 (q & 1) + 4 u + r16 / 4.
            s16x4 h2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                h2[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4_t*)(my + (8 * (q & 1) + 4 * u + (r16 >> 2)) * APITCH + (16 * t + 4 * (r16 & 3)) * 2));
            const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h2[0], h2[1], 0, 1, 2, 3, 4, 5, 6, 7));
            G[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, dbh), G[t], 0, 0, 0);
            G[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, dbl), G[t], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                 // the next tile's writes come after these reads
    };
    // Behind a tile's loads when it is waited for: D fetches of 2 + NE loads and the two da stores of each of the D tiles before
    // it.  The first D tiles have fewer stores behind them: they are waited for outright.
    if (cur.t < ntile) {
        sfor<0, D>([&](auto ic) { fetch(ring[decltype(ic)::value]); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool done = false;
        while (!done) {
            sfor<0, D + 1>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                if (!done) {
                    fetch(ring[(s_ + D) % (D + 1)]);
                    hl_wait<D * ((MODE == 1 ? 2 : 4) + NE)>(ring[s_]);
                    tile(ring[s_]);
                    cur.next(HW); xc += xstep;
                    done = cur.t >= ntile;
                }
            });
        }
        // the repeated fetches still target these registers: keep them out of other use until they have landed
        sfor<0, D + 1>([&](auto ic) { hl_wait<0>(ring[decltype(ic)::value]); });
    }
    if constexpr (MODE != 0) {
        // per-channel sums of the BatchNorm-backward modes: over the 16 pixel lanes, then over the four waves
        __shared__ float redbn[4][128];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float x = s0[j], z = s1[j];
            x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
            z += __shfl_xor(z, 1); z += __shfl_xor(z, 2); z += __shfl_xor(z, 4); z += __shfl_xor(z, 8);
            if (r16 == 0) {
                const int ch = (j >> 3) * 32 + q * 8 + (j & 7);
                redbn[wv][2 * ch] = x; redbn[wv][2 * ch + 1] = z;
            }
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int ch = threadIdx.x >> 1;
            float t = (redbn[0][threadIdx.x] + redbn[1][threadIdx.x]) + (redbn[2][threadIdx.x] + redbn[3][threadIdx.x]);
            if (MODE == 1 && (threadIdx.x & 1)) t *= bb.is[ch];          // sum g * xhat = invstd * sum g * (y - mean)
            bb.part[(long long)blockIdx.x * 128 + threadIdx.x] = t;
        }
        if constexpr (!GW) return;
        __syncthreads();
    }
    // dW: lane (class r16, q) holds channels 16 t + 4 q + e of its class (already summed over the pixels); then over the four waves
    if (r16 < NC) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wv][r16 * 65 + 16 * t + 4 * q + e] = G[t][e];
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        float x = gb[e];
        x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
        if (r16 == 0 && q + 4 * e < NC) red[wv][(q + 4 * e) * 65 + 64] = x;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Ccls * 65; i += 256)
        part[(long long)blockIdx.x * (Ccls * 65) + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// ---------------------------------------------------------------------------
// fea_bce_dice_sums for callers that cannot allocate (HIP graph capture, tight serving / evaluation
// loops, harnesses that forbid hipMalloc): the per-block partial sums live in a fixed device array of
// this code object instead of a caller workspace.  Every GPU holds its own copy of the array at its
// own address; the address is looked up once per device (first call on that device, with the device
// current) and kept.  Host code never uses the array's NAME as a pointer: it is a host-side stand-in.
#define FEA_STATIC_LOSS_BLOCKS 65536                         // 1 MiB of fp32 partials per GPU, held until the process ends
static __device__ __align__(256) float scratch[FEA_STATIC_LOSS_BLOCKS * 4];

#define FEA_MAX_DEVICES 64
static float* g_scratch_ptr[FEA_MAX_DEVICES];                // per device, resolved lazily
static std::mutex g_scratch_mu;

static int static_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    FEA_CHECK(e == hipSuccess && dev >= 0 && dev < FEA_MAX_DEVICES, FEA_E_LAUNCH, "fea_bce_dice_sums_static: hipGetDevice: %s (device %d)",
              hipGetErrorString(e), dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        FEA_CHECK(e == hipSuccess && p, FEA_E_LAUNCH, "fea_bce_dice_sums_static: hipGetSymbolAddress on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return FEA_OK;
}

extern "C" int fea_bce_dice_sums_static(const float* logits, const uint8_t* target, double* sums,
                                        int N, int Ccls, long long HW, void* stream) {
    FEA_CHECK(logits && target && sums && N > 0 && Ccls > 0 && HW > 0, FEA_E_BADARG, "fea_bce_dice_sums_static: bad args");
    int chunks; long long cl; loss_plan(HW, chunks, cl);
    const long long nblk = (long long)N * Ccls * chunks;
    FEA_CHECK(nblk <= FEA_STATIC_LOSS_BLOCKS, FEA_E_WORKSPACE,
              "fea_bce_dice_sums_static: N*classes*ceil(HW/8192) = %lld partial-sum blocks exceed the built-in scratch of %d "
              "(use fea_bce_dice_sums with a caller workspace)", nblk, FEA_STATIC_LOSS_BLOCKS);
    float* part = nullptr;
    int rc = static_scratch(&part);
    if (rc) return rc;
    // every block [0, nblk) of the scratch is overwritten before the finalize kernel reads it: nothing to clear
    return fea_bce_dice_sums(logits, target, sums, part, N, Ccls, HW, stream);
}

extern "C" int fea_bce_dice_grad(const float* logits, const uint8_t* target, const double* sums, float* loss, float* dlogits,
                                 float gscale, int N, int Ccls, long long HW, void* stream) {
    FEA_CHECK(logits && target && sums && (loss || dlogits) && N > 0 && Ccls > 0 && HW > 0, FEA_E_BADARG, "fea_bce_dice_grad: bad args");
    const long long total = (long long)N * Ccls * HW;
    const bool vec = HW % 4 == 0 && (uintptr_t)logits % 16 == 0 && (uintptr_t)target % 4 == 0 && (!dlogits || (uintptr_t)dlogits % 16 == 0);
    const long long per = vec ? 1024 : 256;
    const long long nb = dlogits ? (total + per - 1) / per : 1;
    if (vec) hipLaunchKernelGGL(bce_dice_grad_kernel<4>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, logits, target, sums, loss, dlogits,
                                gscale, N, Ccls, HW);
    else     hipLaunchKernelGGL(bce_dice_grad_kernel<1>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, logits, target, sums, loss, dlogits,
                                gscale, N, Ccls, HW);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_mask_iou(const float* logits, const uint8_t* target, uint8_t* mask, long long* inter, long long* uni,
                            int N, int Ccls, long long HW, void* stream) {
    // target == NULL: threshold only (mask required, the counts are left alone and may be NULL)
    FEA_CHECK(logits && (target ? (inter && uni) : mask != nullptr) && N > 0 && Ccls > 0 && HW > 0, FEA_E_BADARG, "fea_mask_iou: bad args");
    int chunks; long long cl; loss_plan(HW, chunks, cl);
    hipLaunchKernelGGL(mask_iou_kernel, dim3(N * Ccls * chunks), dim3(256), 0, (hipStream_t)stream, logits, target, mask,
                       (unsigned long long*)inter, (unsigned long long*)uni, HW, Ccls, chunks, cl);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_binarise(const uint8_t* grey, uint8_t* out, long long n, void* stream) {
    FEA_CHECK(grey && out && n > 0, FEA_E_BADARG, "fea_binarise: bad args");
    hipLaunchKernelGGL(binarise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grey, out, n);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

// ---- fused head + loss ---------------------------------------------------------------------
extern "C" int fea_head_loss_supported(int H, int W, int Ccls) { return ((long long)H * W) % 4 == 0 && Ccls >= 1 && Ccls <= 8; }

static int head_loss_sums_impl(const char* who, const void* a, int lda, const float* bn_sc, const float* bn_sh, const float* w, const float* bias,
                               const uint8_t* target, float* logits, double* sums, float* part, int N, int H, int W, int Cin, int Ccls,
                               int dtype, void* stream) {
    FEA_CHECK(a && w && bias && target && sums && part && N > 0 && H > 0 && W > 0, FEA_E_BADARG, "%s: bad args", who);
    FEA_CHECK(Cin == 64 && lda >= 64 && lda % 8 == 0, FEA_E_UNSUPPORTED_SHAPE, "%s: Cin=%d lda=%d", who, Cin, lda);
    FEA_CHECK(fea_head_loss_supported(H, W, Ccls), FEA_E_UNSUPPORTED_SHAPE, "%s: needs H*W %% 4 == 0 and 1..8 classes (use fea_head_fwd + fea_bce_dice_sums)", who);
    FEA_CHECK((uintptr_t)target % 4 == 0 && (!logits || (uintptr_t)logits % 16 == 0), FEA_E_BADARG, "%s: alignment", who);
    const long long npix = (long long)N * H * W, HW = (long long)H * W;
    const int nblk = head_blocks(npix);
    hipStream_t s = (hipStream_t)stream;
    const bool mfma = dtype == FEA_BF16 && HW % 16 == 0 && npix * Ccls < (1ll << 31);       // 32-bit tile positions and target offsets
    FEA_CHECK(!bn_sc || (mfma && bn_sh), FEA_E_UNSUPPORTED_SHAPE, "%s: the fused BatchNorm + ReLU needs bf16 and H*W %% 16 == 0", who);
#define HLS_LAUNCH(NEV, BNV) hipLaunchKernelGGL((head_loss_sums_mfma_kernel<NEV, BNV>), dim3(nblk), dim3(256), 0, s, (const __bf16*)a, lda, w, bias, target, logits, part, (int)(npix / 16), (int)HW, Ccls, bn_sc, bn_sh)
    if (mfma && Ccls <= 4) { if (bn_sc) HLS_LAUNCH(1, true); else HLS_LAUNCH(1, false); }
    else if (mfma) { if (bn_sc) HLS_LAUNCH(2, true); else HLS_LAUNCH(2, false); }
#undef HLS_LAUNCH
    else if (dtype == FEA_BF16) hipLaunchKernelGGL((head_loss_sums_kernel<__bf16>), dim3(nblk), dim3(256), 0, s, (const __bf16*)a, lda, w, bias, target, logits, part, npix / 4, HW, Ccls);
    else if (dtype == FEA_F32) hipLaunchKernelGGL((head_loss_sums_kernel<float>), dim3(nblk), dim3(256), 0, s, (const float*)a, lda, w, bias, target, logits, part, npix / 4, HW, Ccls);
    else { fea_set_error("%s: bad dtype", who); return FEA_E_BADARG; }
    FEA_LAUNCH_CHECK();
    // part[blk][class][4]: the layout bce_dice_sums_finalize_kernel reads with "N" = blocks and one chunk per plane
    hipLaunchKernelGGL(bce_dice_sums_finalize_kernel, dim3(Ccls), dim3(256), 0, s, part, sums, nblk, Ccls, 1);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_head_loss_sums(const void* a, int lda, const float* w, const float* bias, const uint8_t* target,
                                  float* logits, double* sums, float* part, int N, int H, int W, int Cin, int Ccls,
                                  int dtype, void* stream) {
    return head_loss_sums_impl("fea_head_loss_sums", a, lda, nullptr, nullptr, w, bias, target, logits, sums, part, N, H, W, Cin, Ccls, dtype, stream);
}
// 1 when fea_head_loss_sums_bn / fea_head_loss_bwd_bn serve this problem (the matrix-pipe form of the fused head + loss kernels)
extern "C" int fea_head_loss_bn_supported(int N, int H, int W, int Ccls, int dtype) {
    const long long HW = (long long)H * W;
    return dtype == FEA_BF16 && N > 0 && fea_head_loss_supported(H, W, Ccls) && Ccls <= 6 && HW % 16 == 0 && (long long)N * HW * Ccls < (1ll << 31);
}
extern "C" int fea_head_loss_sums_bn(const void* y, int ldy, const float* bn_scale, const float* bn_shift, const float* w, const float* bias,
                                     const uint8_t* target, float* logits, double* sums, float* part, int N, int H, int W, int Cin,
                                     int Ccls, int dtype, void* stream) {
    FEA_CHECK(bn_scale && bn_shift, FEA_E_BADARG, "fea_head_loss_sums_bn: null coefficients");
    return head_loss_sums_impl("fea_head_loss_sums_bn", y, ldy, bn_scale, bn_shift, w, bias, target, logits, sums, part, N, H, W, Cin, Ccls, dtype, stream);
}

static int head_loss_bwd_impl(const char* who, const void* a, int lda, const float* bn_sc, const float* bn_sh, const float* w, const float* bias,
                              const uint8_t* target, const double* sums, float* loss, void* da, int ldda, float* part, float gscale,
                              int N, int H, int W, int Cin, int Ccls, int dtype, void* stream) {
    FEA_CHECK(a && w && bias && target && sums && da && part && N > 0 && H > 0 && W > 0, FEA_E_BADARG, "%s: bad args", who);
    FEA_CHECK(Cin == 64 && lda >= 64 && lda % 8 == 0 && ldda >= 64 && ldda % 8 == 0, FEA_E_UNSUPPORTED_SHAPE, "%s: Cin=%d lda=%d ldda=%d", who, Cin, lda, ldda);
    FEA_CHECK(fea_head_loss_supported(H, W, Ccls), FEA_E_UNSUPPORTED_SHAPE, "%s: needs H*W %% 4 == 0 and 1..8 classes", who);
    FEA_CHECK((uintptr_t)target % 4 == 0, FEA_E_BADARG, "%s: alignment", who);
    const long long npix = (long long)N * H * W, HW = (long long)H * W;
    const int nblk = head_blocks(npix);
    long long ppb = (npix + nblk - 1) / nblk;
    ppb = (ppb + 63) / 64 * 64;
    hipStream_t s = (hipStream_t)stream;
    const bool mfma = dtype == FEA_BF16 && HW % 16 == 0 && npix * Ccls < (1ll << 31);       // as in fea_head_loss_sums
    FEA_CHECK(!bn_sc || (mfma && bn_sh), FEA_E_UNSUPPORTED_SHAPE, "%s: the fused BatchNorm + ReLU needs bf16 and H*W %% 16 == 0", who);
#define HLB_LAUNCH(NCV, DV, BNV) hipLaunchKernelGGL((head_loss_bwd_mfma_kernel<NCV, DV, BNV, 0>), dim3(nblk), dim3(256), 0, s, (const __bf16*)a, lda, w, bias, target, sums, loss, (__bf16*)da, ldda, part, gscale, (int)(npix / 16), (int)HW, N, Ccls, bn_sc, bn_sh, HlBnBwd{})
    if (mfma && Ccls <= 4) { if (bn_sc) HLB_LAUNCH(4, HL_BWD_AHEAD, true); else HLB_LAUNCH(4, HL_BWD_AHEAD, false); }
    else if (mfma && Ccls <= 6) { if (bn_sc) HLB_LAUNCH(6, HL_BWD_AHEAD, true); else HLB_LAUNCH(6, HL_BWD_AHEAD, false); }
    else if (mfma) {
        // (7-8 classes with the fused BatchNorm: the 128 dW accumulators + 32 coefficients do not fit two waves per SIMD)
        FEA_CHECK(!bn_sc, FEA_E_UNSUPPORTED_SHAPE, "%s: the fused BatchNorm + ReLU serves up to 6 classes", who);
        HLB_LAUNCH(8, 1, false);
    }
#undef HLB_LAUNCH
    else if (dtype == FEA_BF16) hipLaunchKernelGGL((head_loss_bwd_kernel<__bf16>), dim3(nblk), dim3(256), 0, s, (const __bf16*)a, lda, w, bias, target, sums, loss, (__bf16*)da, ldda, part, gscale, npix, HW, N, Ccls, ppb);
    else if (dtype == FEA_F32) hipLaunchKernelGGL((head_loss_bwd_kernel<float>), dim3(nblk), dim3(256), 0, s, (const float*)a, lda, w, bias, target, sums, loss, (float*)da, ldda, part, gscale, npix, HW, N, Ccls, ppb);
    else { fea_set_error("%s: bad dtype", who); return FEA_E_BADARG; }
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_head_loss_bwd(const void* a, int lda, const float* w, const float* bias, const uint8_t* target,
                                 const double* sums, float* loss, void* da, int ldda, float* part, float gscale,
                                 int N, int H, int W, int Cin, int Ccls, int dtype, void* stream) {
    return head_loss_bwd_impl("fea_head_loss_bwd", a, lda, nullptr, nullptr, w, bias, target, sums, loss, da, ldda, part, gscale, N, H, W, Cin, Ccls, dtype, stream);
}
extern "C" int fea_head_loss_bwd_bn(const void* y, int ldy, const float* bn_scale, const float* bn_shift, const float* w, const float* bias,
                                    const uint8_t* target, const double* sums, float* loss, void* da, int ldda, float* part, float gscale,
                                    int N, int H, int W, int Cin, int Ccls, int dtype, void* stream) {
    FEA_CHECK(bn_scale && bn_shift, FEA_E_BADARG, "fea_head_loss_bwd_bn: null coefficients");
    return head_loss_bwd_impl("fea_head_loss_bwd_bn", y, ldy, bn_scale, bn_shift, w, bias, target, sums, loss, da, ldda, part, gscale, N, H, W, Cin, Ccls, dtype, stream);
}

#ifndef HL_RED_AHEAD
#define HL_RED_AHEAD 1            // tiles of read-ahead of the MODE 1 kernel
#endif
// dec1.3's BatchNorm backward inside the fused head + loss backward (MODE 1 / MODE 2 of head_loss_bwd_mfma_kernel): two passes over
// the pre-BN tensor y replace {store da, fea_bn_relu_bwd_reduce, fea_bn_relu_bwd_apply}.  bn_part: [fea_head_bwd_blocks()][64][2].
static int head_loss_bnbwd_check(const char* who, const void* y, int ldy, const float* w, const float* bias, const uint8_t* target,
                                 const double* sums, int N, int H, int W, int Cin, int Ccls, int dtype) {
    FEA_CHECK(y && w && bias && target && sums && N > 0 && H > 0 && W > 0, FEA_E_BADARG, "%s: bad args", who);
    FEA_CHECK(Cin == 64 && ldy >= 64 && ldy % 8 == 0, FEA_E_UNSUPPORTED_SHAPE, "%s: Cin=%d ldy=%d", who, Cin, ldy);
    FEA_CHECK(fea_head_loss_bn_supported(N, H, W, Ccls, dtype), FEA_E_UNSUPPORTED_SHAPE, "%s: needs bf16, H*W %% 16 == 0, 1..6 classes", who);
    FEA_CHECK((uintptr_t)target % 4 == 0, FEA_E_BADARG, "%s: alignment", who);
    return FEA_OK;
}
extern "C" int fea_head_loss_bwd_bnred(const void* y, int ldy, const float* bn_scale, const float* bn_shift, const float* save_mean,
                                       const float* save_invstd, const float* w, const float* bias, const uint8_t* target,
                                       const double* sums, float* loss, float* part, float* bn_part, float gscale,
                                       int N, int H, int W, int Cin, int Ccls, int dtype, void* stream) {
    const char* who = "fea_head_loss_bwd_bnred";
    int rc = head_loss_bnbwd_check(who, y, ldy, w, bias, target, sums, N, H, W, Cin, Ccls, dtype);
    if (rc) return rc;
    FEA_CHECK(bn_scale && bn_shift && save_mean && save_invstd && part && bn_part, FEA_E_BADARG, "%s: null pointer", who);
    const long long npix = (long long)N * H * W, HW = (long long)H * W;
    const int nblk = head_blocks(npix);
    hipStream_t s = (hipStream_t)stream;
    const HlBnBwd bb{save_mean, save_invstd, nullptr, bn_part};
#define HLR_LAUNCH(NCV) hipLaunchKernelGGL((head_loss_bwd_mfma_kernel<NCV, HL_RED_AHEAD, true, 1>), dim3(nblk), dim3(256), 0, s, (const __bf16*)y, ldy, w, bias, target, sums, loss, (__bf16*)nullptr, 64, part, gscale, (int)(npix / 16), (int)HW, N, Ccls, bn_scale, bn_shift, bb)
    if (Ccls <= 4) HLR_LAUNCH(4); else HLR_LAUNCH(6);
#undef HLR_LAUNCH
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}
extern "C" int fea_head_loss_bwd_bnapply(const void* y, int ldy, const float* bn_scale, const float* bn_shift, const float* save_mean,
                                         const float* save_invstd, const float* coef, const float* w, const float* bias,
                                         const uint8_t* target, const double* sums, void* dy, int lddy, float* bn_part, float gscale,
                                         int N, int H, int W, int Cin, int Ccls, int dtype, void* stream) {
    const char* who = "fea_head_loss_bwd_bnapply";
    int rc = head_loss_bnbwd_check(who, y, ldy, w, bias, target, sums, N, H, W, Cin, Ccls, dtype);
    if (rc) return rc;
    FEA_CHECK(bn_scale && bn_shift && save_mean && save_invstd && coef && dy && bn_part && lddy >= 64 && lddy % 8 == 0, FEA_E_BADARG, "%s: bad args", who);
    const long long npix = (long long)N * H * W, HW = (long long)H * W;
    const int nblk = head_blocks(npix);
    hipStream_t s = (hipStream_t)stream;
    const HlBnBwd bb{save_mean, save_invstd, coef, bn_part};
#define HLA_LAUNCH(NCV) hipLaunchKernelGGL((head_loss_bwd_mfma_kernel<NCV, 1, true, 2>), dim3(nblk), dim3(256), 0, s, (const __bf16*)y, ldy, w, bias, target, sums, (float*)nullptr, (__bf16*)dy, lddy, (float*)nullptr, gscale, (int)(npix / 16), (int)HW, N, Ccls, bn_scale, bn_shift, bb)
    if (Ccls <= 4) HLA_LAUNCH(4); else HLA_LAUNCH(6);
#undef HLA_LAUNCH
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}
