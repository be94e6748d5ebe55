// Bandwidth-bound kernels: layout/packing, BatchNorm(+ReLU+MaxPool) forward and
// backward, per-channel reductions.  All activation accesses are 16 bytes per
// lane with the channel group as the fastest-varying index, so a wave reads
// and writes whole contiguous pixel rows.
#include "common.h"

template <typename T> struct Vec { static constexpr int N = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void unpack(const u32x4& v, float* f);
template <> __device__ __forceinline__ void unpack<float>(const u32x4& v, float* f) {
    const f32x4 t = __builtin_bit_cast(f32x4, v);      // whole-vector cast (see common.h note)
    f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
}
template <> __device__ __forceinline__ void unpack<__bf16>(const u32x4& v, float* f) {
    const bf16x8 b = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)b[j];
}
template <typename T> __device__ __forceinline__ u32x4 pack(const float* f);
template <> __device__ __forceinline__ u32x4 pack<float>(const float* f) {
    f32x4 t;
    t.x = f[0]; t.y = f[1]; t.z = f[2]; t.w = f[3];
    return __builtin_bit_cast(u32x4, t);
}
template <> __device__ __forceinline__ u32x4 pack<__bf16>(const float* f) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)f[j];
    return __builtin_bit_cast(u32x4, b);
}

// The one definition of the BN+ReLU activation, shared by forward, pooling and
// the backward mask / arg-max recomputation so that they agree bit for bit.
__device__ __forceinline__ float bn_act(float y, float sc, float sh) { return fmaxf(__builtin_fmaf(y, sc, sh), 0.f); }

#define GRID1D(n, bs) dim3((unsigned)(((n) + (bs) - 1) / (bs)))

// ---------------------------------------------------------------------------
// layout
// ---------------------------------------------------------------------------
template <typename T>
__global__ void im2col3x3_kernel(const float* __restrict__ x, T* __restrict__ out, int N, int C, int H, int W, int Kpad) {
    constexpr int V = Vec<T>::N;
    const int gpp = Kpad / V;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * H * W * gpp;
    if (i >= total) return;
    const int grp = (int)(i % gpp);
    long long pix = i / gpp;
    const int xx = (int)(pix % W); pix /= W;
    const int yy = (int)(pix % H);
    const int n = (int)(pix / H);
    float f[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int k = grp * V + j;
        float v = 0.f;
        if (k < 9 * C) {
            const int tap = k / C, c = k - tap * C;
            const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
            if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[(((long long)n * C + c) * H + sy) * W + sx];
        }
        f[j] = v;
    }
    *reinterpret_cast<u32x4*>(out + i * V) = pack<T>(f);
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, int ld, float* __restrict__ out, int N, int C, int H, int W) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * C * H * W;
    if (i >= total) return;
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const int c = (int)((i / ((long long)W * H)) % C);
    const int n = (int)(i / ((long long)W * H * C));
    out[i] = to_f32(x[(((long long)n * H + yy) * W + xx) * ld + c]);
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ out, int ld, int N, int C, int H, int W) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * C * H * W;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int xx = (int)((i / C) % W);
    const int yy = (int)((i / ((long long)C * W)) % H);
    const int n = (int)(i / ((long long)C * W * H));
    out[(((long long)n * H + yy) * W + xx) * ld + c] = from_f32<T>(x[(((long long)n * C + c) * H + yy) * W + xx]);
}

// packed conv3x3 weight.  kc = chunk*9 + tap, row = GEMM N index, kk = channel inside the chunk.
//   row-major:      out[kc][row][kk]                         (128-byte rows, staged through LDS)
//   fragment order (bf16 and rows % 128 != 0, see fea_conv3x3_frag_pack):
//         out[kc][row/16][kk/32][lane = ((kk/8)&3)*16 + row%16][kk%8]
//         = the 16x16x32 MFMA B fragments in the order the conv kernel's waves load them
//         (one contiguous KB per wave-instruction, global -> VGPR)
// One thread makes the nine taps of one (row, k) pair: it reads the nine fp32 values of a 3x3 filter (36 contiguous bytes;
// one element per thread read them at a stride of nine floats, each tap from a far-away block: nine sweeps over the
// tensor) and stores one element into each of the nine tap images.  i < rows * kch.
template <typename T>
__device__ __forceinline__ void pack_conv3x3_elem(long long i, const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int dgrad, int force_frag) {
    constexpr int CK = Elem<T>::CK;
    const int rows = dgrad ? Ci : Co;          // GEMM N
    const int kch = dgrad ? Co : Ci;           // GEMM K channels
    if (i >= (long long)rows * kch) return;
    // 32-bit index arithmetic (the entry points check 9 * Co * Ci < 2^31)
    const unsigned iu = (unsigned)i;
    int row, kk, chunk;
    unsigned o0, ostride;                      // element of tap 0, distance between taps
    if (force_frag || fea_conv3x3_frag_pack(rows, sizeof(T))) {
        // out index = ((chunk * 9 + tap) * nt16 + nt) << 10 | ks << 9 | lane << 3 | j
        const int j = (int)(iu & 7), ln = (int)((iu >> 3) & 63), ks = (int)((iu >> 9) & 1);
        const unsigned blk = iu >> 10;
        const unsigned nt16 = (unsigned)rows >> 4;
        chunk = (int)(blk / nt16);
        const unsigned nt = blk - (unsigned)chunk * nt16;
        row = (int)nt * 16 + (ln & 15);
        kk = ks * 32 + (ln >> 4) * 8 + j;
        o0 = (((unsigned)chunk * 9 * nt16 + nt) << 10) | (iu & 1023);
        ostride = nt16 << 10;
    } else {
        // out index = ((chunk * 9 + tap) * rows + row) * CK + kk
        kk = (int)(iu % CK);
        const unsigned r = iu / CK;
        chunk = (int)(r / (unsigned)rows);
        row = (int)(r - (unsigned)chunk * (unsigned)rows);
        o0 = ((unsigned)chunk * 9 * (unsigned)rows + (unsigned)row) * CK + (unsigned)kk;
        ostride = (unsigned)rows * CK;
    }
    const int kg = chunk * CK + kk;
    const float* src = dgrad ? w + ((long long)kg * Ci + row) * 9 : w + ((long long)row * Ci + kg) * 9;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = src[t];
#pragma unroll
    for (int t = 0; t < 9; ++t) out[o0 + (unsigned)t * ostride] = from_f32<T>(v[dgrad ? 8 - t : t]);
}
template <typename T>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int dgrad, int force_frag) {
    pack_conv3x3_elem<T>((long long)blockIdx.x * blockDim.x + threadIdx.x, w, out, Co, Ci, dgrad, force_frag);
}
template <typename T>
__global__ void pack_conv3x3_im2col_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int C, int Kpad) {
    constexpr int CK = Elem<T>::CK;
    const long long total = (long long)Co * Kpad;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // out[chunk][co][j], k = chunk*CK + j
    const int j = (int)(i % CK);
    const int co = (int)((i / CK) % Co);
    const int chunk = (int)(i / ((long long)CK * Co));
    const int k = chunk * CK + j;
    float v = 0.f;
    if (k < 9 * C) { const int tap = k / C, c = k - tap * C; v = w[((long long)co * C + c) * 9 + tap]; }
    out[i] = from_f32<T>(v);
}
// first layer (fea_conv3x3_first): out[ks][co/16][lane = q*16 + co%16][j], k = ks*32 + q*8 + j = tap*C + c
__device__ __forceinline__ void pack_conv3x3_first_elem(long long i, const float* __restrict__ w, __bf16* __restrict__ out, int Co, int C, int KS) {
    const long long total = (long long)KS * Co * 32;
    if (i >= total) return;
    const int j = (int)(i & 7), ln = (int)((i >> 3) & 63);
    const long long blk = i >> 9;
    const int nt16 = Co >> 4;
    const int ks = (int)(blk / nt16), co = (int)(blk % nt16) * 16 + (ln & 15);
    const int k = ks * 32 + (ln >> 4) * 8 + j;
    float v = 0.f;
    if (k < 9 * C) { const int tap = k / C, c = k - tap * C; v = w[((long long)co * C + c) * 9 + tap]; }
    out[i] = (__bf16)v;
}
__global__ void pack_conv3x3_first_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Co, int C, int KS) {
    pack_conv3x3_first_elem((long long)blockIdx.x * blockDim.x + threadIdx.x, w, out, Co, C, KS);
}

template <typename T>
__device__ __forceinline__ void pack_upconv_elem(long long i, const float* __restrict__ w, T* __restrict__ out, int Ci, int Co, int bwd) {
    constexpr int CK = Elem<T>::CK;
    const long long total = (long long)4 * Ci * Co;
    if (i >= total) return;
    const unsigned iu = (unsigned)i;             // (4 * Ci * Co < 2^31, checked by the entry points)
    const int j = (int)(iu % CK);
    const unsigned r = iu / CK;
    float v;
    if (!bwd) {       // out[chunk][ab*Co+co][j] = w[chunk*CK+j][co][ab]
        const int chunk = (int)(r / (unsigned)(4 * Co));
        const int row = (int)(r - (unsigned)chunk * (unsigned)(4 * Co));
        const int ab = row / Co, co = row - ab * Co;
        v = w[(((long long)(chunk * CK + j)) * Co + co) * 4 + ab];
    } else {          // out[ab*(Co/CK)+chunk][ci][j] = w[ci][chunk*CK+j][ab]
        const int kc = (int)(r / (unsigned)Ci);
        const int ci = (int)(r - (unsigned)kc * (unsigned)Ci);
        const int nch = Co / CK;
        const int ab = kc / nch, chunk = kc - ab * nch;
        v = w[((long long)ci * Co + chunk * CK + j) * 4 + ab];
    }
    out[i] = from_f32<T>(v);
}
template <typename T>
__global__ void pack_upconv_kernel(const float* __restrict__ w, T* __restrict__ out, int Ci, int Co, int bwd) {
    pack_upconv_elem<T>((long long)blockIdx.x * blockDim.x + threadIdx.x, w, out, Ci, Co, bwd);
}

// Every weight pack of a training step in ONE launch (fea_pack_batch): a step re-packs ~43 tensors, most of them far
// smaller than a launch is long.  Block b serves the job with the largest first_block <= b.
struct PackJob { const float* src; void* dst; int kind, a, b, c, first_block, reserved; };     // == FeaPackJob (fea_unet.h)
template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackJob* __restrict__ jobs, int njobs) {
    __shared__ int which;
    if (threadIdx.x < 64) {
        const bool le = (int)threadIdx.x < njobs && jobs[threadIdx.x].first_block <= (int)blockIdx.x;
        const unsigned long long m = __ballot(le);
        if (threadIdx.x == 0) which = __popcll(m) - 1;
    }
    __syncthreads();
    const PackJob j = jobs[which];
    const long long i = (long long)(blockIdx.x - j.first_block) * 256 + threadIdx.x;
    switch (j.kind) {
        case 0: pack_conv3x3_elem<T>(i, j.src, (T*)j.dst, j.a, j.b, j.c, 0); break;
        case 1: if constexpr (sizeof(T) == 2) pack_conv3x3_elem<T>(i, j.src, (T*)j.dst, j.a, j.b, j.c, 1); break;
        case 2: pack_upconv_elem<T>(i, j.src, (T*)j.dst, j.a, j.b, j.c); break;
        case 3: if constexpr (sizeof(T) == 2) pack_conv3x3_first_elem(i, j.src, (__bf16*)j.dst, j.a, j.b, (9 * j.b + 31) / 32); break;
        default: break;
    }
}

// ---------------------------------------------------------------------------
// BatchNorm statistics
// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    // 256 threads; result valid in thread 0
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// Partial-sum tables with very many rows (one row per conv pixel tile: 65,536 at 512x512,
// batch 64) are folded IN PLACE before the per-channel finalize kernels: block b adds rows
// [b*rpb, (b+1)*rpb) with coalesced row reads (double accumulation) and stores the result
// into its own first row, so the finalize kernels read `nb` rows at stride rpb.
__global__ __launch_bounds__(256) void fold_rows_kernel(float* __restrict__ part, int nrows, int ldf, int rpb) {
    // 16 bytes per lane, four rows in flight per thread (one dword per lane and one row at a time took 24 us per call on
    // the 33 MB tables of the 512x512 layers: twelve calls per step)
    __shared__ double red[256][4];
    const int ld4 = ldf >> 2;                    // host guarantees ldf % 4 == 0
    const int L = ld4 < 256 ? ld4 : 256, slots = 256 / L;
    const int cl = threadIdx.x % L, slot = threadIdx.x / L;
    const int r0 = blockIdx.x * rpb, r1 = min(r0 + rpb, nrows);
    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(part);
    for (int c0 = 0; c0 < ld4; c0 += 256) {
        const int col = c0 + cl;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if (slot < slots && col < ld4) {
            int r = r0 + slot;
            for (; r + 3 * slots < r1; r += 4 * slots) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = p4[(long long)(r + u * slots) * ld4 + col];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[0] += (double)v[u].x; a[1] += (double)v[u].y; a[2] += (double)v[u].z; a[3] += (double)v[u].w; }
            }
            for (; r < r1; r += slots) {
                const float4 v = p4[(long long)r * ld4 + col];
                a[0] += (double)v.x; a[1] += (double)v.y; a[2] += (double)v.z; a[3] += (double)v.w;
            }
        }
        __syncthreads();                          // every read of row r0 is done before it is overwritten
#pragma unroll
        for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = a[j];
        __syncthreads();
        if (slot == 0 && col < ld4) {
            double t[4] = {0.0, 0.0, 0.0, 0.0};
            for (int sl = 0; sl < slots; ++sl)
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] += red[sl * L + cl][j];
            reinterpret_cast<float4*>(part)[(long long)r0 * ld4 + col] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
        }
    }
}
static void fold_rows(float* part, int& nrows, int& rstride, int ldf, hipStream_t s) {
    rstride = 1;
    if (nrows < 4096 || ldf % 4 || (ldf > 1024 && ldf % 1024) || ((uintptr_t)part & 15)) return;
    const int nb = 1024, rpb = (nrows + nb - 1) / nb;
    const int nfold = (nrows + rpb - 1) / rpb;
    hipLaunchKernelGGL(fold_rows_kernel, dim3(nfold), dim3(256), 0, s, part, nrows, ldf, rpb);
    nrows = nfold; rstride = rpb;
}

// one block per channel
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ stats, int ntiles, int rstride, long long count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum, float eps,
                                                          float* scale, float* shift, float* save_mean, float* save_invstd, int C) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += 256) {
        const float2 v = *reinterpret_cast<const float2*>(stats + ((long long)t * rstride * C + c) * 2);
        s += v.x; q += v.y;
    }
    s = block_sum_d(s, sh);
    q = block_sum_d(q, sh);
    if (threadIdx.x == 0) {
        const double mean = s / (double)count;
        double var = q / (double)count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        save_mean[c] = (float)mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// Streaming access hints of the BatchNorm passes (A/B switches; bit 0: nontemporal loads in the backward passes, bit 1:
// nontemporal stores, bit 2: nontemporal loads of y in the forward pass).  Measured on one box, ms per step: backward
// apply 9.63 -> 9.27 and reduce 2.31 -> 2.06 with bit 0, forward 5.75 -> 5.62 with bit 2, stores: no change.
#ifndef BN_NT
#define BN_NT 5
#endif
__device__ __forceinline__ u32x4 bn_ld(const void* p) {
    if constexpr (BN_NT & 1) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else return *reinterpret_cast<const u32x4*>(p);
}
__device__ __forceinline__ void bn_st(void* p, const u32x4& v) {
    if constexpr (BN_NT & 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = v;
}

// ---------------------------------------------------------------------------
// BN + ReLU (+ 2x2 max-pool) forward
// ---------------------------------------------------------------------------
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, T* __restrict__ out_full, int ld_full,
                                                           T* __restrict__ out_pool, int N, int H, int W, int C) {
    constexpr int V = Vec<T>::N;
    const int gpp = C / V;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    const long long total = (long long)N * Ho * Wo * gpp;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // 32-bit divisions where the item count allows (always, for this network): the 64-bit ones are ~100 VALU instructions
    // each, per 16-byte item of a kernel that should only stream
    int grp, xo, yo, n;
    if (total < (1ll << 32)) {
        const unsigned iu = (unsigned)i;
        grp = (int)(iu % (unsigned)gpp);
        unsigned pix = iu / (unsigned)gpp;
        xo = (int)(pix % (unsigned)Wo); pix /= (unsigned)Wo;
        yo = (int)(pix % (unsigned)Ho);
        n = (int)(pix / (unsigned)Ho);
    } else {
        grp = (int)(i % gpp);
        long long pix = i / gpp;
        xo = (int)(pix % Wo); pix /= Wo;
        yo = (int)(pix % Ho);
        n = (int)(pix / Ho);
    }
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = scale[grp * V + j]; sh[j] = shift[grp * V + j]; }
    if (!POOL) {
        const long long p = ((long long)n * H + yo) * W + xo;
        float f[V];
        unpack<T>((BN_NT & 4) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y + p * C + grp * V))
                              : *reinterpret_cast<const u32x4*>(y + p * C + grp * V), f);
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = bn_act(f[j], sc[j], sh[j]);
        bn_st(out_full + p * ld_full + grp * V, pack<T>(f));
    } else {
        float mx[V];
#pragma unroll
        for (int j = 0; j < V; ++j) mx[j] = 0.f;     // activations are >= 0
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
            const long long p = ((long long)n * H + 2 * yo + (w4 >> 1)) * W + 2 * xo + (w4 & 1);
            float f[V];
            unpack<T>((BN_NT & 4) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y + p * C + grp * V))
                                  : *reinterpret_cast<const u32x4*>(y + p * C + grp * V), f);
#pragma unroll
            for (int j = 0; j < V; ++j) { f[j] = bn_act(f[j], sc[j], sh[j]); mx[j] = fmaxf(mx[j], f[j]); }
            *reinterpret_cast<u32x4*>(out_full + p * ld_full + grp * V) = pack<T>(f);
        }
        const long long po = ((long long)n * Ho + yo) * Wo + xo;
        *reinterpret_cast<u32x4*>(out_pool + po * C + grp * V) = pack<T>(mx);
    }
}

// ---------------------------------------------------------------------------
// BN + ReLU (+ pool routing) backward
// ---------------------------------------------------------------------------
// Work item = (window or pixel, channel group).  Blocks are 256 threads laid out
// as [256/gpp pixel slots][gpp channel groups] when gpp <= 256.
template <typename T, bool POOL, bool APPLY>
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(const T* __restrict__ da_full, int ld_da, const T* __restrict__ dpool,
                                                          const T* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, const float* __restrict__ coef,
                                                          T* __restrict__ dy, float* __restrict__ part,
                                                          int N, int H, int W, int C, long long items_per_block, int rows_bound) {
    constexpr int V = Vec<T>::N;
    const int gpp = C / V;                       // host guarantees gpp <= 256 and 256 % gpp == 0
    const int slots = 256 / gpp;
    const int grp = threadIdx.x % gpp, slot = threadIdx.x / gpp;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    const long long nitems = (long long)N * Ho * Wo;
    const long long it0 = (long long)blockIdx.x * items_per_block;
    const long long it1 = it0 + items_per_block < nitems ? it0 + items_per_block : nitems;

    float sc[V], sh[V], mu[V], is[V], c1[V], c2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = grp * V + j;
        sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = save_mean[c]; is[j] = save_invstd[c];
        if (APPLY) { c1[j] = coef[2 * c]; c2[j] = coef[2 * c + 1]; } else { c1[j] = 0.f; c2[j] = 0.f; }
    }
    float a0[V], a1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { a0[j] = 0.f; a1[j] = 0.f; }

    // UN items per iteration: all loads of the batch are issued before the first use
    // (memory-level parallelism; the loop is otherwise one dependent round trip per item)
    constexpr int NP = POOL ? 4 : 1;
#ifndef BN_APPLY_UN
#define BN_APPLY_UN 2
#endif
    constexpr int UN = POOL ? 1 : (APPLY ? BN_APPLY_UN : 4);
    for (long long itb = it0 + slot; itb < it1; itb += (long long)slots * UN) {
        u32x4 ry[UN][NP], rg[UN][NP], rp[UN];
        long long pidx[UN][NP];
        bool live[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long it = itb + (long long)u * slots;
            live[u] = it < it1;
            long long t = live[u] ? it : itb;
            // without pooling the item index IS the pixel index; with it, 32-bit divisions where the count allows
            int xo = 0, yo = 0, n = 0;
            if constexpr (POOL) {
                if (nitems < (1ll << 32)) {
                    unsigned tu = (unsigned)t;
                    xo = (int)(tu % (unsigned)Wo); tu /= (unsigned)Wo;
                    yo = (int)(tu % (unsigned)Ho);
                    n = (int)(tu / (unsigned)Ho);
                } else {
                    long long tl = t;
                    xo = (int)(tl % Wo); tl /= Wo;
                    yo = (int)(tl % Ho);
                    n = (int)(tl / Ho);
                }
            }
#pragma unroll
            for (int w4 = 0; w4 < NP; ++w4) {
                pidx[u][w4] = POOL ? ((long long)n * H + 2 * yo + (w4 >> 1)) * W + 2 * xo + (w4 & 1) : t;
                ry[u][w4] = bn_ld(y + pidx[u][w4] * C + grp * V);
                if (da_full) rg[u][w4] = bn_ld(da_full + pidx[u][w4] * ld_da + grp * V);
            }
            if (POOL && dpool) rp[u] = *reinterpret_cast<const u32x4*>(dpool + (((long long)n * Ho + yo) * Wo + xo) * C + grp * V);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (!live[u]) continue;
            float yv[NP][V], g[NP][V];
#pragma unroll
            for (int w4 = 0; w4 < NP; ++w4) {
                unpack<T>(ry[u][w4], yv[w4]);
                if (da_full) unpack<T>(rg[u][w4], g[w4]);
                else {
#pragma unroll
                    for (int j = 0; j < V; ++j) g[w4][j] = 0.f;
                }
            }
            if (POOL && dpool) {
                float dp[V];
                unpack<T>(rp[u], dp);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    // first maximum in row-major window order (MaxPool2d backward, models.py:24)
                    float best = bn_act(yv[0][j], sc[j], sh[j]);
                    int bi = 0;
#pragma unroll
                    for (int w4 = 1; w4 < NP; ++w4) {
                        const float a = bn_act(yv[w4][j], sc[j], sh[j]);
                        if (a > best) { best = a; bi = w4; }
                    }
#pragma unroll
                    for (int w4 = 0; w4 < NP; ++w4) g[w4][j] += (bi == w4) ? dp[j] : 0.f;
                }
            }
#pragma unroll
            for (int w4 = 0; w4 < NP; ++w4) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float act = bn_act(yv[w4][j], sc[j], sh[j]);
                    const float gg = act > 0.f ? g[w4][j] : 0.f;
                    const float xh = (yv[w4][j] - mu[j]) * is[j];
                    if (!APPLY) { a0[j] += gg; a1[j] += gg * xh; }
                    else {
                        const float d = sc[j] * (gg - c1[j] - xh * c2[j]);
                        o[j] = d; a0[j] += d;
                    }
                }
                if (APPLY) bn_st(dy + pidx[u][w4] * C + grp * V, pack<T>(o));
            }
        }
    }
    // reduce over the pixel slots of this block, fixed order
    __shared__ float red[256 * 2 * 8];           // [slot][grp][2][V] <= 256*16 floats
    float* mine = red + (size_t)threadIdx.x * 2 * V;
#pragma unroll
    for (int j = 0; j < V; ++j) { mine[j] = a0[j]; mine[V + j] = a1[j]; }
    __syncthreads();
    if (threadIdx.x < gpp) {
        float s0[V], s1[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
        for (int sl = 0; sl < slots; ++sl) {
            const float* o = red + (size_t)(sl * gpp + threadIdx.x) * 2 * V;
#pragma unroll
            for (int j = 0; j < V; ++j) { s0[j] += o[j]; s1[j] += o[V + j]; }
        }
        float* dst = part + ((long long)blockIdx.x * C + threadIdx.x * V) * 2;
#pragma unroll
        for (int j = 0; j < V; ++j) { dst[2 * j] = s0[j]; dst[2 * j + 1] = s1[j]; }
        // the caller sized the table for rows_bound rows (fea_bn_bwd_blocks) and reduces that many: the rows no block owns are
        // zeroed here, dealt over the blocks (was a hipMemsetAsync in front of every launch: one more small launch, 22 per step)
        for (int r = gridDim.x + blockIdx.x; r < rows_bound; r += gridDim.x) {
            float* z = part + ((long long)r * C + threadIdx.x * V) * 2;
#pragma unroll
            for (int j = 0; j < 2 * V; ++j) z[j] = 0.f;
        }
    }
}

// one block per channel: dgamma, dbeta, coef
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int rstride, long long count,
                                                              const float* __restrict__ gamma, const float* __restrict__ save_invstd,
                                                              float* dgamma, float* dbeta, float* coef, int C) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int t = threadIdx.x; t < nblk; t += 256) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((long long)t * rstride * C + c) * 2);
        s += v.x; q += v.y;
    }
    s = block_sum_d(s, sh);
    q = block_sum_d(q, sh);
    if (threadIdx.x == 0) {
        dbeta[c] = (float)s;
        dgamma[c] = (float)q;
        coef[2 * c] = (float)(s / (double)count);
        coef[2 * c + 1] = (float)(q / (double)count);
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nblk, int rstride, int C, int slot,
                                                              float* __restrict__ out) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int t = threadIdx.x; t < nblk; t += 256) s += part[((long long)t * rstride * C + c) * 2 + slot];
    s = block_sum_d(s, sh);
    if (threadIdx.x == 0) out[c] = (float)s;
}

template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, int ld, long long npix, int C,
                                                          float* __restrict__ part, long long pix_per_block) {
    constexpr int V = Vec<T>::N;
    const int gpp = C / V, slots = 256 / gpp;
    const int grp = threadIdx.x % gpp, slot = threadIdx.x / gpp;
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    const long long p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
    float a[V];
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = 0.f;
    for (long long p = p0 + slot; p < p1; p += slots) {
        float f[V];
        unpack<T>(*reinterpret_cast<const u32x4*>(x + p * ld + grp * V), f);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] += f[j];
    }
    __shared__ float red[256 * 8];
#pragma unroll
    for (int j = 0; j < V; ++j) red[threadIdx.x * V + j] = a[j];
    __syncthreads();
    if (threadIdx.x < gpp) {
        float s[V];
#pragma unroll
        for (int j = 0; j < V; ++j) s[j] = 0.f;
        for (int sl = 0; sl < slots; ++sl)
#pragma unroll
            for (int j = 0; j < V; ++j) s[j] += red[(sl * gpp + threadIdx.x) * V + j];
        float* dst = part + ((long long)blockIdx.x * C + threadIdx.x * V) * 2;
#pragma unroll
        for (int j = 0; j < V; ++j) { dst[2 * j] = s[j]; dst[2 * j + 1] = 0.f; }
    }
}

// ---------------------------------------------------------------------------
// host wrappers
// ---------------------------------------------------------------------------
#define DT_SWITCH(dtype, who, ...)                                                     \
    if ((dtype) == FEA_BF16) { typedef __bf16 T; __VA_ARGS__; }                        \
    else if ((dtype) == FEA_F32) { typedef float T; __VA_ARGS__; }                     \
    else { fea_set_error("%s: bad dtype %d", who, (int)(dtype)); return FEA_E_BADARG; }

static inline int velems(int dtype) { return dtype == FEA_BF16 ? 8 : 4; }

extern "C" int fea_im2col3x3_nchw(const float* x, void* out, int N, int C, int H, int W, int Kpad, int dtype, void* stream) {
    FEA_CHECK(x && out && N > 0 && C > 0 && H > 0 && W > 0, FEA_E_BADARG, "fea_im2col3x3_nchw: bad args");
    FEA_CHECK(Kpad % 64 == 0 && Kpad >= 9 * C, FEA_E_UNSUPPORTED_SHAPE, "fea_im2col3x3_nchw: Kpad=%d for C=%d", Kpad, C);
    const long long total = (long long)N * H * W * (Kpad / velems(dtype));
    DT_SWITCH(dtype, "fea_im2col3x3_nchw",
              hipLaunchKernelGGL((im2col3x3_kernel<T>), GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, x, (T*)out, N, C, H, W, Kpad));
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_nhwc_to_nchw_f32(const void* x, int ld, float* out, int N, int C, int H, int W, int dtype, void* stream) {
    FEA_CHECK(x && out && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C, FEA_E_BADARG, "fea_nhwc_to_nchw_f32: bad args");
    const long long total = (long long)N * C * H * W;
    DT_SWITCH(dtype, "fea_nhwc_to_nchw_f32",
              hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, (const T*)x, ld, out, N, C, H, W));
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_nchw_f32_to_nhwc(const float* x, void* out, int ld, int N, int C, int H, int W, int dtype, void* stream) {
    FEA_CHECK(x && out && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C, FEA_E_BADARG, "fea_nchw_f32_to_nhwc: bad args");
    const long long total = (long long)N * C * H * W;
    DT_SWITCH(dtype, "fea_nchw_f32_to_nhwc",
              hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, x, (T*)out, ld, N, C, H, W));
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_pack_conv3x3(const float* w, void* out, int Co, int Ci, int dgrad, int dtype, void* stream) {
    const int CK = dtype == FEA_BF16 ? 64 : 32;
    FEA_CHECK(w && out && Co > 0 && Ci > 0, FEA_E_BADARG, "fea_pack_conv3x3: bad args");
    FEA_CHECK((dgrad ? Co : Ci) % CK == 0, FEA_E_UNSUPPORTED_SHAPE, "fea_pack_conv3x3: K channels not a multiple of %d", CK);
    const long long total = (long long)9 * Co * Ci;
    FEA_CHECK(total < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "fea_pack_conv3x3: %lld elements", total);
    DT_SWITCH(dtype, "fea_pack_conv3x3",
              hipLaunchKernelGGL((pack_conv3x3_kernel<T>), GRID1D(total / 9, 256), dim3(256), 0, (hipStream_t)stream, w, (T*)out, Co, Ci, dgrad, 0));
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

// always MFMA fragment order (the persistent kernels of conv_p.hip keep the fragments in registers), bf16
extern "C" int fea_pack_conv3x3_frag(const float* w, void* out, int Co, int Ci, int dgrad, int dtype, void* stream) {
    FEA_CHECK(w && out && Co > 0 && Ci > 0 && dtype == FEA_BF16, FEA_E_BADARG, "fea_pack_conv3x3_frag: bad args (bf16 only)");
    FEA_CHECK(Co % 64 == 0 && Ci % 64 == 0, FEA_E_UNSUPPORTED_SHAPE, "fea_pack_conv3x3_frag: channels must be multiples of 64");
    const long long total = (long long)9 * Co * Ci;
    FEA_CHECK(total < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "fea_pack_conv3x3_frag: %lld elements", total);
    hipLaunchKernelGGL((pack_conv3x3_kernel<__bf16>), GRID1D(total / 9, 256), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)out, Co, Ci, dgrad, 1);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_pack_conv3x3_im2col(const float* w, void* out, int Co, int C, int Kpad, int dtype, void* stream) {
    FEA_CHECK(w && out && Co > 0 && C > 0 && Kpad % 64 == 0 && Kpad >= 9 * C, FEA_E_BADARG, "fea_pack_conv3x3_im2col: bad args");
    const long long total = (long long)Co * Kpad;
    DT_SWITCH(dtype, "fea_pack_conv3x3_im2col",
              hipLaunchKernelGGL((pack_conv3x3_im2col_kernel<T>), GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, w, (T*)out, Co, C, Kpad));
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_pack_conv3x3_first(const float* w, void* out, int Co, int C, int dtype, void* stream) {
    FEA_CHECK(w && out && Co > 0 && Co % 16 == 0 && C > 0 && 9 * C <= 64 && dtype == FEA_BF16, FEA_E_BADARG, "fea_pack_conv3x3_first: bad args");
    const int KS = (9 * C + 31) / 32;
    const long long total = (long long)KS * Co * 32;
    hipLaunchKernelGGL(pack_conv3x3_first_kernel, GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)out, Co, C, KS);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_pack_upconv2x2(const float* w, void* out, int Ci, int Co, int bwd, int dtype, void* stream) {
    FEA_CHECK(w && out && Ci > 0 && Co > 0 && Ci % 64 == 0 && Co % 64 == 0, FEA_E_BADARG, "fea_pack_upconv2x2: bad args");
    const long long total = (long long)4 * Ci * Co;
    FEA_CHECK(total < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "fea_pack_upconv2x2: %lld elements", total);
    DT_SWITCH(dtype, "fea_pack_upconv2x2",
              hipLaunchKernelGGL((pack_upconv_kernel<T>), GRID1D(total, 256), dim3(256), 0, (hipStream_t)stream, w, (T*)out, Ci, Co, bwd));
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

// blocks of 256 elements that job (kind, a, b, c) needs; kinds: 0 fea_pack_conv3x3(Co, Ci, dgrad), 1 fea_pack_conv3x3_frag(Co, Ci,
// dgrad), 2 fea_pack_upconv2x2(Ci, Co, bwd), 3 fea_pack_conv3x3_first(Co, C)
extern "C" int fea_pack_job_blocks(int kind, int a, int b, int c) {
    long long total;
    if (kind == 0 || kind == 1) total = (long long)a * b;                 // one thread per 3x3 filter
    else if (kind > 3 || kind < 0) return FEA_E_BADARG;
    else if (kind == 2) total = (long long)4 * a * b;
    else if (kind == 3) total = (long long)((9 * b + 31) / 32) * a * 32;
    else return FEA_E_BADARG;
    if (a <= 0 || b <= 0 || total * (kind <= 1 ? 9 : 1) >= (1ll << 31)) return FEA_E_BADARG;
    return (int)((total + 255) / 256);
}

extern "C" int fea_pack_batch(const void* jobs_dev, int njobs, int nblocks, int dtype, void* stream) {
    FEA_CHECK(jobs_dev && njobs > 0 && njobs <= 64 && nblocks > 0, FEA_E_BADARG, "fea_pack_batch: bad args (1..64 jobs)");
    DT_SWITCH(dtype, "fea_pack_batch",
              hipLaunchKernelGGL((pack_batch_kernel<T>), dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs_dev, njobs));
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_bn_finalize(const float* stats, int ntiles, long long count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* stream) {
    FEA_CHECK(stats && gamma && beta && scale && shift && save_mean && save_invstd && ntiles > 0 && count > 0 && C > 0,
              FEA_E_BADARG, "fea_bn_finalize: bad args");
    FEA_CHECK((running_mean == nullptr) == (running_var == nullptr), FEA_E_BADARG, "fea_bn_finalize: running stats must both be given");
    int rstride;
    fold_rows(const_cast<float*>(stats), ntiles, rstride, C * 2, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stats, ntiles, rstride, count, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd, C);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                  float* scale, float* shift, int C, void* stream) {
    FEA_CHECK(gamma && beta && rm && rv && scale && shift && C > 0, FEA_E_BADARG, "fea_bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, GRID1D(C, 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, rm, rv, eps, scale, shift, C);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_bn_relu_pool(const void* y, const float* scale, const float* shift, void* out_full, int ld_full,
                                void* out_pool, int N, int H, int W, int C, int dtype, void* stream) {
    FEA_CHECK(y && scale && shift && out_full && N > 0 && H > 0 && W > 0 && C > 0, FEA_E_BADARG, "fea_bn_relu_pool: bad args");
    FEA_CHECK(C % 8 == 0 && ld_full >= C && ld_full % 8 == 0, FEA_E_UNSUPPORTED_SHAPE, "fea_bn_relu_pool: C=%d ld=%d", C, ld_full);
    FEA_CHECK(!out_pool || (H % 2 == 0 && W % 2 == 0), FEA_E_UNSUPPORTED_SHAPE, "fea_bn_relu_pool: pooling needs even H,W");
    const int V = velems(dtype);
    hipStream_t s = (hipStream_t)stream;
    if (out_pool) {
        const long long total = (long long)N * (H / 2) * (W / 2) * (C / V);
        DT_SWITCH(dtype, "fea_bn_relu_pool",
                  hipLaunchKernelGGL((bn_relu_pool_kernel<T, true>), GRID1D(total, 256), dim3(256), 0, s, (const T*)y, scale, shift,
                                     (T*)out_full, ld_full, (T*)out_pool, N, H, W, C));
    } else {
        const long long total = (long long)N * H * W * (C / V);
        DT_SWITCH(dtype, "fea_bn_relu_pool",
                  hipLaunchKernelGGL((bn_relu_pool_kernel<T, false>), GRID1D(total, 256), dim3(256), 0, s, (const T*)y, scale, shift,
                                     (T*)out_full, ld_full, (T*)nullptr, N, H, W, C));
    }
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

// number of blocks (= rows of `part`) used by the BN backward kernels; a pure
// function of the shape so reduce and apply agree.
static int red_blocks(long long nitems, int slots) {
    long long per_iter = slots;
    long long want = (nitems + per_iter * 8 - 1) / (per_iter * 8);   // >= 8 iterations per block
    if (want > 2048) want = 2048;
    if (want < 1) want = 1;
    return (int)want;
}
extern "C" int fea_bn_bwd_blocks(int N, int H, int W, int C) {
    (void)C;
    // upper bound valid for both dtypes and both POOL variants
    return red_blocks((long long)N * H * W, 1);
}

static int bn_bwd_launch(bool apply, const void* da_full, int ld_da, const void* dpool, const void* y, const float* scale,
                         const float* shift, const float* save_mean, const float* save_invstd, const float* coef, void* dy,
                         float* part, int N, int H, int W, int C, int dtype, void* stream, const char* who) {
    FEA_CHECK(y && scale && shift && save_mean && save_invstd && part && (da_full || dpool), FEA_E_BADARG, "%s: null pointer", who);
    FEA_CHECK(!apply || (coef && dy), FEA_E_BADARG, "%s: null coef/dy", who);
    const int V = velems(dtype);
    FEA_CHECK(dtype == FEA_BF16 || dtype == FEA_F32, FEA_E_BADARG, "%s: bad dtype", who);
    FEA_CHECK(C % V == 0 && C / V <= 256 && 256 % (C / V) == 0, FEA_E_UNSUPPORTED_SHAPE, "%s: C=%d unsupported", who, C);
    FEA_CHECK(!da_full || (ld_da >= C && ld_da % 8 == 0), FEA_E_BADARG, "%s: ld_da=%d", who, ld_da);
    const bool pool = dpool != nullptr;
    FEA_CHECK(!pool || (H % 2 == 0 && W % 2 == 0), FEA_E_UNSUPPORTED_SHAPE, "%s: pooling needs even H,W", who);
    const long long nitems = pool ? (long long)N * (H / 2) * (W / 2) : (long long)N * H * W;
    const int slots = 256 / (C / V);
    const int nblk = red_blocks(nitems, slots);
    FEA_CHECK(nblk <= fea_bn_bwd_blocks(N, H, W, C), FEA_E_WORKSPACE, "%s: internal block count", who);
    long long ipb = (nitems + nblk - 1) / nblk;
    ipb = (ipb + slots - 1) / slots * slots;
    hipStream_t s = (hipStream_t)stream;
    // the rows of `part` that no block writes (nblk may be < the bound the caller sized for) are zeroed by the kernel itself
    const int bound = fea_bn_bwd_blocks(N, H, W, C);
#define BNB(POOLV, APPLYV) hipLaunchKernelGGL((bn_relu_bwd_kernel<T, POOLV, APPLYV>), dim3(nblk), dim3(256), 0, s, (const T*)da_full, ld_da, \
        (const T*)dpool, (const T*)y, scale, shift, save_mean, save_invstd, coef, (T*)dy, part, N, H, W, C, ipb, bound)
    DT_SWITCH(dtype, who,
              if (pool) { if (apply) BNB(true, true); else BNB(true, false); }
              else      { if (apply) BNB(false, true); else BNB(false, false); });
#undef BNB
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_bn_relu_bwd_reduce(const void* da_full, int ld_da, const void* dpool, const void* y, const float* scale,
                                      const float* shift, const float* save_mean, const float* save_invstd, float* part,
                                      int N, int H, int W, int C, int dtype, void* stream) {
    return bn_bwd_launch(false, da_full, ld_da, dpool, y, scale, shift, save_mean, save_invstd, nullptr, nullptr, part,
                         N, H, W, C, dtype, stream, "fea_bn_relu_bwd_reduce");
}
extern "C" int fea_bn_relu_bwd_apply(const void* da_full, int ld_da, const void* dpool, const void* y, const float* scale,
                                     const float* shift, const float* save_mean, const float* save_invstd, const float* coef,
                                     void* dy, float* part, int N, int H, int W, int C, int dtype, void* stream) {
    return bn_bwd_launch(true, da_full, ld_da, dpool, y, scale, shift, save_mean, save_invstd, coef, dy, part,
                         N, H, W, C, dtype, stream, "fea_bn_relu_bwd_apply");
}

extern "C" int fea_bn_bwd_finalize(const float* part, int nblk, long long count, const float* gamma, const float* save_invstd,
                                   float* dgamma, float* dbeta, float* coef, int C, void* stream) {
    FEA_CHECK(part && gamma && save_invstd && dgamma && dbeta && coef && nblk > 0 && count > 0 && C > 0, FEA_E_BADARG,
              "fea_bn_bwd_finalize: bad args");
    int rstride;
    fold_rows(const_cast<float*>(part), nblk, rstride, C * 2, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, part, nblk, rstride, count, gamma, save_invstd,
                       dgamma, dbeta, coef, C);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_reduce_partials(const float* part, int nblk, int C, int slot, float* out, void* stream) {
    FEA_CHECK(part && out && nblk > 0 && C > 0 && (slot == 0 || slot == 1), FEA_E_BADARG, "fea_reduce_partials: bad args");
    int rstride;
    fold_rows(const_cast<float*>(part), nblk, rstride, C * 2, (hipStream_t)stream);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, part, nblk, rstride, C, slot, out);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_reduce_partials_n(const float* part, int nblk, int ld, int n, int slot, float* out, void* stream) {
    FEA_CHECK(part && out && nblk > 0 && ld > 0 && n > 0 && n <= ld && (slot == 0 || slot == 1), FEA_E_BADARG,
              "fea_reduce_partials_n: bad args");
    int rstride;
    fold_rows(const_cast<float*>(part), nblk, rstride, ld * 2, (hipStream_t)stream);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, part, nblk, rstride, ld, slot, out);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_channel_sum(const void* x, int ld, long long npix, int C, float* part, int* nblk_out, int dtype, void* stream) {
    const int V = velems(dtype);
    FEA_CHECK(x && part && nblk_out && npix > 0 && C > 0, FEA_E_BADARG, "fea_channel_sum: bad args");
    FEA_CHECK(C % V == 0 && C / V <= 256 && 256 % (C / V) == 0 && ld >= C && ld % 8 == 0, FEA_E_UNSUPPORTED_SHAPE,
              "fea_channel_sum: C=%d ld=%d", C, ld);
    const int slots = 256 / (C / V);
    const int nblk = red_blocks(npix, slots);
    long long ppb = (npix + nblk - 1) / nblk;
    ppb = (ppb + slots - 1) / slots * slots;
    DT_SWITCH(dtype, "fea_channel_sum",
              hipLaunchKernelGGL((channel_sum_kernel<T>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const T*)x, ld, npix, C, part, ppb));
    FEA_LAUNCH_CHECK();
    *nblk_out = nblk;
    return FEA_OK;
}
