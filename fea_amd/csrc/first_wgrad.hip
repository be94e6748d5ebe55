// Weight gradient of the FIRST layer, enc1[0] = Conv2d(in_channels, 64, 3, padding=1)
// (models.py:10,28), bf16 path:
//
//   dW[co][c][tap] = sum_p dy[p][co] * x[p + tap][c]       dy: NHWC bf16, x: fp32 NCHW
//
// GEMM view: M = co (64 per workgroup), N = 9*C (<= 64, index n = tap*C + c), reduction
// over pixels.  HBM-bound on reading dy once (the input has 3 channels): the kernel only
// has to keep enough bytes in flight.  One 256-thread workgroup walks a range of
// 8 x 32-pixel tiles; each tile's dy rows and the fp32 halo of x are brought in by
// LDS-DMA into a two-stage ring (loads of tile t+1 in flight while tile t is multiplied).
// Each of the four waves takes two of the tile's eight 32-pixel rows (= MFMA k-steps,
// v_mfma_f32_16x16x32_bf16): A (dy, k = pixel) comes from the transposing LDS read, B
// is gathered from the fp32 halo and rounded to bf16 exactly as the forward kernel
// (igemm.hip MODE 3) does.  The four waves are summed through LDS; workgroups write
// fp32 slabs that a second kernel adds in a fixed order (deterministic).
#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

struct FirstWgradArgs {
    const __bf16* dy; const float* x; float* slab;
    int lddy, C, H, W, Cout;
    int tiles_x, tiles_y, ntiles, tiles_per_split;
    // BN variant (fea_conv3x3_first_wgrad_bnbwd): `dy` is the gradient da of the activated tensor, the conv-output gradient
    // is formed in LDS from it and the pre-BN tensor y (BatchNorm + ReLU backward apply, elementwise.hip bn_relu_bwd_kernel)
    const __bf16* ybn; int ldy;
    const float *sc, *sh, *mu, *is, *coef;
    float* part;                       // [split][Cout][2]: slot 0 = sum of the conv-output gradient (conv bias gradient), slot 1 = 0
};

constexpr int FW_HALO = 10 * 34;                 // floats per channel
constexpr int FW_DY_BYTES = 256 * 128;
// halo region of a stage: XB = 4 KB for C <= 3 (3 * 340 * 4 = 4080 B: two 36 KB stages, two
// workgroups per CU), 10 KB up to C = 7

// 32-byte unit swizzle of the dy rows: rows read together by one ds_read_b64_tr_b16
// half-wave ({x..x+3} and {x+8..x+11}, same 32-byte channel block) land in distinct banks
__device__ __forceinline__ int fw_swz(int x) { return ((x >> 1) & 1) | (((x >> 3) & 1) << 1); }

template <bool BN, int NW>
__device__ __forceinline__ void fw_issue(const FirstWgradArgs& g, int m0, int wave, int lane, int tile, char* sb,
                                         const int (&xdesc)[10], int nxi, bool fast = false, int lo_d = 0, int lo_y = 0) {
    constexpr int DYB = BN ? 2 * FW_DY_BYTES : FW_DY_BYTES;      // BN: [da tile][y tile][halo]
    int tt = tile;
    const int tile_x = tt % g.tiles_x; tt /= g.tiles_x;
    const int tile_y = tt % g.tiles_y;
    const int n_img = tt / g.tiles_y;
    const int y0 = tile_y * 8, x0 = tile_x * 32;
    const int H = g.H, W = g.W;
    const __bf16* dimg = g.dy + (long long)n_img * H * W * g.lddy + m0;
    const float* ximg = g.x + (long long)n_img * g.C * H * W;
    if (fast) {
        // whole tiles: no bounds checks; a lane's element offset from the tile origin is lo + it * (NW / 4 rows), the origin is uniform
        const __bf16* dt0 = dimg + (long long)(y0 * W + x0) * g.lddy;
        const __bf16* yt0 = BN ? g.ybn + ((long long)n_img * H * W + y0 * W + x0) * g.ldy + m0 : nullptr;
#pragma unroll
        for (int it = 0; it < 32 / NW; ++it) {
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(dt0 + lo_d + it * (NW / 4) * W * g.lddy), (lds_void_t*)(sb + (it * NW + wave) * 1024), 16, 0, 0);
            if constexpr (BN)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(yt0 + lo_y + it * (NW / 4) * W * g.ldy),
                                                 (lds_void_t*)(sb + FW_DY_BYTES + (it * NW + wave) * 1024), 16, 0, 0);
        }
    } else
#pragma unroll
    for (int it = 0; it < 32 / NW; ++it) {
        const int row = (it * NW + wave) * 8 + (lane >> 3);
        const int ty = row >> 5, tx = row & 31;
        const int gy = y0 + ty, gx = x0 + tx;
        const int chunk = (lane & 7) ^ (fw_swz(tx) << 1);
        const __bf16* src = (gy < H && gx < W) ? dimg + (long long)(gy * W + gx) * g.lddy + chunk * 8
                                               : reinterpret_cast<const __bf16*>(fea_zero_line);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sb + (it * NW + wave) * 1024), 16, 0, 0);
        if constexpr (BN) {
            const __bf16* yimg = g.ybn + (long long)n_img * H * W * g.ldy + m0;
            const __bf16* ysrc = (gy < H && gx < W) ? yimg + (long long)(gy * W + gx) * g.ldy + chunk * 8
                                                    : reinterpret_cast<const __bf16*>(fea_zero_line);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)ysrc, (lds_void_t*)(sb + FW_DY_BYTES + (it * NW + wave) * 1024), 16, 0, 0);
        }
    }
#pragma unroll
    for (int it = 0; it < 40 / NW; ++it) {
        if (it * NW + wave < nxi) {                      // wave-uniform
            const int d = xdesc[it];                      // c | hy << 8 | hx << 16 | invalid << 24
            const int c = d & 0xff, gy = y0 + ((d >> 8) & 0xff) - 1, gx = x0 + ((d >> 16) & 0xff) - 1;
            const bool ok = !(d >> 24) && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const float* src = ok ? ximg + ((long long)c * H + gy) * W + gx : reinterpret_cast<const float*>(fea_zero_line);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sb + DYB + (it * NW + wave) * 256), 4, 0, 0);
        }
    }
}

// Multiply the tile resident in `rd` (rows ty = wave, wave + 4) while the LDS-DMA of the
// next tile lands in `dma`.  __restrict__ on purpose: see wgrad2.hip (keeps hipcc from
// draining vmcnt before every LDS read).
template <int XB, bool BN, int NW>
__device__ __forceinline__ void fw_step(const FirstWgradArgs& g, int m0, int wave, int lane, int next_tile,
                                        char* __restrict__ dma, const char* __restrict__ rd,
                                        const int (&xdesc)[10], int nxi, const int (&boff)[4], int NT, f32x4 (&acc)[4][4]) {
    if constexpr (!BN) { if (next_tile >= 0) fw_issue<false, NW>(g, m0, wave, lane, next_tile, dma, xdesc, nxi); }
    const int r16 = lane & 15, q = lane >> 4;
    const float* hal = reinterpret_cast<const float*>(rd + (BN ? 2 * FW_DY_BYTES : FW_DY_BYTES));
#pragma unroll
    for (int kk = 0; kk < 8 / NW; ++kk) {
        const int ty = wave + NW * kk;
        // A: dy^T fragments of the four 16-channel M tiles, pixels 8q .. 8q+7 of row ty
        u32x4 af[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            s16x4 h2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tx = 8 * q + 4 * t + (r16 >> 2);
                const int unit = m ^ fw_swz(tx);
                h2[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4_t*)(rd + (ty * 32 + tx) * 128 + unit * 32 + (r16 & 3) * 8));
            }
            af[m] = __builtin_bit_cast(u32x4, __builtin_shufflevector(h2[0], h2[1], 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt < NT) {                                // wave-uniform
                bf16x8 bfr;
                const int base = boff[nt] + ty * 34 + 8 * q;
#pragma unroll
                for (int j = 0; j < 8; ++j) bfr[j] = (__bf16)(boff[nt] >= 0 ? hal[base + j] : 0.f);
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[m]), bfr, acc[m][nt], 0, 0, 0);
            }
        }
    }
}

// BN: 8 waves (two per SIMD: the transform is ~800 dependent VALU operations per thread and tile, which one wave per SIMD does not
// hide behind the next tile's loads), each takes one 32-pixel row of the tile as its MFMA k-step
template <int XB, bool BN>
__global__ __launch_bounds__(BN ? 512 : 256) void first_wgrad_kernel(const FirstWgradArgs g) {
    constexpr int NW = BN ? 8 : 4, NT_ = NW * 64;
    constexpr int FW_STAGE = (BN ? 2 : 1) * FW_DY_BYTES + XB;
    static_assert(2 * FW_STAGE >= 4 * 64 * 64 * 4 + (BN ? NT_ * 8 * 4 : 0), "the epilogue sums 64 x 64 fp32 tiles through LDS");
    static_assert(2 * FW_STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(128))) char smem[2 * FW_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int split = blockIdx.x, m0 = blockIdx.y * 64;
    const int C = g.C, NT = (9 * C + 15) >> 4;
    const int nxi = (C * FW_HALO + 63) >> 6;             // 4-byte LDS-DMA wave-instructions for the halo

    int xdesc[10];
#pragma unroll
    for (int it = 0; it < 40 / NW; ++it) {
        const int e = (it * NW + wave) * 64 + lane;
        const int c = e / FW_HALO, rr = e - c * FW_HALO;
        const int hy = rr / 34, hx = rr - hy * 34;
        xdesc[it] = c | (hy << 8) | (hx << 16) | ((c >= C ? 1 : 0) << 24);
    }
    // B column n = 16 nt + r16 = tap*C + c  ->  float offset of (c, dy, dx) in the halo, -1 if n >= 9C
    int boff[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = nt * 16 + r16;
        const int tap = n / C, c = n - tap * C, dy = tap / 3, dx = tap - dy * 3;
        boff[nt] = n < 9 * C ? c * FW_HALO + dy * 34 + dx : -1;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][nt][j] = 0.f;

    // BN: this thread transforms the 16-byte pieces its own wave's LDS-DMA wrote: row (it*NW + wave)*8 + lane/8 of the tile,
    // i.e. tile column tx = (wave*8 + lane/8) & 31 for every `it`, LDS slot lane & 7 = source chunk ^ swizzle(tx): always the
    // same eight channels
    const int txf = (wave * 8 + (lane >> 3)) & 31;
    const int chunkf = (lane & 7) ^ (fw_swz(txf) << 1);
    float sc[8], sh[8], mu[8], is[8], c1[8], c2[8], a0[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = m0 + chunkf * 8 + j;
        sc[j] = BN ? g.sc[ch] : 0.f; sh[j] = BN ? g.sh[ch] : 0.f; mu[j] = BN ? g.mu[ch] : 0.f; is[j] = BN ? g.is[ch] : 0.f;
        c1[j] = BN ? g.coef[2 * ch] : 0.f; c2[j] = BN ? g.coef[2 * ch + 1] : 0.f;
        a0[j] = 0.f;
    }
    const bool partial = (g.H & 7) || (g.W & 31);
    const bool fast = BN && !partial;
    const int row0f = wave * 8 + (lane >> 3);
    const int lo_d = ((row0f >> 5) * g.W + txf) * g.lddy + chunkf * 8, lo_y = ((row0f >> 5) * g.W + txf) * g.ldy + chunkf * 8;

    const int t_begin = split * g.tiles_per_split;
    const int t_end = min(t_begin + g.tiles_per_split, g.ntiles);
    if (t_begin < t_end) fw_issue<BN, NW>(g, m0, wave, lane, t_begin, smem, xdesc, nxi, fast, lo_d, lo_y);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int stage = (tile - t_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // tile landed; the other stage is free
        if constexpr (BN) {
            char* sb = smem + stage * FW_STAGE;
            // the next tile's loads go first: they are in flight during the transform and the multiplication of this one
            if (tile + 1 < t_end) fw_issue<true, NW>(g, m0, wave, lane, tile + 1, smem + (stage ^ 1) * FW_STAGE, xdesc, nxi, fast, lo_d, lo_y);
            int y0 = 0, x0 = 0;
            if (partial) {
                int tt = tile;
                x0 = (tt % g.tiles_x) * 32; tt /= g.tiles_x;
                y0 = (tt % g.tiles_y) * 8;
            }
#pragma unroll
            for (int it = 0; it < 32 / NW; ++it) {
                char* p = sb + (it * NW + wave) * 1024 + lane * 16;
                const bf16x8 gv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
                const bf16x8 yv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + FW_DY_BYTES));
                const int row = (it * NW + wave) * 8 + (lane >> 3);
                const bool ok = !partial || (y0 + (row >> 5) < g.H && x0 + (row & 31) < g.W);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // BatchNorm + ReLU backward, as bn_relu_bwd_kernel<T, false, true> (elementwise.hip) forms it
                    const float yf = (float)yv[j], gf = (float)gv[j];
                    const float act = fmaxf(__builtin_fmaf(yf, sc[j], sh[j]), 0.f);
                    const float gg = act > 0.f ? gf : 0.f;
                    const float xh = (yf - mu[j]) * is[j];
                    float d = sc[j] * (gg - c1[j] - xh * c2[j]);
                    d = ok ? d : 0.f;
                    o[j] = (__bf16)d; a0[j] += d;
                }
                *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, o);
            }
            __syncthreads();             // every wave's rows hold the conv-output gradient
        }
        fw_step<XB, BN, NW>(g, m0, wave, lane, tile + 1 < t_end ? tile + 1 : -1, smem + (stage ^ 1) * FW_STAGE, smem + stage * FW_STAGE,
                        xdesc, nxi, boff, NT, acc);
    }
    __syncthreads();
    // sum the waves: [wave % 4][64 co][64 n] fp32 in LDS (64 KB); with 8 waves the second four add into the first four's slots
    float* ot = reinterpret_cast<float*>(smem);
    if (wave < 4) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) ot[(wave * 64 + m * 16 + 4 * q + j) * 64 + nt * 16 + r16] = acc[m][nt][j];
    }
    float* cb = ot + 4 * 64 * 64;                          // BN: [thread][8] channel sums
    if constexpr (BN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cb[tid * 8 + j] = a0[j];
        __syncthreads();
        if (wave >= 4) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ot[((wave - 4) * 64 + m * 16 + 4 * q + j) * 64 + nt * 16 + r16] += acc[m][nt][j];
        }
    }
    __syncthreads();
    float* dst = g.slab + ((long long)split * g.Cout + m0) * 64;
    for (int i = tid; i < 64 * 64; i += NT_) dst[i] = (ot[i] + ot[4096 + i]) + (ot[8192 + i] + ot[12288 + i]);
    if constexpr (BN) {
        if (tid < 64) {
            // channel tid = chunk cc, element j: held by one thread of every (wave, lane / 8) pair; summed in a fixed order
            const int cc = tid >> 3, j = tid & 7;
            float s = 0.f;
            for (int w_ = 0; w_ < NW; ++w_)
                for (int r = 0; r < 8; ++r) {
                    const int l7 = cc ^ (fw_swz((w_ * 8 + r) & 31) << 1);
                    s += cb[(w_ * 64 + r * 8 + l7) * 8 + j];
                }
            float* pd = g.part + ((long long)split * g.Cout + m0 + tid) * 2;
            pd[0] = s; pd[1] = 0.f;
        }
    }
}

// out[co][c][tap] = sum over splits of slab[split][co][tap*C + c]
__global__ __launch_bounds__(256) void first_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                                 int splits, int Cout, int C) {
    // 16 lanes share an output element (lane j adds splits j, j + 16, ...), their sums are added in lane order: a fixed
    // association, bit-identical from run to run, without a chain of `splits` dependent loads per thread
    const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    __shared__ float part[256];
    float s = 0.f;
    if (i < Cout * 9 * C) {
        const int tap = i % 9, c = (i / 9) % C, co = i / (9 * C);
        const int n = tap * C + c;
        for (int sp = j; sp < splits; sp += 16) s += slab[((long long)sp * Cout + co) * 64 + n];
    }
    part[j * 16 + e] = s;
    __syncthreads();
    if (j == 0 && i < Cout * 9 * C) {
        float r = part[e];
#pragma unroll
        for (int k = 1; k < 16; ++k) r += part[k * 16 + e];
        out[i] = r;
    }
}

struct FwPlan { int tiles_x, tiles_y, ntiles, splits, tiles_per_split; };
static FwPlan fw_plan(int N, int H, int W, int Cout, int per_cu = 2) {
    FwPlan p;
    p.tiles_x = (W + 31) / 32; p.tiles_y = (H + 7) / 8;
    p.ntiles = N * p.tiles_x * p.tiles_y;
    int want = 256 * per_cu / (Cout / 64);
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tiles_per_split = (p.ntiles + want - 1) / want;
    p.splits = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
    return p;
}

extern "C" size_t fea_first_wgrad_ws_bytes(int N, int H, int W, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 64) return 0;
    FwPlan p = fw_plan(N, H, W, Cout);
    return (size_t)p.splits * Cout * 64 * sizeof(float);
}

extern "C" int fea_conv3x3_first_wgrad(const void* dy, int lddy, const float* x, float* dW, void* ws, size_t ws_bytes,
                                       int N, int C, int H, int W, int Cout, int dtype, void* stream) {
    FEA_CHECK(dtype == FEA_BF16, FEA_E_UNSUPPORTED_SHAPE, "fea_conv3x3_first_wgrad: bf16 only (fp32: fea_conv1x1_im2col_wgrad)");
    FEA_CHECK(dy && x && dW && ws, FEA_E_BADARG, "fea_conv3x3_first_wgrad: null pointer");
    FEA_CHECK(N > 0 && H > 0 && W > 0 && C > 0 && 9 * C <= 64, FEA_E_UNSUPPORTED_SHAPE, "fea_conv3x3_first_wgrad: N=%d H=%d W=%d C=%d", N, H, W, C);
    FEA_CHECK(Cout > 0 && Cout % 64 == 0 && lddy >= Cout && lddy % 8 == 0 && ((uintptr_t)dy % 16 == 0), FEA_E_BADARG,
              "fea_conv3x3_first_wgrad: Cout=%d lddy=%d", Cout, lddy);
    FEA_CHECK((long long)H * W * lddy < (1ll << 31) && (long long)C * H * W < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE,
              "fea_conv3x3_first_wgrad: image too large");
    FwPlan p = fw_plan(N, H, W, Cout);
    const size_t need = (size_t)p.splits * Cout * 64 * sizeof(float);
    FEA_CHECK(ws_bytes >= need, FEA_E_WORKSPACE, "fea_conv3x3_first_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    FirstWgradArgs a{};
    a.dy = (const __bf16*)dy; a.x = x; a.slab = (float*)ws; a.lddy = lddy; a.C = C; a.H = H; a.W = W; a.Cout = Cout;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split;
    hipStream_t s = (hipStream_t)stream;
    if (C <= 3) hipLaunchKernelGGL((first_wgrad_kernel<4096, false>), dim3((unsigned)p.splits, (unsigned)(Cout / 64)), dim3(256), 0, s, a);
    else        hipLaunchKernelGGL((first_wgrad_kernel<10240, false>), dim3((unsigned)p.splits, (unsigned)(Cout / 64)), dim3(256), 0, s, a);
    FEA_LAUNCH_CHECK();
    const int total = Cout * 9 * C;
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s,
                       (const float*)ws, dW, p.splits, Cout, C);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

// rows of `part` that fea_conv3x3_first_wgrad_bnbwd writes (one per pixel split)
extern "C" int fea_first_wgrad_bnbwd_rows(int N, int H, int W, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 64) return 0;
    return fw_plan(N, H, W, Cout, 1).splits;
}

// BatchNorm + ReLU backward of enc1[1..2] fused into the first layer's weight gradient: the conv-output gradient
// dy = scale * (g - c1 - xhat * c2), g = da * [relu(bn(y)) > 0] (what fea_bn_relu_bwd_apply writes) is formed in LDS from da and
// y and never goes to HBM -- the first layer has no data gradient, so the weight gradient is dy's only reader.
extern "C" int fea_conv3x3_first_wgrad_bnbwd(const void* da, int ldda, const void* y, int ldy, const float* scale, const float* shift,
                                             const float* save_mean, const float* save_invstd, const float* coef,
                                             const float* x, float* dW, float* part, void* ws, size_t ws_bytes,
                                             int N, int C, int H, int W, int Cout, int dtype, void* stream) {
    const char* who = "fea_conv3x3_first_wgrad_bnbwd";
    FEA_CHECK(dtype == FEA_BF16, FEA_E_UNSUPPORTED_SHAPE, "%s: bf16 only", who);
    FEA_CHECK(da && y && scale && shift && save_mean && save_invstd && coef && x && dW && part && ws, FEA_E_BADARG, "%s: null pointer", who);
    FEA_CHECK(N > 0 && H > 0 && W > 0 && C > 0 && 9 * C <= 64, FEA_E_UNSUPPORTED_SHAPE, "%s: N=%d H=%d W=%d C=%d", who, N, H, W, C);
    FEA_CHECK(Cout > 0 && Cout % 64 == 0 && ldda >= Cout && ldda % 8 == 0 && ldy >= Cout && ldy % 8 == 0 &&
              ((uintptr_t)da % 16 == 0) && ((uintptr_t)y % 16 == 0), FEA_E_BADARG, "%s: Cout=%d ldda=%d ldy=%d", who, Cout, ldda, ldy);
    FEA_CHECK((long long)H * W * ldda < (1ll << 31) && (long long)H * W * ldy < (1ll << 31) && (long long)C * H * W < (1ll << 31),
              FEA_E_UNSUPPORTED_SHAPE, "%s: image too large", who);
    FwPlan p = fw_plan(N, H, W, Cout, 1);
    const size_t need = (size_t)p.splits * Cout * 64 * sizeof(float);
    FEA_CHECK(ws_bytes >= need, FEA_E_WORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
    FirstWgradArgs a{};
    a.dy = (const __bf16*)da; a.x = x; a.slab = (float*)ws; a.lddy = ldda; a.C = C; a.H = H; a.W = W; a.Cout = Cout;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split;
    a.ybn = (const __bf16*)y; a.ldy = ldy; a.sc = scale; a.sh = shift; a.mu = save_mean; a.is = save_invstd; a.coef = coef; a.part = part;
    hipStream_t s = (hipStream_t)stream;
    if (C <= 3) hipLaunchKernelGGL((first_wgrad_kernel<4096, true>), dim3((unsigned)p.splits, (unsigned)(Cout / 64)), dim3(512), 0, s, a);
    else        hipLaunchKernelGGL((first_wgrad_kernel<10240, true>), dim3((unsigned)p.splits, (unsigned)(Cout / 64)), dim3(512), 0, s, a);
    FEA_LAUNCH_CHECK();
    const int total = Cout * 9 * C;
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s,
                       (const float*)ws, dW, p.splits, Cout, C);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}
