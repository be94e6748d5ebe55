// Weight-gradient kernels (MFMA, gfx950): D[tap][m][n] = sum_p P[p][m] * Q[shift_tap(p)][n]
//
// The reduction runs over PIXELS, which is the slow dimension of an NHWC
// tensor, so both MFMA operands need k (= pixel) contiguous per lane while
// memory has channels contiguous.  Tiles are staged in LDS as [pixel][64
// channels] rows; bf16 operands are then fetched with the transposing LDS read
// (ds_read_b64_tr_b16: 4 pixels x 16 channels per 16-lane group, delivered
// channel-major), fp32 operands with plain ds_read_b32 (one k per lane).
//
//   MODE 0  conv3x3 (models.py:28,31): P = dy (m = co), Q = x halo tile (n = ci),
//           nine accumulator tiles per wave share one staged (P, Q) pair.
//   MODE 1  1 tap, Q at the same pixel (enc1.0 on im2col rows).
//   MODE 2  ConvTranspose2d 2x2/s2 (models.py:15-21): P = x (m = ci), Q = dy at
//           (2i+a, 2j+b) (n = co); tap = blockIdx.y.
// Work is split over pixel tiles; every block writes its partial 64x64(x taps)
// result to an fp32 slab and `wgrad_reduce_kernel` sums the slabs in a fixed
// order (deterministic) while scattering into the parameter's own layout.
#include "common.h"

struct WgradArgs {
    const void* p; const void* q; float* slab;
    long long p_img_stride, q_img_stride;
    int ldp, ldq;
    int H, W;                 // P's pixel grid
    int M, Nn;                // multiples of 64
    int tiles_x, tiles_y, ntiles, tiles_per_split;
    int nmt, nnt, ntaps;
};

__device__ __forceinline__ int swzw(int x) {      // bit-reverse of bits 1..3 of x
    const int u = (x >> 1) & 7;
    return ((u & 1) << 2) | (u & 2) | (u >> 2);
}

template <typename T, int MODE, int TWS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs g) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int ROWB = 64 * sizeof(T);
    constexpr int CPR = ROWB / 16;               // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;               // rows staged per pass
    constexpr int EPC = 16 / sizeof(T);
    constexpr int PT = BF ? 256 : 128;
    constexpr int TW = 1 << TWS, TH = PT / TW, HW2 = TW + 2;
    constexpr int QROWS_RAW = MODE == 0 ? (TH + 2) * HW2 : PT;
    constexpr int QROWS = (QROWS_RAW + RPP - 1) / RPP * RPP;
    constexpr int P_PASSES = PT / RPP, Q_PASSES = QROWS / RPP;
    constexpr int NTAP = MODE == 0 ? 9 : 1;
    constexpr int XSEG = TW / 16;
    __shared__ __attribute__((aligned(16))) char smem[(PT + QROWS) * ROWB];
    char* const ldsP = smem;
    char* const ldsQ = smem + PT * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntp = g.nmt * g.nnt;
    const int tp = lid % ntp, split = lid / ntp;
    const int m0 = (tp / g.nnt) * 64, n0 = (tp % g.nnt) * 64;
    const int tap_z = MODE == 2 ? blockIdx.y : 0;
    const int H = g.H, W = g.W;

    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- per-lane fragment offsets (bytes inside the LDS images) ----------
    // bf16: transposing reads.  16-lane group gi = lane>>4: channel half gi&1,
    // lane 4q+pp of the group addresses pixel-row q, channels 4pp..4pp+3.
    int pcol[XSEG][2];
    int qcol[XSEG][2][MODE == 0 ? 3 : 1];
    int fp_p = 0, fp_q = 0;
    if constexpr (BF) {
        const int gi = lane >> 4, li = lane & 15;
        const int chl = 16 * (gi & 1) + 4 * (li & 3);      // channel inside the wave's 32
        const int qrow = li >> 2;
#pragma unroll
        for (int xs = 0; xs < XSEG; ++xs)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int x = xs * 16 + 8 * h + 4 * t + qrow;
                const int chp = wa * 32 + chl, chq = wb * 32 + chl;
                pcol[xs][t] = x * ROWB + ((((chp >> 3) ^ swzw(x))) << 4) + (chp & 7) * 2;
                if (MODE == 0) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int hx = x + dx;
                        qcol[xs][t][dx] = hx * ROWB + ((((chq >> 3) ^ swzw(hx))) << 4) + (chq & 7) * 2;
                    }
                } else {
                    qcol[xs][t][0] = x * ROWB + ((((chq >> 3) ^ swzw(x))) << 4) + (chq & 7) * 2;
                }
            }
    } else {
        fp_p = h * ROWB + (wa * 32 + r) * 4;
        fp_q = h * ROWB + (wb * 32 + r) * 4;
    }

    const int sc = tid % CPR;            // chunk this thread stages
    const int srow = tid / CPR;
    const T* __restrict__ pg = (const T*)g.p;
    const T* __restrict__ qg = (const T*)g.q;

    const int t_begin = split * g.tiles_per_split;
    const int t_end = min(t_begin + g.tiles_per_split, g.ntiles);
    for (int tile = t_begin; tile < t_end; ++tile) {
        int tt = tile;
        const int tile_x = tt % g.tiles_x; tt /= g.tiles_x;
        const int tile_y = tt % g.tiles_y;
        const int n_img = tt / g.tiles_y;
        const int y0 = tile_y * TH, x0 = tile_x * TW;
        const T* pimg = pg + (long long)n_img * g.p_img_stride + m0;
        const T* qimg = qg + (long long)n_img * g.q_img_stride + n0;

        __syncthreads();                 // previous tile fully consumed
        {   // ---- stage P: PT tile-linear pixel rows x 64 channels ----------
            u32x4 v[P_PASSES];
            unsigned okm = 0;
#pragma unroll
            for (int i = 0; i < P_PASSES; ++i) {
                const int row = i * RPP + srow;
                const int gy = y0 + (row >> TWS), gx = x0 + (row & (TW - 1));
                const bool ok = gy < H && gx < W;
                okm |= (ok ? 1u : 0u) << i;
                const int off = ok ? (gy * W + gx) * g.ldp + sc * EPC : 0;
                v[i] = *reinterpret_cast<const u32x4*>(pimg + off);
            }
#pragma unroll
            for (int i = 0; i < P_PASSES; ++i) {
                const int row = i * RPP + srow;
                const int csw = BF ? (sc ^ swzw(row & (TW - 1))) : sc;
                u32x4 z = {0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(ldsP + row * ROWB + (csw << 4)) = ((okm >> i) & 1) ? v[i] : z;
            }
        }
        {   // ---- stage Q ----------------------------------------------------
            constexpr int HALF = (Q_PASSES + 1) / 2;
#pragma unroll
            for (int b0 = 0; b0 < Q_PASSES; b0 += HALF) {
                u32x4 v[HALF];
                unsigned okm = 0;
                int keyv[HALF];
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    if (b0 + i < Q_PASSES) {
                        const int row = (b0 + i) * RPP + srow;
                        int off; bool ok;
                        if (MODE == 0) {
                            const int hy = row / HW2, hx = row - hy * HW2;
                            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
                            ok = hy < TH + 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                            off = (gy * W + gx) * g.ldq;
                            keyv[i] = hx;
                        } else {
                            const int gy = y0 + (row >> TWS), gx = x0 + (row & (TW - 1));
                            ok = gy < H && gx < W;
                            off = MODE == 1 ? (gy * W + gx) * g.ldq
                                            : ((2 * gy + (tap_z >> 1)) * (2 * W) + 2 * gx + (tap_z & 1)) * g.ldq;
                            keyv[i] = row & (TW - 1);
                        }
                        okm |= (ok ? 1u : 0u) << i;
                        v[i] = *reinterpret_cast<const u32x4*>(qimg + (ok ? off + sc * EPC : 0));
                    }
                }
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    if (b0 + i < Q_PASSES) {
                        const int row = (b0 + i) * RPP + srow;
                        const int csw = BF ? (sc ^ swzw(keyv[i])) : sc;
                        u32x4 z = {0u, 0u, 0u, 0u};
                        *reinterpret_cast<u32x4*>(ldsQ + row * ROWB + (csw << 4)) = ((okm >> i) & 1) ? v[i] : z;
                    }
                }
            }
        }
        __syncthreads();

        // ---- k loop over the tile's pixels, 16 per step ---------------------
#pragma unroll 1
        for (int ty = 0; ty < TH; ++ty) {
            const char* prow = ldsP + ty * TW * ROWB;
            const char* qrow0 = ldsQ + (MODE == 0 ? ty * HW2 * ROWB : ty * TW * ROWB);
#pragma unroll
            for (int xs = 0; xs < XSEG; ++xs) {
                if constexpr (BF) {
                    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(prow + pcol[xs][0]));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(prow + pcol[xs][1]));
                    const u32x4 af = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int tap = 0; tap < NTAP; ++tap) {
                        const int dy = tap / 3, dx = tap % 3;
                        const char* qr = qrow0 + dy * HW2 * ROWB;
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(qr + qcol[xs][0][MODE == 0 ? dx : 0]));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(qr + qcol[xs][1][MODE == 0 ? dx : 0]));
                        const u32x4 bfr = __builtin_bit_cast(u32x4, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                        mma16(acc[tap], af, bfr, (const __bf16*)nullptr);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float a = *reinterpret_cast<const float*>(prow + (xs * 16 + 2 * t) * ROWB + fp_p);
#pragma unroll
                        for (int tap = 0; tap < NTAP; ++tap) {
                            const int dy = tap / 3, dx = tap % 3;
                            const float b = *reinterpret_cast<const float*>(
                                qrow0 + (MODE == 0 ? (dy * HW2 + dx) : 0) * ROWB + (xs * 16 + 2 * t) * ROWB + fp_q);
                            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tap], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- write the partial slab: [split][tap][M][Nn] ------------------------
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
        const int tg = MODE == 2 ? tap_z : tap;
        float* dst = g.slab + (((long long)split * g.ntaps + tg) * g.M + m0 + wa * 32) * g.Nn + n0 + wb * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int mr = (e & 3) + 8 * (e >> 2) + 4 * h;
            dst[(long long)mr * g.Nn] = acc[tap][e];
        }
    }
}

// out index: kind 0: (m*Nn + n)*ntaps + tap      (OIHW conv3x3 / [Ci,Co,2,2] upconv)
//            kind 1: n = tap*Creal + c (n < 9*Creal): (m*Creal + c)*9 + tap  (enc1.0 via im2col)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int splits, int ntaps,
                                    int M, int Nn, int kind, int Creal) {
    const long long per = (long long)ntaps * M * Nn;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += slab[sp * per + i];
    const int n = (int)(i % Nn);
    const int m = (int)((i / Nn) % M);
    const int tap = (int)(i / ((long long)Nn * M));
    if (kind == 0) {
        out[((long long)m * Nn + n) * ntaps + tap] = s;
    } else if (n < 9 * Creal) {
        const int t9 = n / Creal, c = n - t9 * Creal;
        out[((long long)m * Creal + c) * 9 + t9] = s;
    }
}

// ---------------------------------------------------------------------------
struct WgradPlan { int tws, TW, TH, tiles_x, tiles_y, ntiles, splits, tiles_per_split, ntaps, nmt, nnt; };

static WgradPlan wgrad_plan(int mode, int N, int H, int W, int M, int Nn, int dtype) {
    WgradPlan p;
    const int PT = dtype == FEA_BF16 ? 256 : 128;
    p.tws = W >= 32 ? 5 : 4; p.TW = 1 << p.tws; p.TH = PT / p.TW;
    p.tiles_x = (W + p.TW - 1) / p.TW; p.tiles_y = (H + p.TH - 1) / p.TH;
    p.ntiles = N * p.tiles_x * p.tiles_y;
    p.ntaps = mode == 0 ? 9 : (mode == 2 ? 4 : 1);
    p.nmt = M / 64; p.nnt = Nn / 64;
    const int zdim = mode == 2 ? 4 : 1;
    int want = 2048 / (p.nmt * p.nnt * zdim);
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tiles_per_split = (p.ntiles + want - 1) / want;
    p.splits = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
    return p;
}

// bf16 conv3x3 path: wgrad2.hip
size_t fea_wgrad2_ws_bytes(int N, int H, int W, int Cm, int Cn);
int fea_wgrad2_launch(const void* dy, int lddy, const void* x, int ldx, float* dW, void* ws, size_t ws_bytes,
                      int N, int H, int W, int Cin, int Cout, hipStream_t s);

// bf16 ConvTranspose2d path: wgrad_up.hip
bool fea_wgrad_up_supported(int Cin, int Cout, int dtype);
size_t fea_wgrad_up_ws_bytes(int N, int H, int W, int Cin, int Cout);
int fea_wgrad_up_launch(const void* x, int ldx, const void* dy, int lddy, float* dW, void* ws, size_t ws_bytes,
                        int N, int H, int W, int Cin, int Cout, hipStream_t s);

extern "C" size_t fea_wgrad_ws_bytes(int mode, int N, int H, int W, int Cm, int Cn, int dtype) {
    if (N <= 0 || H <= 0 || W <= 0 || Cm <= 0 || Cn <= 0 || Cm % 64 || Cn % 64) return 0;
    if (mode == 2 && fea_wgrad_up_supported(Cm, Cn, dtype)) { const size_t b = fea_wgrad_up_ws_bytes(N, H, W, Cm, Cn); return b ? b : 256; }
    if (mode == 0 && dtype == FEA_BF16) { const size_t b = fea_wgrad2_ws_bytes(N, H, W, Cm, Cn); return b ? b : 256; }
    WgradPlan p = wgrad_plan(mode, N, H, W, Cm, Cn, dtype);
    return (size_t)p.splits * p.ntaps * Cm * Cn * sizeof(float);
}

template <typename T, int MODE>
static int launch_wgrad(WgradArgs& a, const WgradPlan& p, hipStream_t s) {
    dim3 grid((unsigned)(p.nmt * p.nnt * p.splits), MODE == 2 ? 4 : 1), block(256);
    if (p.tws == 5) hipLaunchKernelGGL((wgrad_kernel<T, MODE, 5>), grid, block, 0, s, a);
    else            hipLaunchKernelGGL((wgrad_kernel<T, MODE, 4>), grid, block, 0, s, a);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

static int wgrad_common(int mode, const void* P, int ldp, const void* Q, int ldq, float* dW, void* ws, size_t ws_bytes,
                        int N, int H, int W, int M, int Nn, long long p_img, long long q_img, int kind, int Creal,
                        int dtype, void* stream, const char* who) {
    FEA_CHECK(dtype == FEA_BF16 || dtype == FEA_F32, FEA_E_BADARG, "%s: bad dtype %d", who, dtype);
    FEA_CHECK(P && Q && dW && ws, FEA_E_BADARG, "%s: null pointer", who);
    FEA_CHECK(N > 0 && H > 0 && W > 0, FEA_E_BADARG, "%s: bad dims", who);
    FEA_CHECK(M > 0 && Nn > 0 && M % 64 == 0 && Nn % 64 == 0, FEA_E_UNSUPPORTED_SHAPE,
              "%s: channel counts %d,%d must be multiples of 64", who, M, Nn);
    FEA_CHECK(ldp >= M && ldq >= Nn && ldp % 8 == 0 && ldq % 8 == 0, FEA_E_BADARG, "%s: ld %d,%d", who, ldp, ldq);
    FEA_CHECK(((uintptr_t)P % 16 == 0) && ((uintptr_t)Q % 16 == 0), FEA_E_BADARG, "%s: pointers must be 16-byte aligned", who);
    FEA_CHECK(p_img < (1ll << 31) && q_img < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "%s: image too large", who);
    if (mode == 0 && dtype == FEA_BF16)
        return fea_wgrad2_launch(P, ldp, Q, ldq, dW, ws, ws_bytes, N, H, W, Nn, M, (hipStream_t)stream);
    if (mode == 2 && fea_wgrad_up_supported(M, Nn, dtype))
        return fea_wgrad_up_launch(P, ldp, Q, ldq, dW, ws, ws_bytes, N, H, W, M, Nn, (hipStream_t)stream);
    WgradPlan p = wgrad_plan(mode, N, H, W, M, Nn, dtype);
    const size_t need = (size_t)p.splits * p.ntaps * M * Nn * sizeof(float);
    FEA_CHECK(ws_bytes >= need, FEA_E_WORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
    WgradArgs a{};
    a.p = P; a.q = Q; a.slab = (float*)ws; a.ldp = ldp; a.ldq = ldq; a.H = H; a.W = W; a.M = M; a.Nn = Nn;
    a.p_img_stride = p_img; a.q_img_stride = q_img;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split;
    a.nmt = p.nmt; a.nnt = p.nnt; a.ntaps = p.ntaps;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (dtype == FEA_BF16) {
        rc = mode == 0 ? launch_wgrad<__bf16, 0>(a, p, s) : mode == 1 ? launch_wgrad<__bf16, 1>(a, p, s) : launch_wgrad<__bf16, 2>(a, p, s);
    } else {
        rc = mode == 0 ? launch_wgrad<float, 0>(a, p, s) : mode == 1 ? launch_wgrad<float, 1>(a, p, s) : launch_wgrad<float, 2>(a, p, s);
    }
    if (rc) return rc;
    const long long per = (long long)p.ntaps * M * Nn;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s,
                       (const float*)ws, dW, p.splits, p.ntaps, M, Nn, kind, Creal);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_conv3x3_wgrad(const void* dy, int lddy, const void* x, int ldx, float* dW, void* ws, size_t ws_bytes,
                                 int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    return wgrad_common(0, dy, lddy, x, ldx, dW, ws, ws_bytes, N, H, W, Cout, Cin,
                        (long long)H * W * lddy, (long long)H * W * ldx, 0, 0, dtype, stream, "fea_conv3x3_wgrad");
}

extern "C" int fea_conv1x1_im2col_wgrad(const void* dy, int lddy, const void* xcol, int ldx, float* dW, void* ws, size_t ws_bytes,
                                        int N, int H, int W, int Kpad, int Creal, int Cout, int dtype, void* stream) {
    FEA_CHECK(Creal > 0 && 9 * Creal <= Kpad, FEA_E_BADARG, "fea_conv1x1_im2col_wgrad: Creal=%d Kpad=%d", Creal, Kpad);
    return wgrad_common(1, dy, lddy, xcol, ldx, dW, ws, ws_bytes, N, H, W, Cout, Kpad,
                        (long long)H * W * lddy, (long long)H * W * ldx, 1, Creal, dtype, stream, "fea_conv1x1_im2col_wgrad");
}

extern "C" int fea_upconv2x2_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dW, void* ws, size_t ws_bytes,
                                   int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    return wgrad_common(2, x, ldx, dy, lddy, dW, ws, ws_bytes, N, H, W, Cin, Cout,
                        (long long)H * W * ldx, (long long)4 * H * W * lddy, 0, 0, dtype, stream, "fea_upconv2x2_wgrad");
}
