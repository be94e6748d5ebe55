// Input pipeline on the GPU: the reference's per-image transform
// (custom_dataset.py:23-28, applied at :69-71)
//
//     Resize((S,S)) on a PIL image  ->  ToTensor()  ->  Normalize(mean, std)
//
// as two kernels over a decoded 8-bit image resident in HBM:
//   1. horizontal pass   src u8 [h][w][C]   -> tmp u8 [h][W][C]
//   2. vertical pass     tmp                -> one of
//        EPI 0  u8  [H][W][C]                    (plain resize)
//        EPI 1  f32 [C][H][W] = lut[c][v]        (ToTensor + Normalize: the module's NCHW fp32 input)
//        EPI 2  u8  [C][H][W] = v > threshold    (mask target)
// The arithmetic is Pillow's antialiased bilinear resample (fixed point, 22 fractional
// bits, uint8 rounding between the passes), so the result is bit-identical to the
// reference's CPU path.  Coefficient tables are computed on the host in double precision
// (fea_resize_coeffs) exactly as Pillow does, the kernels only do integer MACs.
// HBM-bound byte work: one thread per output pixel, consecutive threads = consecutive
// output x (coalesced writes in both passes, coalesced reads in the vertical pass).
#include "common.h"
#include <math.h>

constexpr int RS_BITS = 32 - 8 - 2;

template <int C>
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, int src_w, int rows,
                                                       uint8_t* __restrict__ tmp, int W,
                                                       const int* __restrict__ bounds, const int* __restrict__ coef, int ks) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)rows * W) return;
    const int x = (int)(idx % W);
    const long long row = idx / W;
    const int x0 = bounds[2 * x], n = bounds[2 * x + 1];
    const int* k = coef + (long long)x * ks;
    const uint8_t* p = src + (row * src_w + x0) * C;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (RS_BITS - 1);
    for (int t = 0; t < n; ++t) {
        const int kv = k[t];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += (int)p[t * C + c] * kv;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int v = acc[c] >> RS_BITS;
        tmp[idx * C + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

template <int C, int EPI>
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ tmp, int W, int H, void* __restrict__ dst,
                                                       const int* __restrict__ bounds, const int* __restrict__ coef, int ks,
                                                       const float* __restrict__ lut, int threshold) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)H * W) return;
    const int x = (int)(idx % W);
    const int y = (int)(idx / W);
    const int y0 = bounds[2 * y], n = bounds[2 * y + 1];
    const int* k = coef + (long long)y * ks;
    const uint8_t* p = tmp + ((long long)y0 * W + x) * C;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (RS_BITS - 1);
    for (int t = 0; t < n; ++t) {
        const int kv = k[t];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += (int)p[(long long)t * W * C + c] * kv;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        int v = acc[c] >> RS_BITS;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        if (EPI == 0) static_cast<uint8_t*>(dst)[idx * C + c] = (uint8_t)v;
        else if (EPI == 1) static_cast<float*>(dst)[(long long)c * H * W + idx] = lut[c * 256 + v];
        else static_cast<uint8_t*>(dst)[(long long)c * H * W + idx] = v > threshold ? 1 : 0;
    }
}

// ---- host side -------------------------------------------------------------------------
static double rs_scale(int in_size, int out_size) { return (double)in_size / (double)out_size; }

extern "C" int fea_resize_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return FEA_E_BADARG;
    double fs = rs_scale(in_size, out_size);
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(fs) * 2 + 1;
}

extern "C" int fea_resize_coeffs(int in_size, int out_size, int* bounds, int* coeffs) {
    FEA_CHECK(in_size > 0 && out_size > 0 && bounds && coeffs, FEA_E_BADARG, "fea_resize_coeffs: bad argument");
    const double scale = rs_scale(in_size, out_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;             // bilinear
    const int ks = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    double w[ks];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w[x] = a < 1.0 ? 1.0 - a : 0.0;
            ww += w[x];
        }
        int* k = coeffs + (long long)xx * ks;
        for (int x = 0; x < ks; ++x) {
            if (x < xmax) {
                const double v = ww != 0.0 ? w[x] / ww : w[x];
                k[x] = v < 0 ? (int)(-0.5 + v * (1 << RS_BITS)) : (int)(0.5 + v * (1 << RS_BITS));
            } else k[x] = 0;
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ks;
}

extern "C" int fea_normalize_lut(const float* mean, const float* stdv, int C, float* lut) {
    FEA_CHECK(mean && stdv && lut && C > 0, FEA_E_BADARG, "fea_normalize_lut: bad argument");
    for (int c = 0; c < C; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float t = (float)v / 255.0f;          // ToTensor: u8 -> f32, / 255 (one rounding each)
            volatile float u = t - mean[c];                // Normalize: sub, then div
            lut[c * 256 + v] = u / stdv[c];
        }
    return FEA_OK;
}

extern "C" int fea_resize_u8(const uint8_t* src, int src_h, int src_w, int C, uint8_t* tmp, void* dst, int H, int W,
                             const int* xbounds, const int* xcoef, int xks, const int* ybounds, const int* ycoef, int yks,
                             int epilogue, const float* lut, int threshold, void* stream) {
    const char* who = "fea_resize_u8";
    FEA_CHECK(src && tmp && dst && xbounds && xcoef && ybounds && ycoef, FEA_E_BADARG, "%s: null pointer", who);
    FEA_CHECK(src_h > 0 && src_w > 0 && H > 0 && W > 0, FEA_E_BADARG, "%s: bad size", who);
    FEA_CHECK(C == 1 || C == 3, FEA_E_UNSUPPORTED_SHAPE, "%s: C must be 1 or 3 (got %d)", who, C);
    FEA_CHECK(epilogue >= 0 && epilogue <= 2, FEA_E_BADARG, "%s: epilogue must be 0, 1 or 2", who);
    FEA_CHECK(epilogue != 1 || lut, FEA_E_BADARG, "%s: epilogue 1 needs a lut", who);
    FEA_CHECK(xks == fea_resize_ksize(src_w, W) && yks == fea_resize_ksize(src_h, H), FEA_E_BADARG,
              "%s: coefficient tables do not belong to these sizes", who);
    hipStream_t s = (hipStream_t)stream;
    const long long n1 = (long long)src_h * W, n2 = (long long)H * W;
    const unsigned g1 = (unsigned)((n1 + 255) / 256), g2 = (unsigned)((n2 + 255) / 256);
    if (C == 3) hipLaunchKernelGGL((resize_h_kernel<3>), dim3(g1), dim3(256), 0, s, src, src_w, src_h, tmp, W, xbounds, xcoef, xks);
    else hipLaunchKernelGGL((resize_h_kernel<1>), dim3(g1), dim3(256), 0, s, src, src_w, src_h, tmp, W, xbounds, xcoef, xks);
    FEA_LAUNCH_CHECK();
#define RSV(CC, EE) hipLaunchKernelGGL((resize_v_kernel<CC, EE>), dim3(g2), dim3(256), 0, s, tmp, W, H, dst, ybounds, ycoef, yks, lut, threshold)
    if (C == 3) { if (epilogue == 0) RSV(3, 0); else if (epilogue == 1) RSV(3, 1); else RSV(3, 2); }
    else        { if (epilogue == 0) RSV(1, 0); else if (epilogue == 1) RSV(1, 1); else RSV(1, 2); }
#undef RSV
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}
