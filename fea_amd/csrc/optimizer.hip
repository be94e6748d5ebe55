// Optimiser step of the training loop (reference train_direct.py:28-32 AdamW(lr, weight_decay),
// :105 clip_grad_norm_(params, 1.0), :106 optimizer.step()) over the engine's flat gradient buffer:
//   1. fea_grad_sumsq      sum of squares of all gradients (fixed-order reduction, fp64 total)
//   2. fea_adamw_clip_step every parameter tensor updated in ONE launch; the clip factor
//      min(1, max_norm / (||g|| + 1e-6)) is applied to the gradient on the fly, so the gradient
//      buffer is read once and never rewritten.
// HBM-bound: per element read g, p, m, v (16 B) + write p, m, v (12 B).
#include "common.h"
#include <math.h>

constexpr int OPT_CHUNK = 8192;          // elements per workgroup (256 threads x 8 float4)

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
    const long long i0 = (long long)blockIdx.x * OPT_CHUNK;
    const long long i1 = i0 + OPT_CHUNK < n ? i0 + OPT_CHUNK : n;
    float s = 0.f;
    if ((i1 - i0) == OPT_CHUNK) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + i0 + (k * 256 + threadIdx.x) * 4);
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += 256) s += g[i] * g[i];
    }
    __shared__ float sh[4];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nblk, double* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)part[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

struct AdamArgs {
    const long long* seg;        // [nseg][3]: parameter chunk address, offset in the flat buffers, element count
    const float* g; float* m; float* v;
    const double* sumsq;
    float max_norm, decay, beta1, beta2, omb1, omb2, eps, step_size, bc2_sqrt;   // decay = 1 - lr*wd, omb = 1 - beta
};

// torch.optim.AdamW arithmetic (decoupled decay, lerp for the first moment), fp32 throughout
__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
    g *= coef;
    p *= a.decay;
    m = m + a.omb1 * (g - m);
    v = a.beta2 * v + a.omb2 * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= a.step_size * m / denom;
}

__global__ __launch_bounds__(256) void adamw_clip_kernel(const AdamArgs a) {
    const long long* sg = a.seg + (long long)blockIdx.x * 3;
    float* __restrict__ p = reinterpret_cast<float*>(sg[0]);
    const long long off = sg[1];
    const int cnt = (int)sg[2];
    const float nrm = (float)sqrt(a.sumsq[0]);
    const float coef = a.max_norm > 0.f ? fminf(a.max_norm / (nrm + 1e-6f), 1.f) : 1.f;
    const float* __restrict__ g = a.g + off;
    float* __restrict__ m = a.m + off;
    float* __restrict__ v = a.v + off;
    const bool vec = ((uintptr_t)p % 16 == 0) && (off % 4 == 0);
    const int n4 = vec ? cnt / 4 : 0;
    for (int i = threadIdx.x; i < n4; i += 256) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pe = pp[k], me = mm[k], ve = vv[k];
            adamw_elem(pe, gg[k], me, ve, a, coef);
            pp[k] = pe; mm[k] = me; vv[k] = ve;
        }
        reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    for (int i = n4 * 4 + threadIdx.x; i < cnt; i += 256) adamw_elem(p[i], g[i], m[i], v[i], a, coef);
}

extern "C" int fea_opt_chunk(void) { return OPT_CHUNK; }

extern "C" int fea_sumsq_blocks(long long n) { return n > 0 ? (int)((n + OPT_CHUNK - 1) / OPT_CHUNK) : 0; }

extern "C" int fea_grad_sumsq(const float* g, long long n, float* part, double* sumsq, void* stream) {
    FEA_CHECK(g && part && sumsq && n > 0, FEA_E_BADARG, "fea_grad_sumsq: bad args");
    FEA_CHECK((uintptr_t)g % 16 == 0, FEA_E_BADARG, "fea_grad_sumsq: g must be 16-byte aligned");
    const int nblk = fea_sumsq_blocks(n);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(256), 0, s, g, n, part);
    FEA_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, (const float*)part, nblk, sumsq);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_adamw_clip_step(const long long* seg, int nseg, const float* g_flat, float* m_flat, float* v_flat,
                                   const double* sumsq, double max_norm, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, int step, void* stream) {
    FEA_CHECK(seg && nseg > 0 && g_flat && m_flat && v_flat && sumsq, FEA_E_BADARG, "fea_adamw_clip_step: bad args");
    FEA_CHECK(step >= 1 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, FEA_E_BADARG, "fea_adamw_clip_step: step=%d betas=(%g,%g)", step, beta1, beta2);
    AdamArgs a{};
    a.seg = seg; a.g = g_flat; a.m = m_flat; a.v = v_flat; a.sumsq = sumsq;
    a.max_norm = (float)max_norm; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps;
    // every derived constant as torch computes it: on the host in double (Python floats), then rounded to
    // fp32 once -- 1.f - 0.999f is 4.7e-5 away from (float)(1 - 0.999)
    a.decay = (float)(1.0 - lr * weight_decay);
    a.omb1 = (float)(1.0 - beta1);
    a.omb2 = (float)(1.0 - beta2);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adamw_clip_kernel, dim3((unsigned)nseg), dim3(256), 0, (hipStream_t)stream, a);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}
