// ConvTranspose2d(ci, co, 2, stride=2) weight gradient, bf16, LDS-DMA pipelined
// (models.py:15-21 backward):
//
//   dW[ci][co][a][b] = sum_{n,i,j} x[n,i,j,ci] * dy[n,2i+a,2j+b,co]
//
// (XQ = 4, Cin a multiple of 256: the workgroup owns 256 input channels -- 64 KB per tile for 2,048 MFMA cycles instead of 48 KB
// per 1,024: the kernel is bound by the CU's LDS fill rate, ~13-30 B/clk (MI355X_MICROARCH.md, gather into LDS), so bytes per
// MFMA are what counts; a wave then multiplies 128 x 64, six fragments per eight MFMAs instead of four per four; two stages.)
//
// Four independent pixel-reduction GEMMs (one per tap (a,b)) that share the x operand.
// The op has little arithmetic per byte (1.33*Cout FLOP per byte at best), so the tile is
// chosen to stage each byte as few times as possible: one 512-thread workgroup owns
// 128 input channels x 64 output channels x all 4 taps and walks a range of 4x16-pixel
// (low-res) tiles.  Per tile, two 64-channel halves of x and the four tap images of dy
// (64 pixels x 64 channels each, 128-byte rows) are brought in by LDS-DMA into a three-stage
// ring (two tiles in flight: the op is latency-bound with one).  Wave w = (x half, tap) multiplies a 64 x 64 block: transposing LDS reads
// (ds_read_b64_tr_b16) feed v_mfma_f32_32x32x16_bf16, reduction step = one 16-pixel tile row.
// Workgroups write fp32 slabs in the parameter's own [ci][co][2][2] order; the slabs are
// summed in a fixed order by fea_slab_reduce_kernel (common.h; deterministic).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

struct WgradUpArgs {
    const __bf16* x; const __bf16* dy; float* out;        // out: slab [split][Cin][Cout][4] (dW itself when splits == 1)
    long long out_split_stride;
    int ldx, lddy, H, W, Cin, Cout;                        // H, W: low-res (x) dims
    int tiles_x, tiles_y, ntiles, tiles_per_split, nmt, nnt;
};

__device__ __forceinline__ int swzu(int x) {              // as wgrad2.hip: bit-reverse of bits 1..3
    const int u = (x >> 1) & 7;
    return ((u & 1) << 2) | (u & 2) | (u >> 2);
}

constexpr int WU_REGION = 64 * 128;                        // 64 pixels x 128 B
// a stage: XQ 64-channel slices of x, then the dy taps 0..3

template <int XQ>
__device__ __forceinline__ void wu_issue(const WgradUpArgs& g, int m0, int n0, int wave, int lane, int tile, char* sb) {
    int tt = tile;
    const int tile_x = tt % g.tiles_x; tt /= g.tiles_x;
    const int tile_y = tt % g.tiles_y;
    const int n_img = tt / g.tiles_y;
    const int p = 8 * wave + (lane >> 3);                  // pixel of the 4 x 16 tile this lane moves
    const int gy = tile_y * 4 + (p >> 4), gx = tile_x * 16 + (p & 15);
    const bool ok = gy < g.H && gx < g.W;
    const int chunk = ((lane & 7) ^ swzu(p & 15)) * 8;
    const __bf16* zero = reinterpret_cast<const __bf16*>(fea_zero_line);
    const __bf16* xs = g.x + ((long long)(n_img * g.H + gy) * g.W + gx) * g.ldx + m0 + chunk;
    const __bf16* ds = g.dy + ((long long)(n_img * 2 * g.H + 2 * gy) * (2 * g.W) + 2 * gx) * g.lddy + n0 + chunk;
#pragma unroll
    for (int r = 0; r < XQ + 4; ++r) {
        const __bf16* src = r < XQ ? xs + r * 64 : ds + ((long long)((r - XQ) >> 1) * (2 * g.W) + ((r - XQ) & 1)) * g.lddy;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(ok ? src : zero), (lds_void_t*)(sb + r * WU_REGION + wave * 1024), 16, 0, 0);
    }
}

// __restrict__ on purpose (see wgrad2.hip): lets hipcc keep the LDS-DMA of the next tile in
// flight across the LDS reads of this one.
template <int XQ>
__device__ __forceinline__ void wu_step(const WgradUpArgs& g, int m0, int n0, int wave, int lane, int next_tile,
                                        char* __restrict__ dma, const char* __restrict__ rd,
                                        const int (&col)[2][2], f32x16 (&acc)[XQ / 2][2][2]) {
    if (next_tile >= 0) wu_issue<XQ>(g, m0, n0, wave, lane, next_tile, dma);
    const char* P = rd + (wave >> 2) * (XQ / 2) * WU_REGION;
    const char* Q = rd + (XQ + (wave & 3)) * WU_REGION;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 af[XQ / 2][2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < XQ / 2; ++q) {
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(P + q * WU_REGION + ks * 2048 + col[i][0]));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(P + q * WU_REGION + ks * 2048 + col[i][1]));
                af[q][i] = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Q + ks * 2048 + col[i][0]));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Q + ks * 2048 + col[i][1]));
            bf[i] = __builtin_bit_cast(u32x4, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int q = 0; q < XQ / 2; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma16(acc[q][i][j], af[q][i], bf[j], (const __bf16*)nullptr);
    }
}

template <int XQ>
__global__ __launch_bounds__(512, 2) void wgrad_up_kernel(const WgradUpArgs g) {
    constexpr int WU_STAGE = (XQ + 4) * WU_REGION;
    constexpr int NS = XQ == 2 ? 3 : 2;            // ring stages: NS - 1 tiles in flight (48 KB stages: three; 64 KB: two)
    constexpr int NP = XQ + 4;                     // LDS-DMA pieces per wave and tile
    __shared__ __attribute__((aligned(128))) char smem[NS * WU_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntp = g.nmt * g.nnt;
    const int tp = lid % ntp, split = lid / ntp;
    const int m0 = (tp / g.nnt) * (64 * XQ), n0 = (tp % g.nnt) * 64;

    // fragment byte offsets inside a 16-pixel row block (identical to wgrad2.hip, TW = 16):
    // 32-channel group i, k half t: pixel x = 8h + 4t + (li >> 2), channels 16(gi&1) + 4(li&3) ..
    int col[2][2];
    {
        const int gi = lane >> 4, li = lane & 15;
        const int chl = 16 * (gi & 1) + 4 * (li & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int x = 8 * h + 4 * t + (li >> 2);
                const int ch = i * 32 + chl;
                col[i][t] = x * 128 + (((ch >> 3) ^ swzu(x)) << 4) + (ch & 7) * 2;
            }
    }
    f32x16 acc[XQ / 2][2][2];
#pragma unroll
    for (int q = 0; q < XQ / 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][i][j][e] = 0.f;

    const int t_begin = split * g.tiles_per_split;
    const int t_end = min(t_begin + g.tiles_per_split, g.ntiles);
    if (t_begin < t_end) wu_issue<XQ>(g, m0, n0, wave, lane, t_begin, smem);
    if (NS == 3 && t_begin + 1 < t_end) wu_issue<XQ>(g, m0, n0, wave, lane, t_begin + 1, smem + WU_STAGE);
    int stage = 0;
    for (int tile = t_begin; tile < t_end; ++tile) {
        // LDS-DMA loads retire in order: all but the pieces of the tiles behind this one have landed
        if (NS == 3 && tile + 1 < t_end) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");
        else                             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // tile landed; the stage read in the previous iteration is free
        const int dma = stage >= 1 ? stage - 1 : NS - 1;       // (stage + NS - 1) % NS
        wu_step<XQ>(g, m0, n0, wave, lane, tile + NS - 1 < t_end ? tile + NS - 1 : -1, smem + dma * WU_STAGE, smem + stage * WU_STAGE, col, acc);
        stage = stage == NS - 1 ? 0 : stage + 1;
    }
    // out[(m0 + mh*64 + row)][n0 + col][tap]
    const int mh = wave >> 2, tap = wave & 3;
    float* dst = g.out + (long long)split * g.out_split_stride;
#pragma unroll
    for (int q = 0; q < XQ / 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + (mh * (XQ / 2) + q) * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    dst[((long long)row * g.Cout + n0 + j * 32 + r) * 4 + tap] = acc[q][i][j][e];
                }
}


struct WuPlan { int tiles_x, tiles_y, ntiles, splits, tiles_per_split, nmt, nnt, xq; };
static WuPlan wu_plan(int N, int H, int W, int Cin, int Cout) {
    WuPlan p;
    p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 3) / 4;
    p.ntiles = N * p.tiles_x * p.tiles_y;
    const char* e = getenv("FEA_AMD_WU256");                 // A/B switch: 0 = 128-channel workgroups everywhere
    p.xq = (Cin % 256 == 0 && !(e && e[0] == '0')) ? 4 : 2;
    p.nmt = Cin / (64 * p.xq); p.nnt = Cout / 64;
    int want = 256 / (p.nmt * p.nnt);
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tiles_per_split = (p.ntiles + want - 1) / want;
    p.splits = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
    return p;
}

bool fea_wgrad_up_supported(int Cin, int Cout, int dtype) { return dtype == FEA_BF16 && Cin % 128 == 0 && Cout % 64 == 0; }

size_t fea_wgrad_up_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    WuPlan p = wu_plan(N, H, W, Cin, Cout);
    return p.splits > 1 ? (size_t)p.splits * 4 * Cin * Cout * sizeof(float) : 0;
}

// arguments already validated by the caller (wgrad.hip)
int fea_wgrad_up_launch(const void* x, int ldx, const void* dy, int lddy, float* dW, void* ws, size_t ws_bytes,
                        int N, int H, int W, int Cin, int Cout, hipStream_t s) {
    WuPlan p = wu_plan(N, H, W, Cin, Cout);
    const size_t need = p.splits > 1 ? (size_t)p.splits * 4 * Cin * Cout * sizeof(float) : 0;
    FEA_CHECK(ws_bytes >= need, FEA_E_WORKSPACE, "fea_upconv2x2_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    FEA_CHECK(((uintptr_t)dW % 16 == 0) && ((uintptr_t)ws % 16 == 0), FEA_E_BADARG, "fea_upconv2x2_wgrad: dW/ws must be 16-byte aligned");
    FEA_CHECK((long long)N * 4 * H * W < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "fea_upconv2x2_wgrad: too many pixels for 32-bit pixel indices");
    WgradUpArgs a{};
    a.x = (const __bf16*)x; a.dy = (const __bf16*)dy;
    a.out = p.splits > 1 ? (float*)ws : dW;
    a.out_split_stride = (long long)4 * Cin * Cout;
    a.ldx = ldx; a.lddy = lddy; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split;
    a.nmt = p.nmt; a.nnt = p.nnt;
    if (p.xq == 4) hipLaunchKernelGGL(wgrad_up_kernel<4>, dim3((unsigned)(p.nmt * p.nnt * p.splits)), dim3(512), 0, s, a);
    else           hipLaunchKernelGGL(wgrad_up_kernel<2>, dim3((unsigned)(p.nmt * p.nnt * p.splits)), dim3(512), 0, s, a);
    FEA_LAUNCH_CHECK();
    if (p.splits > 1) {
        const long long per4 = (long long)Cin * Cout;
        fea_launch_slab_reduce((const float*)ws, dW, p.splits, per4, s);
        FEA_LAUNCH_CHECK();
    }
    return FEA_OK;
}
