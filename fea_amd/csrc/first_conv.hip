// First layer, persistent form: nn.Conv2d(in_channels, 64, 3, padding=1) (enc1[0], models.py:10,28) over the fp32 NCHW input, bf16 out.
//
// The layer is one output store (2 bytes per pixel and channel against 4 C / 64 of input): igemm.hip's MODE 3 runs it as one
// 256-pixel tile per workgroup -- halo load, LDS, im2col gather, 16 MFMAs, staged store, every latency in a row, four workgroups per
// CU to hide them: 0.74 ms at batch 64 x 512^2 on random data.  Here a workgroup walks a range of 8x32-pixel tiles: 0.62 ms.
// That is 3.4 TB/s of stores where a store-only kernel of the same tile shape and lane pattern reaches 4.5-5.7
// (scripts/hbm_tile_probe.hip): with three of every four stores removed the kernel takes 0.31 ms, with the gather removed
// 0.62 -- the gather / MFMA part and the store stream add up instead of overlapping (open).
//   * the NEXT tile's fp32 halo (C x 10 x 34) travels global -> registers while this tile is multiplied, and goes to the other
//     of two LDS halo buffers behind one barrier per tile;
//   * a wave owns 64 pixels x 64 channels, two 32-pixel halves in turn;
//     im2col fragments are gathered from the LDS halo (k = tap * C + c), the packed weights (fea_pack_conv3x3_first) live in
//     registers as the MFMA's A operand and the bias is the C operand: an accumulator register holds 4 consecutive channels of
//     one pixel, one v_permlane16_swap per dword gives a lane 8 consecutive channels = one 16-byte store (conv_p.hip);
//   * BatchNorm partial sums (sum y, sum y^2) are accumulated in registers over all tiles of the workgroup and written once per
//     wave: fea_conv3x3_first_rows() rows instead of one per tile.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

struct FcArgs {
    const float* x; const __bf16* w; const float* bias; __bf16* y; float* stats;
    int ldy, C, H, W;
    int tiles_x, tiles_y, ntiles, tiles_per_wg;
    int walk, wsx, wsn;                 // walk = 1: workgroup b takes tiles b, b + gridDim.x, ... (the cursor keeps its tile row: stride (wsx, wsn) in (x, image))
    unsigned long long* dbg;            // FC_STAMPS builds only: per-wave phase cycle sums (scripts/fc_stamps.py)
};

constexpr int FC_TH = 8, FC_TW = 32, FC_HW2 = FC_TW + 2, FC_HALO = (FC_TH + 2) * FC_HW2;       // 340 halo pixels per channel
#ifdef FC_EXP_NO_GATHER      // EXPERIMENT (wrong results): one LDS word per fragment instead of eight
#define FC_KOFF(ks, j) koff[ks][0]
#else
#define FC_KOFF(ks, j) koff[ks][j]
#endif
#ifndef FC_AHEAD
#define FC_AHEAD 1        // tiles whose halo is in flight (3, at two workgroups per CU for the registers: 0.693 ms against 0.662)
#endif
constexpr int FC_WGS_PER_CU = 3;                  // (with the 32 registers of BatchNorm sums the kernel holds ~150: three waves per SIMD)

__device__ __forceinline__ float fc_dpp_row(float v, int step) {      // partner of a 16-lane-row butterfly step
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (step) {
        case 0: r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false); break;      // lane ^ 1
        case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false); break;      // lane ^ 2
        case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false); break;     // half-row mirror
        default: r = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false); break;    // row mirror
    }
    return __builtin_bit_cast(float, r);
}

// KS: k-steps of 32 (1: 9 C <= 32, 2: 9 C <= 64); STATS: BatchNorm partial sums; RELU: inference form relu(conv + bias)
template <int KS, bool STATS, bool RELU>
__global__ __launch_bounds__(256, KS == 1 ? 3 : 2) void first_conv_kernel(const FcArgs g) {
    constexpr int HIT = KS == 1 ? (3 * FC_HALO + 255) / 256 : (7 * FC_HALO + 255) / 256;       // halo elements per thread (C <= 3 / <= 7)
    constexpr int HBUF = (KS == 1 ? 3 : 7) * FC_HALO;
    __shared__ float hal[2][HBUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q16 = lane >> 4;
    const int C = g.C, H = g.H, W = g.W;
    // tile walk: consecutive tiles, or (walk) every gridDim.x-th tile -- the chip then writes one compact window of gridDim.x tiles
    // at any moment, which the store stream likes better (conv_p.hip, scripts/hbm_tile_probe.hip); t counts this workgroup's tiles
    const int first = g.walk ? (int)blockIdx.x : (int)blockIdx.x * g.tiles_per_wg;
    const int t0 = 0;
    const int t1 = first >= g.ntiles ? 0 : g.walk ? (g.ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x : min(g.tiles_per_wg, g.ntiles - first);
    if (t0 >= t1) return;
    auto advance = [&](int& n_, int& x_, int& y_) {
        if (g.walk) { x_ += g.wsx; if (x_ >= g.tiles_x) { x_ -= g.tiles_x; ++n_; } n_ += g.wsn; }
        else if (++y_ == g.tiles_y) { y_ = 0; if (++x_ == g.tiles_x) { x_ = 0; ++n_; } }
    };

    // packed weights [ks][co/16][lane][8]: the fragment of (k-step, 16-channel tile) is this lane's 16 bytes
    u32x4 Wf[KS][4];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int n = 0; n < 4; ++n) Wf[ks][n] = *reinterpret_cast<const u32x4*>(g.w + ((long long)(ks * 4 + n) * 64 + lane) * 8);
    // bias (channels n*16 + 4 q16 + e of a lane) is the C operand of a tile's first MFMAs: kept in LDS and re-read per tile half
    // (16 registers less for the kernel's life: three waves per SIMD without spills)
    __shared__ __attribute__((aligned(16))) float bsh[64];
    if (tid < 64) bsh[tid] = g.bias ? g.bias[tid] : 0.f;
    // im2col gather offsets of this lane's 8 k values per k-step: k = ks*32 + q16*8 + j = tap*C + c -> c*HALO + dy*34 + dx
    int koff[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 32 + q16 * 8 + j;
            const int tap = C == 3 ? k / 3 : k / C, c = k - tap * C;
            const int dy = tap / 3, dx = tap - dy * 3;
            koff[ks][j] = k < 9 * C ? c * FC_HALO + dy * FC_HW2 + dx : 0;      // k >= 9 C: the packed weight is zero, any finite operand will do
        }
    // halo elements this thread moves: e = tid + 256 i -> (c, hy, hx)
    constexpr int NONE = 0x7fffffff;
    int hoff[HIT];                  // offset from the tile's first pixel (negative on the top / left halo lines), or NONE (beyond the C channels)
    unsigned long long hedge = 0;   // bit 4 i + {0,1,2,3}: element i lies on the top / bottom / left / right halo line
#pragma unroll
    for (int i = 0; i < HIT; ++i) {
        const int e = tid + i * 256;
        const int c = e / FC_HALO, rr = e - c * FC_HALO;
        const int hy = rr / FC_HW2, hx = rr - hy * FC_HW2;
        hoff[i] = e < C * FC_HALO ? (c * H + hy - 1) * W + hx - 1 : NONE;
        hedge |= (unsigned long long)((hy == 0 ? 1u : 0u) | (hy == FC_TH + 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == FC_TW + 1 ? 8u : 0u)) << (4 * i);
    }
    float hv[FC_AHEAD][HIT];
    int ty = first % g.tiles_y, tx = (first / g.tiles_y) % g.tiles_x, ni = first / (g.tiles_y * g.tiles_x);
    int fy = ty, fx = tx, fn = ni;          // the fetch cursor runs FC_AHEAD tiles in front
    auto fetch = [&](auto slotc) {
        constexpr int slot = decltype(slotc)::value;
        const float* __restrict__ xin = g.x + (long long)fn * C * H * W + (fy * FC_TH) * W + fx * FC_TW;
        const unsigned edge = (fy == 0 ? 1u : 0u) | (fy == g.tiles_y - 1 ? 2u : 0u) | (fx == 0 ? 4u : 0u) | (fx == g.tiles_x - 1 ? 8u : 0u);
#pragma unroll
        for (int i = 0; i < HIT; ++i) {
            const bool ok = hoff[i] != NONE && (((unsigned)(hedge >> (4 * i)) & edge) == 0);       // zero padding outside the image
            hv[slot][i] = ok ? xin[hoff[i]] : 0.f;
        }
        advance(fn, fx, fy);
    };
    sfor<0, FC_AHEAD>([&](auto dc) { if (t0 + decltype(dc)::value < t1) fetch(dc); });

    f32x4 S[4], Q[4];                // sum y, sum y^2 of channels n*16 + 4 q16 + e over this wave's pixels of every tile
#pragma unroll
    for (int n = 0; n < 4; ++n) { S[n] = f32x4{0.f, 0.f, 0.f, 0.f}; Q[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int chl = (q16 & 1) * 16 + (q16 >> 1) * 8;

#ifdef FC_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define FC_STAMP(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define FC_STAMP(k) do {} while (0)
#endif
    for (int tb = t0; tb < t1; tb += FC_AHEAD)
    sfor<0, FC_AHEAD>([&](auto dc) {
        constexpr int slot = decltype(dc)::value;
        const int t = tb + slot;
        if (t >= t1) return;
        float* const hb = hal[(t - t0) & 1];
#pragma unroll
        for (int i = 0; i < HIT; ++i)
            if (tid + i * 256 < HBUF) hb[tid + i * 256] = hv[slot][i];
        FC_STAMP(0);                 // wait for the prefetched halo + LDS write
        __syncthreads();
        FC_STAMP(1);                 // barrier             // this tile's halo is complete; every wave has left the tile before the previous one (the other buffer)
        if (t + FC_AHEAD < t1) fetch(dc);          // in flight behind the next FC_AHEAD tiles' gathers, MFMAs and stores
        FC_STAMP(2);                 // issue of the next halo
        __bf16* const yt = g.y + (((long long)ni * H + ty * FC_TH) * W + tx * FC_TW) * g.ldy;
#pragma unroll 1
        for (int mh = 0; mh < 2; ++mh) {          // the wave's tile rows 2 wave + mh: two 16-pixel tiles each (rolled: registers)
            f32x4 acc[2][4], bv[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bv[n] = *reinterpret_cast<const f32x4*>(bsh + n * 16 + 4 * q16);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int pb = (2 * wave + mh) * FC_HW2 + m * 16 + r16;       // halo position of tap (0,0) of this lane's pixel
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 af;
#pragma unroll
                    for (int j = 0; j < 8; ++j) af[j] = (__bf16)hb[pb + FC_KOFF(ks, j)];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[ks][n]), af, ks == 0 ? bv[n] : acc[m][n], 0, 0, 0);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if constexpr (RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[m][n][e] = fmaxf(acc[m][n][e], 0.f);
                    }
                    if constexpr (STATS) { S[n] += acc[m][n]; Q[n] = __builtin_elementwise_fma(acc[m][n], acc[m][n], Q[n]); }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x2 a01 = {acc[m][2 * j][0], acc[m][2 * j][1]}, a23 = {acc[m][2 * j][2], acc[m][2 * j][3]};
                    const f32x2 b01 = {acc[m][2 * j + 1][0], acc[m][2 * j + 1][1]}, b23 = {acc[m][2 * j + 1][2], acc[m][2 * j + 1][3]};
                    const uint32_t lo0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(a01, bf16x2));
                    const uint32_t hi0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(a23, bf16x2));
                    const uint32_t lo1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(b01, bf16x2));
                    const uint32_t hi1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(b23, bf16x2));
                    // odd 16-lane rows of the first operand <-> even rows of the second: every lane ends up with 8 consecutive channels
                    const auto r0 = __builtin_amdgcn_permlane16_swap(lo0, lo1, false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(hi0, hi1, false, false);
                    const u32x4 out = {r0[0], r1[0], r0[1], r1[1]};
#ifdef FC_EXP_NO_STORE       // EXPERIMENT (wrong results): one store per tile half instead of four
                    if (m == 0 && j == 0)
#endif
                    *reinterpret_cast<u32x4*>(yt + ((long long)(2 * wave + mh) * W + m * 16 + r16) * g.ldy + j * 32 + chl) = out;
                }
            }
        }
        FC_STAMP(3);                 // gathers, MFMAs, epilogue and stores of both halves
        advance(ni, tx, ty);
    });
#ifdef FC_STAMPS
    if (lane == 0 && g.dbg) for (int k = 0; k < 6; ++k) g.dbg[((long long)blockIdx.x * 4 + wave) * 6 + k] = ph[k];
#endif

    if constexpr (STATS) {
        // one row per wave: sums over the 16 pixel lanes of each 16-lane row, written by its first lane
        const long long row = (long long)blockIdx.x * 4 + wave;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = S[n][e], q = Q[n][e];
#pragma unroll
                for (int st = 0; st < 4; ++st) { s += fc_dpp_row(s, st); q += fc_dpp_row(q, st); }
                if (r16 == 0) {
                    float* dst = g.stats + (row * 64 + n * 16 + 4 * q16 + e) * 2;
                    dst[0] = s; dst[1] = q;
                }
            }
    }
}

// ---------------------------------------------------------------------------
struct FcPlan { int tiles_x, tiles_y, ntiles, nwg, tiles_per_wg, walk; };

static bool fc_plan(int N, int C, int H, int W, int Cout, FcPlan& p) {
    const char* e = getenv("FEA_AMD_FIRST_P");              // A/B switch: 0 = the one-tile-per-workgroup kernel of igemm.hip
    if (e && e[0] == '0') return false;
    if (Cout != 64 || C <= 0 || 9 * C > 64 || N <= 0 || H % FC_TH || W % FC_TW) return false;
    if ((long long)C * H * W >= (1ll << 31) || (long long)N * (H / FC_TH) * (W / FC_TW) >= (1ll << 30)) return false;
    p.tiles_x = W / FC_TW; p.tiles_y = H / FC_TH;
    p.ntiles = N * p.tiles_x * p.tiles_y;
    int want = 256 * FC_WGS_PER_CU * fea_wg_split();
    if (want > p.ntiles) want = p.ntiles;
    p.tiles_per_wg = (p.ntiles + want - 1) / want;
    p.nwg = (p.ntiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    // interleaved walk when the stride (= the grid) is a whole number of tile columns; FEA_AMD_FIRST_WALK=0: consecutive (A/B runs)
    static const bool nowalk = getenv("FEA_AMD_FIRST_WALK") && getenv("FEA_AMD_FIRST_WALK")[0] == '0';
    p.walk = !nowalk && want % p.tiles_y == 0 && p.tiles_per_wg > 1;
    if (p.walk) p.nwg = want;
    return true;
}

// rows of the BatchNorm partial table fea_conv3x3_first writes; 0 = one row per 256-pixel tile (fea_conv_num_tiles)
extern "C" int fea_conv3x3_first_rows(int N, int C, int H, int W, int Cout, int dtype) {
    FcPlan p;
    if (dtype != FEA_BF16 || !fc_plan(N, C, H, W, Cout, p)) return 0;
    return p.nwg * 4;
}

// false = shape not served (the caller runs igemm.hip's kernel); arguments already validated by the caller
bool fea_first_conv_p(const float* x, const void* w, const float* bias, void* y, int ldy, float* stats, int relu,
                      int N, int C, int H, int W, int Cout, hipStream_t s, int* rc) {
    FcPlan p;
    if (!fc_plan(N, C, H, W, Cout, p)) return false;
    FcArgs a{};
    a.x = x; a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y; a.stats = stats;
    a.ldy = ldy; a.C = C; a.H = H; a.W = W;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.tiles_per_wg = p.tiles_per_wg;
    a.walk = p.walk; a.wsx = (p.nwg / p.tiles_y) % p.tiles_x; a.wsn = p.nwg / (p.tiles_y * p.tiles_x);
#ifdef FC_STAMPS
    a.dbg = getenv("FC_DBG_PTR") ? (unsigned long long*)strtoull(getenv("FC_DBG_PTR"), nullptr, 0) : nullptr;
#endif
    dim3 grid((unsigned)p.nwg), block(256);
    const bool k1 = 9 * C <= 32;
#define FC_LAUNCH(KSV, STV, RV) hipLaunchKernelGGL((first_conv_kernel<KSV, STV, RV>), grid, block, 0, s, a)
    if (relu)       { if (k1) FC_LAUNCH(1, false, true); else FC_LAUNCH(2, false, true); }
    else if (stats) { if (k1) FC_LAUNCH(1, true, false); else FC_LAUNCH(2, true, false); }
    else            { if (k1) FC_LAUNCH(1, false, false); else FC_LAUNCH(2, false, false); }
#undef FC_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fea_set_error("fea_conv3x3_first (persistent): launch failed: %s", hipGetErrorString(e)); *rc = FEA_E_LAUNCH; }
    else *rc = FEA_OK;
    return true;
}
