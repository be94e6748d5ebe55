// Implicit-GEMM convolution kernels (MFMA, gfx950).
//
// One workgroup (256 threads = 4 waves) computes a tile of 256 output pixels
// (8x32 or 16x16 of one image) x BN output channels.  GEMM view: M = pixels,
// N = output channels, K = taps x input channels.  A rows (pixels) and B rows
// (output channels of the packed weight) are staged in LDS as 128-byte channel
// rows with a 16-byte-chunk XOR swizzle, and every lane reads one 16-byte chunk
// of an A row and of a B row per k-step (common.h: mma16).
//
//   MODE 0  3x3, pad 1: the (TH+2)x(TW+2) halo tile of one channel chunk is
//           staged ONCE and reused by the nine taps (row index shifted per tap);
//           the weight tile of each tap is double-buffered.
//   MODE 1  1x1: A rows = the tile's pixels.
//   MODE 2  2x2 stride-2 gather (ConvTranspose2d data gradient): A row of pixel
//           (i,j), tap (a,b) = input pixel (2i+a, 2j+b).
//   MODE 3  first layer, bf16: 3x3 pad 1 over an fp32 NCHW input with few channels
//           (K = 9*C <= 64).  The fp32 halo tile is staged in LDS and the im2col A
//           fragments are gathered from it on the fly (no im2col tensor in HBM);
//           the weight fragments live in registers.
//   EPI 0   y[pixel][cout0+col] (+bias, + BatchNorm partial sums)
//   EPI 1   pixel-shuffle store of ConvTranspose2d forward: GEMM column
//           n = ab*Cup + co goes to pixel (2i+a, 2j+b), channel co.
//   EPI 3   inference with BatchNorm folded into the weights: relu(acc + bias), stored as the
//           activated tensor (channel slice `ldy`), plus its 2x2 max-pool when `pool` is set
//   EPI 2   EPI 0 + the first pass of the NEXT BatchNorm backward fused into the
//           store loop: the tile just computed is an activation gradient da; with the
//           layer's pre-BN output y, (sum g, sum g*xhat) per channel are written per
//           pixel tile (g = da where relu(bn(y)) > 0), replacing fea_bn_relu_bwd_reduce.
//
// Reference layers served: nn.Conv2d(ci,co,3,padding=1) models.py:28,31 (fwd and
// dgrad), nn.ConvTranspose2d(ci,co,2,stride=2) models.py:15-21 (fwd, bwd-data).
#include "common.h"
#include <stdlib.h>

#ifndef EPI2_GRP
#define EPI2_GRP 4          // y loads in flight per thread in the fused BatchNorm-backward epilogue
#endif

struct IgemmArgs {
    const void* x; const void* w; const float* bias; void* y; float* stats;
    long long x_img_stride, y_img_stride;   // elements per image
    int ldx, ldy;
    int H, W;          // tile-space dims
    int Cin, Cout;     // per-tap GEMM K channels (multiple of CK), GEMM N (multiple of BN)
    int Cup;           // EPI 1: channels per (a,b) quadrant
    int tw_shift, tiles_x, tiles_y;
    int nct;           // Cout / BN
    // EPI 2: the output is the activation gradient da of a conv->BN->ReLU layer; reduce
    // (sum g, sum g*xhat), g = da * (relu(bn(y)) > 0), per channel over the tile
    const void* ybn; int ldybn;
    const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
    float* red_part;   // [pixel tile][Cout][2]
    // EPI 3: optional 2x2 max-pool of the activated output, [N][H/2][W/2] pixels, ldpool apart
    void* pool; int ldpool;
    unsigned long long* dbg;     // IG_STAMPS builds only: per-wave phase cycle sums [workgroup][wave][6]
};
#ifdef IG_STAMPS
#define ISTAMP(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); iph[k] += tn_ - itprev; itprev = tn_; } while (0)
#else
#define ISTAMP(k) do {} while (0)
#endif

// XOR applied to a row's 16-byte chunk index.  A ds_read_b128 is served in groups of 16 lanes = 16 consecutive rows, 8 of them
// reading chunk q and 8 chunk q ^ 1 (MI355X_MICROARCH.md, LDS table): bit 0 of the chunk is left alone so the two subsets never
// meet, and any 8 consecutive rows get 8 distinct (row parity, XOR) pairs -- conflict-free for every base row, where
// (row >> 1) & 7 was only for base rows that are multiples of 4 (PMC: 30 % of the LDS cycles were conflict cycles).
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 3) << 1; }

// LDS read schedule of one tap of the bf16 path (see compute()): reads are issued
// in a fixed order, the LDS returns them in order, so "read number i has
// landed" == "at most issued - (i+1) reads outstanding".
template <int MT16>
struct TapSched {
    static constexpr int S = 2 * MT16;
    // k-step-1 B fragments read after slot i
    static constexpr int nb(int i) { return MT16 == 8 ? ((i >= 2 && i <= 5) ? 1 : 0) : (i <= 1 ? 2 : 0); }
    static constexpr int bfirst(int i) { int c = 0; for (int u = 0; u < i; ++u) c += nb(u); return c; }
    static constexpr int na(int i) { return i + 3 < S ? 1 : 0; }
    static constexpr int issued_before(int i) { int c = 7; for (int u = 0; u < i; ++u) c += nb(u) + na(u); return c; }
    static constexpr int idxA(int s) { return s < 3 ? 4 + s : issued_before(s - 3) + nb(s - 3); }
    static constexpr int idxBlast() {
        int last = 0;
        for (int i = 0; i < S; ++i) if (nb(i) > 0) last = issued_before(i) + nb(i) - 1;
        return last;
    }
    static constexpr int need(int s) { int n = idxA(s); if (s == MT16 && idxBlast() > n) n = idxBlast(); return n; }
    static constexpr int N(int s) { return issued_before(s) - (need(s) + 1); }
};

#ifndef IG_GATHER_REGS
#define IG_GATHER_REGS 1
#endif
#ifndef IG_FAST_ADDR
#define IG_FAST_ADDR 1       // precomputed LDS addresses + unrolled taps in the 128-column 3x3 kernel (0: the round-2 loop, for A/B builds)
#endif
// One 16-byte-per-lane LDS-DMA piece as a buffer load: descriptor over [base, base + nbytes), per-lane byte offset `voff` (out of
// range: the lane receives zeros), wave-uniform byte offset `soff`.  A free function on purpose: with the descriptor made inside a
// lambda of the kernel template, hipcc 7.2 drops the kernel's host stub without a diagnostic.
__device__ __forceinline__ void ig_buffer_lds16(const void* base, int nbytes, char* lds, int voff, int soff) {
    typedef __attribute__((address_space(3))) void lds_v_t;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_v_t*)lds, 16, voff, soff, 0, 0);
}

template <typename T, int MODE, int BN, int EPI, int TWS>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs g) {
    constexpr int CK = Elem<T>::CK, EPC = Elem<T>::EPC;
    constexpr int MT = BN == 128 ? 4 : 2;        // 32-row M tiles per wave
    constexpr int WM = BN == 128 ? 2 : 4;        // waves along M
    constexpr int A_ROWS = MODE == 0 ? 352 : 256;
    constexpr int A_PASSES = A_ROWS / 32;        // 256 threads x 16 B = 32 rows per pass
    constexpr int A_BYTES = A_ROWS * 128;
    // bf16 3x3: the weight (B) fragments go global -> VGPR directly, one tap ahead
    // (fragment-ordered pack, 1 KB contiguous per wave-instruction): no weight tile
    // in LDS and NO per-tap barrier; the A halo tile is the only LDS resident.
    constexpr bool DB = MODE == 0 && sizeof(T) == 2 && (BN == 64 || IG_DB128);   // (BN = 128: doubling the L2 weight traffic costs more than the barrier)
    constexpr bool GR = MODE == 2 && IG_GATHER_REGS;   // (MODE 1 measured slower with it: 128->64 @256^2 0.80 -> 0.96 ms)
    constexpr int NWBUF = DB ? 0 : ((MODE == 0 || GR) ? 2 : 1);
    constexpr int W_BYTES = BN * 128;
    constexpr int W_PASSES = BN / 32;
    constexpr int EPI_STRIDE = BN * 2 + 16;      // bf16 staging row, padded
    constexpr int EPI_BYTES = sizeof(T) == 2 ? 256 * EPI_STRIDE : 0;
    constexpr int STAT_BYTES = 4 * BN * 2 * 4;     // per-wave partial sums [4][BN][2]
    constexpr int MAIN_BYTES = A_BYTES + NWBUF * W_BYTES;
    constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES + STAT_BYTES ? MAIN_BYTES : EPI_BYTES + STAT_BYTES;
    __shared__ __attribute__((aligned(128))) char smem[LDS_BYTES];
    char* const ldsA = smem;
    char* const ldsW = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = BN == 128 ? (wave >> 1) : wave;
    const int wn = BN == 128 ? (wave & 1) : 0;
    const int r = lane & 31, h = lane >> 5;          // fp32: 32x32x2 MFMA lane map
    const int r16 = lane & 15, q16 = lane >> 4;      // bf16: 16x16x32 MFMA lane map
    constexpr bool BF = sizeof(T) == 2;
    constexpr int MT16 = MT * 2;                     // bf16: 16-row M tiles per wave (x 4 16-col N tiles)

    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = lid % g.nct;
    int pt = lid / g.nct;                         // pixel tile id (also the stats slot)
    const int ptile = pt;
    const int tile_x = pt % g.tiles_x; pt /= g.tiles_x;
    const int tile_y = pt % g.tiles_y;
    const int n_img = pt / g.tiles_y;
    const int cout0 = ct * BN;
    constexpr int tws = TWS, TW = 1 << tws, TH = 256 >> tws;
    const int y0 = tile_y * TH, x0 = tile_x * TW;
    const int H = g.H, W = g.W;
    const int nchunks = g.Cin / CK;

    const T* __restrict__ xg = (const T*)g.x + (long long)n_img * g.x_img_stride;
    const T* __restrict__ wg = (const T*)g.w;

    // ---- per-thread staging descriptors for the A tile --------------------
    // MODE 0 stages once per nine taps, so its offsets are recomputed at each
    // staging (saves 11 live registers); the gather modes stage every k-chunk
    // and keep theirs.
    // Staging is LDS-DMA (global_load_lds, 16 B per lane): wave-instruction j = 4 i + wave
    // fills LDS rows 8j .. 8j+7 linearly, lane -> (row 8j + lane/8, 16-byte slot lane%8).
    // The XOR swizzle is applied on the SOURCE: slot s of a row holds global chunk s ^ swz(row).
    const int sc = tid & 7;         // LDS slot of the 128-byte row this lane fills
    auto a_desc = [&](int i, bool& ok) -> int {
        const int row = i * 32 + (tid >> 3);
        const int sc = (tid & 7) ^ swz(row);      // source chunk
        int gy, gx;
        if (MODE == 0) {
            const int HW2 = TW + 2;
            const int hy = (tws == 5) ? row / 34 : row / 18;
            const int hx = row - hy * HW2;
            gy = y0 + hy - 1; gx = x0 + hx - 1;
            ok = (hy < TH + 2) && gy >= 0 && gy < H && gx >= 0 && gx < W;
            return ok ? (gy * W + gx) * g.ldx + sc * EPC : 0;
        }
        gy = y0 + (row >> tws); gx = x0 + (row & (TW - 1));
        ok = gy < H && gx < W;
        if (MODE == 1) return ok ? (gy * W + gx) * g.ldx + sc * EPC : 0;
        return ok ? ((2 * gy) * (2 * W) + 2 * gx) * g.ldx + sc * EPC : 0;
    };
    constexpr int NDESC = MODE == 0 ? 1 : A_PASSES;
    int a_off[NDESC];
    unsigned a_valid = 0;
    if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int i = 0; i < NDESC; ++i) {
            bool ok; a_off[i] = a_desc(i, ok);
            a_valid |= (ok ? 1u : 0u) << i;
        }
    }

    // ---- per-lane fragment row descriptors ---------------------------------
    // fp32 (32x32x2): a_row[MT] / b rows of 32;  bf16 (16x16x32): a_row16[MT16] / 4 b rows of 16.
    // A rows: MODE 0: halo row of tap (0,0); else tile row.
    int a_row[MT], b_addr[2], b_swz[2];
    int a_row16[MT16], b_row16[4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = wm * (MT * 32) + m * 32 + r;
        a_row[m] = MODE == 0 ? (p >> tws) * (TW + 2) + (p & (TW - 1)) : p;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int brow = wn * 64 + nt * 32 + r;
        b_addr[nt] = brow * 128;
        b_swz[nt] = swz(brow);
    }
#pragma unroll
    for (int m = 0; m < MT16; ++m) {
        const int p = wm * (MT * 32) + m * 16 + r16;
        a_row16[m] = MODE == 0 ? (p >> tws) * (TW + 2) + (p & (TW - 1)) : p;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) b_row16[n] = wn * 64 + n * 16 + r16;

    f32x16 acc[MT][2];              // fp32 path
    f32x4 acc16[MT16][4];           // bf16 path
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][nt][e] = 0.f;
#pragma unroll
    for (int m = 0; m < MT16; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc16[m][n][e] = 0.f;

    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void gbl_void_t;
    auto stage_A = [&](int chunk, int tap_off) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            int off; bool ok;
            if constexpr (MODE == 0) off = a_desc(i, ok);
            else { off = a_off[i]; ok = (a_valid >> i) & 1; }
            const T* src = ok ? xg + off + tap_off + chunk * CK : reinterpret_cast<const T*>(fea_zero_line);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(ldsA + (i * 4 + wave) * 1024), 16, 0, 0);
        }
    };
    auto stage_W = [&](int kc, int buf) {
#pragma unroll
        for (int j = 0; j < W_PASSES; ++j) {
            const int row = j * 32 + (tid >> 3);
            const T* src = wg + ((long long)kc * g.Cout + cout0 + row) * CK + ((sc ^ swz(row)) * EPC);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(ldsW + buf * W_BYTES + (j * 4 + wave) * 1024), 16, 0, 0);
        }
    };
    // One tap (or one gathered k-chunk): 4 k-steps of 16.  bf16: the LDS reads
    // are inline asm, software-pipelined one k-step ahead of the MFMAs and
    // retired by counted lgkmcnt (hipcc otherwise schedules each read right in
    // front of its first use and exposes the LDS latency every k-step).
    const uint32_t lds_a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ldsA;
    const uint32_t lds_w0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ldsW;
    auto compute = [&](int row_shift, int buf) {
        if constexpr (BF) {
            // 16x16x32 MFMA.  A tap is S = 2*MT16 "slots" (k-step ks = s / MT16, M tile
            // m = s % MT16), each 4 MFMAs sharing one A fragment.  A fragments run
            // through a 4-deep ring read 3 slots ahead; the four B fragments of
            // k-step 1 are read during k-step 0.  TapSched gives the counted waits.
            typedef TapSched<MT16> TS;
            constexpr int S = TS::S;
            uint32_t aa[MT16], ba[4];
#pragma unroll
            for (int m = 0; m < MT16; ++m) {
                const int row = a_row16[m] + row_shift;
                aa[m] = lds_a0 + row * 128 + ((q16 ^ swz(row)) << 4);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) ba[n] = lds_w0 + buf * W_BYTES + b_row16[n] * 128 + ((q16 ^ swz(b_row16[n])) << 4);
            u32x4 Ar[4], Br[2][4];
            auto rd = [&](u32x4& dst, uint32_t addr) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
            };
            sfor<0, 4>([&](auto nc) { rd(Br[0][decltype(nc)::value], ba[decltype(nc)::value]); });
            sfor<0, 3>([&](auto mc) { rd(Ar[decltype(mc)::value], aa[decltype(mc)::value]); });
            sfor<0, S>([&](auto sc_) {
                constexpr int s_ = decltype(sc_)::value, ks = s_ / MT16, m = s_ % MT16;
                if constexpr (m == 0)
                    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(Ar[s_ % 4]), "+v"(Br[ks][0]), "+v"(Br[ks][1]), "+v"(Br[ks][2]), "+v"(Br[ks][3])
                                 : "n"(TS::N(s_)));
                else
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(Ar[s_ % 4]) : "n"(TS::N(s_)));
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc16[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Ar[s_ % 4]),
                                                                          __builtin_bit_cast(bf16x8, Br[ks][n]), acc16[m][n], 0, 0, 0);
                sfor<0, TS::nb(s_)>([&](auto jc) {
                    constexpr int j = TS::bfirst(s_) + decltype(jc)::value;
                    rd(Br[1][j], ba[j] ^ (1u << 6));
                });
                if constexpr (s_ + 3 < S) rd(Ar[(s_ + 3) % 4], aa[(s_ + 3) % MT16] ^ ((uint32_t)((s_ + 3) / MT16) << 6));
            });
        } else {
            int abase[MT], aswz[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = a_row[m] + row_shift;
                abase[m] = row * 128;
                aswz[m] = swz(row);
            }
            const char* wb = ldsW + buf * W_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 2 * ks + h;
                u32x4 af[MT], bf[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const u32x4*>(ldsA + abase[m] + ((c ^ aswz[m]) << 4));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bf[nt] = *reinterpret_cast<const u32x4*>(wb + b_addr[nt] + ((c ^ b_swz[nt]) << 4));
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) mma16(acc[m][nt], af[m], bf[nt], (const T*)nullptr);
            }
        }
    };

    // MODE 0, bf16, weights through LDS (BN = 128): the taps of a chunk are chained.
    // The A ring runs on into the next tap (its first three fragments are read during
    // the last three slots), the next tap's weight tile is written to LDS in mid-tap
    // (`mid`), so that after the per-tap barrier only the four B fragments of k-step 0
    // are outstanding before the first MFMA.
    u32x4 ArC[4];
    auto a_addr16 = [&](int m, int row_shift) -> uint32_t {
        const int row = a_row16[m] + row_shift;
        return lds_a0 + row * 128 + ((q16 ^ swz(row)) << 4);
    };
    auto rd128 = [&](u32x4& dst, uint32_t addr) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
    };
    auto compute_cont = [&](int row_shift, int next_shift, int buf, auto&& mid) {
        typedef TapSched<MT16> TS;
        constexpr int S = TS::S;
        uint32_t aa[MT16], aan[3], ba[4];
#pragma unroll
        for (int m = 0; m < MT16; ++m) aa[m] = a_addr16(m, row_shift);
#pragma unroll
        for (int m = 0; m < 3; ++m) aan[m] = a_addr16(m, next_shift);
#pragma unroll
        for (int n = 0; n < 4; ++n) ba[n] = lds_w0 + buf * W_BYTES + b_row16[n] * 128 + ((q16 ^ swz(b_row16[n])) << 4);
        u32x4 Br[2][4];
        sfor<0, 4>([&](auto nc) { rd128(Br[0][decltype(nc)::value], ba[decltype(nc)::value]); });
        sfor<0, S>([&](auto sc_) {
            constexpr int s_ = decltype(sc_)::value, ks = s_ / MT16, m = s_ % MT16;
            // reads issued after A(s): those of the two previous slots (slot 0: everything
            // must have landed, the B fragments were issued last).  LDS writes of `mid`
            // are not counted: the waits around it only get more conservative.
            constexpr int N = s_ == 0 ? 0 : (s_ == 1 ? TS::nb(0) + 1 : (s_ == 2 ? TS::nb(0) + TS::nb(1) + 2 : TS::nb(s_ - 2) + TS::nb(s_ - 1) + 2));
            if constexpr (s_ == MT16) mid();
            if constexpr (m == 0)
                asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(ArC[s_ % 4]), "+v"(Br[ks][0]), "+v"(Br[ks][1]), "+v"(Br[ks][2]), "+v"(Br[ks][3])
                             : "n"(N) : "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ArC[s_ % 4]) : "n"(N));
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc16[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ArC[s_ % 4]),
                                                                      __builtin_bit_cast(bf16x8, Br[ks][n]), acc16[m][n], 0, 0, 0);
            sfor<0, TS::nb(s_)>([&](auto jc) {
                constexpr int j = TS::bfirst(s_) + decltype(jc)::value;
                rd128(Br[1][j], ba[j] ^ (1u << 6));
            });
            if constexpr (s_ + 3 < S) rd128(ArC[(s_ + 3) % 4], aa[(s_ + 3) % MT16] ^ ((uint32_t)((s_ + 3) / MT16) << 6));
            else rd128(ArC[(s_ + 3) % 4], aan[s_ + 3 - S]);
        });
    };

    if constexpr (DB) {
        const int HW2 = TW + 2;
        constexpr int S = 2 * MT16;                 // slots per tap (multiple of 4: the A ring index carries across taps)
        // B fragments: ring of three half-taps (one k-step of 32 = 4 fragments each),
        // loaded two halves (= one tap) ahead of their use
        u32x4 Ar[4], Hb[3][4];
        auto rd = [&](u32x4& dst, uint32_t addr) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
        };
        const int nhalf = nchunks * 18;
        // packed weights: [kc][cout/16][ks][lane][8]
        const int wn_u = __builtin_amdgcn_readfirstlane(wn);      // provably wave-uniform: scalar base + lane offset addressing
        const T* wbase = wg + ((long long)((cout0 + wn_u * 64) >> 4) * 2) * 512;
        const long long w_kc_stride = (long long)(g.Cout >> 4) * 2 * 512;
        auto load_half = [&](int half, u32x4 (&b)[4]) {
            half = half < nhalf ? half : nhalf - 1;           // past the end: a harmless re-load
            const T* src = wbase + (half >> 1) * w_kc_stride + (half & 1) * 512;
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const u32x4*>(src + n * 1024 + lane * 8);
        };
        // Halo row of M tile m (p = p0 + 16 m): TW = 32: row0 + (m>>1)*HW2 + (m&1)*16, and a
        // step of 16 rows leaves the swizzle ((row>>1)&7) unchanged, so odd tiles are the
        // even tile's address + 2048 (an immediate); TW = 16: row0 + m*HW2.
        constexpr bool PAIR = TWS == 5;
        constexpr int NAA = PAIR ? MT16 / 2 : MT16;
        const int a_row0 = a_row16[0];
        auto a_addr = [&](int i, int row_shift) -> uint32_t {       // i-th address register
            const int row = a_row0 + row_shift + i * HW2;
            return lds_a0 + row * 128 + ((q16 ^ swz(row)) << 4);
        };
        auto rd_a = [&](u32x4& dst, const uint32_t* ad, auto mc, uint32_t flip) {
            constexpr int m = decltype(mc)::value;
            if constexpr (PAIR) {
                if constexpr (m & 1) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(dst) : "v"(ad[m >> 1] ^ flip) : "memory");
                else                 asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(ad[m >> 1] ^ flip) : "memory");
            } else {
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(ad[m] ^ flip) : "memory");
            }
        };
        // one tap: S slots of 4 MFMAs; A fragments through a 4-deep ring read 3 slots
        // ahead, continuing into the next tap (after tap 8 the three look-ahead reads
        // fetch tap 0 of the tile being replaced: never used, re-read after staging).
        // half0 = index of this tap's first half-tap; its fragments are in Hb[BI], Hb[(BI+1)%3].
        auto tap_mma = [&](int tap, int half0, auto bic) {
            constexpr int BI = decltype(bic)::value;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int rs = dy * HW2 + dx;
            const int tn = tap == 8 ? 0 : tap + 1, dyn = tn / 3, dxn = tn - dyn * 3;
            const int rsn = dyn * HW2 + dxn;
            uint32_t aa[NAA], aan[PAIR ? 2 : 3];
#pragma unroll
            for (int i = 0; i < NAA; ++i) aa[i] = a_addr(i, rs);
#pragma unroll
            for (int i = 0; i < (PAIR ? 2 : 3); ++i) aan[i] = a_addr(i, rsn);
            sfor<0, S>([&](auto sc_) {
                constexpr int s_ = decltype(sc_)::value, ks = s_ / MT16, m = s_ % MT16;
                if constexpr (s_ == 0) load_half(half0 + 2, Hb[(BI + 2) % 3]);
                if constexpr (s_ == MT16) load_half(half0 + 3, Hb[BI]);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(Ar[s_ % 4]));
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc16[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Ar[s_ % 4]),
                                                                          __builtin_bit_cast(bf16x8, Hb[(BI + ks) % 3][n]), acc16[m][n], 0, 0, 0);
                if constexpr (s_ + 3 < S)
                    rd_a(Ar[(s_ + 3) % 4], aa, std::integral_constant<int, (s_ + 3) % MT16>{}, (uint32_t)((s_ + 3) / MT16) << 6);
                else
                    rd_a(Ar[(s_ + 3) % 4], aan, std::integral_constant<int, s_ + 3 - S>{}, 0u);
            });
        };
        load_half(0, Hb[0]);
        load_half(1, Hb[1]);
        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();                       // previous chunk fully consumed
            stage_A(ch, 0);
            __syncthreads();
            {
                uint32_t a0[PAIR ? 2 : 3];
#pragma unroll
                for (int i = 0; i < (PAIR ? 2 : 3); ++i) a0[i] = a_addr(i, 0);
                sfor<0, 3>([&](auto mc) { rd_a(Ar[decltype(mc)::value], a0, mc, 0u); });
            }
            const int h0 = ch * 18;
#pragma unroll 1
            for (int t3 = 0; t3 < 3; ++t3) {
                tap_mma(3 * t3, h0 + 6 * t3, std::integral_constant<int, 0>{});
                tap_mma(3 * t3 + 1, h0 + 6 * t3 + 2, std::integral_constant<int, 2>{});
                tap_mma(3 * t3 + 2, h0 + 6 * t3 + 4, std::integral_constant<int, 1>{});
            }
        }
    } else if constexpr (MODE == 3) {
        // g.x: fp32 [N,C,H,W]; g.Cin = C; k = tap*C + c; packed weights [ks][cout/16][lane][8]
        constexpr int HW2 = TW + 2, HALO = (TH + 2) * HW2;
        const int C = g.Cin;
        const int KS = (9 * C + 31) >> 5;                  // 1 or 2 k-steps of 32
        float* hal = reinterpret_cast<float*>(smem);
        const float* __restrict__ xin = (const float*)g.x + (long long)n_img * g.x_img_stride;
        // all of a thread's halo loads are issued before the first is stored (a rolled loop waits for each load in turn:
        // four exposed HBM latencies per workgroup of a kernel that is otherwise one output store)
        constexpr int HIT = (7 * HALO + 255) / 256;           // 9 C <= 64  =>  C <= 7
        float hv[HIT];
#pragma unroll
        for (int i = 0; i < HIT; ++i) {
            const int e = tid + i * 256;
            const int c = e / HALO, rr = e - c * HALO;
            const int hy = rr / HW2, hx = rr - hy * HW2;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = e < C * HALO && gy >= 0 && gy < H && gx >= 0 && gx < W;
            hv[i] = ok ? xin[((long long)c * H + gy) * W + gx] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < HIT; ++i) {
            const int e = tid + i * 256;
            if (e < C * HALO) hal[e] = hv[i];
        }
        u32x4 Bf[2][4];
        int koff[2][8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int ksc = ks < KS ? ks : 0;
                Bf[ks][n] = *reinterpret_cast<const u32x4*>(wg + (((long long)ksc * (g.Cout >> 4) + ((cout0 + wn * 64) >> 4) + n) * 64 + lane) * 8);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = ks * 32 + q16 * 8 + j;
                const int tap = C == 3 ? k / 3 : k / C, c = k - tap * C;    // (RGB: constant divisor; 16 runtime divisions otherwise)
                const int dy = tap / 3, dx = tap - dy * 3;
                // k >= 9 C: the packed weight is zero (pack_conv3x3_first_kernel), any finite operand will do: no branch
                koff[ks][j] = k < 9 * C ? c * HALO + dy * HW2 + dx : 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT16; ++m) {
            const int p = wm * (MT * 32) + m * 16 + r16;
            const int pb = (p >> tws) * HW2 + (p & (TW - 1));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks < KS) {
                    bf16x8 af;
#pragma unroll
                    for (int j = 0; j < 8; ++j) af[j] = (__bf16)hal[pb + koff[ks][j]];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc16[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, Bf[ks][n]), acc16[m][n], 0, 0, 0);
                }
            }
        }
    } else if constexpr (MODE == 0 && BF && TWS == 5 && BN == 128 && IG_FAST_ADDR) {
        // 8x32 tile, weights through LDS (BN = 128): as the branch below, with every LDS address precomputed.  The loop was
        // ISSUE-bound: ~90 VALU of address arithmetic per tap and wave (row, swizzle, shift, add for 11 A and 4 B addresses)
        // beside 64 MFMAs that hold the issue port for 8 of their 16 cycles each -- two waves per SIMD then need > 2,048 issue cycles
        // per 2,048 cycles of matrix pipe (removing the per-tap barriers or the halo staging changed nothing).  A
        // wave's 128 pixels are 4 tile rows x 2 halves of 16: m tile -> halo row wm*4 + (m >> 1), x = (m & 1)*16 + r16, and
        // +16 halo rows leave the swizzle unchanged, so odd tiles are an immediate offset (2,048 B) on the even tile's address:
        // 6 (halo rows incl. dy) x 3 (dx) base addresses cover every tap; taps are compile-time constants (loop unrolled).
        uint32_t abase[6][3], bbase[4];
        // LDS-DMA as buffer loads: 32-bit per-lane offsets (no 64-bit address arithmetic for the compiler to hoist out of the
        // unrolled taps and spill), the tap / chunk part is the scalar offset operand, padding lanes take an out-of-range offset
        int woff[W_PASSES];                           // a lane's byte offset inside one tap's [Cout][64] weight slab
#pragma unroll
        for (int j = 0; j < W_PASSES; ++j) {
            const int row = j * 32 + (tid >> 3);
            woff[j] = ((cout0 + row) * CK + ((sc ^ swz(row)) * EPC)) * (int)sizeof(T);
        }
        const int wslab = g.Cout * CK * (int)sizeof(T), wbytes = 9 * g.Cin * g.Cout * (int)sizeof(T), abytes = (int)(g.x_img_stride * sizeof(T));
        auto stage_Wf = [&](int kc, auto bufc) {
#ifdef IG_EXP_NO_STAGE_W      // EXPERIMENT (wrong results): the weight tile is staged once = what its LDS fill traffic costs
            if (kc > 1) return;
#endif
#pragma unroll
            for (int j = 0; j < W_PASSES; ++j)
                ig_buffer_lds16(wg, wbytes, ldsW + decltype(bufc)::value * W_BYTES + (j * 4 + wave) * 1024, woff[j], kc * wslab);
        };
        int aoff[A_PASSES];                           // a lane's byte offset inside the image (chunk 0), out of range for padding
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            bool ok;
            const int off = a_desc(i, ok);
            aoff[i] = ok ? off * (int)sizeof(T) : (int)0xfffffff0u;
        }
        auto stage_Af = [&](int chunk) {
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i)
                ig_buffer_lds16(xg, abytes, ldsA + (i * 4 + wave) * 1024, aoff[i], chunk * CK * (int)sizeof(T));
        };
#pragma unroll
        for (int hr = 0; hr < 6; ++hr)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int row = (wm * 4 + hr) * (TW + 2) + r16 + dx;
                abase[hr][dx] = lds_a0 + row * 128 + ((q16 ^ swz(row)) << 4);
            }
#pragma unroll
        for (int n = 0; n < 4; ++n) bbase[n] = lds_w0 + b_row16[n] * 128 + ((q16 ^ swz(b_row16[n])) << 4);
        auto rdi = [&](u32x4& dst, uint32_t addr, auto offc) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(offc)::value) : "memory");
        };
        auto rd_a = [&](u32x4& dst, auto dyc, auto dxc, auto mc, auto ksc) {          // A fragment of m tile `m`, k-step ks, tap (dy, dx)
            constexpr int dy = decltype(dyc)::value, dx = decltype(dxc)::value, m = decltype(mc)::value, ks = decltype(ksc)::value;
            if constexpr (ks == 0) rdi(dst, abase[(m >> 1) + dy][dx], std::integral_constant<int, (m & 1) * 2048>{});
            else                   rdi(dst, abase[(m >> 1) + dy][dx] ^ (1u << 6), std::integral_constant<int, (m & 1) * 2048>{});
        };
        auto compute_fast = [&](auto tapc) {
            constexpr int tap = decltype(tapc)::value, dy = tap / 3, dx = tap % 3, buf = tap & 1;
            constexpr int tn = tap == 8 ? 0 : tap + 1, dyn = tn / 3, dxn = tn % 3;      // (after tap 8: a harmless look-ahead)
            typedef TapSched<MT16> TS;
            constexpr int S = TS::S;
            u32x4 Br[2][4];
            sfor<0, 4>([&](auto nc) { rdi(Br[0][decltype(nc)::value], bbase[decltype(nc)::value], std::integral_constant<int, buf * W_BYTES>{}); });
            sfor<0, S>([&](auto sc_) {
                constexpr int s_ = decltype(sc_)::value, ks = s_ / MT16, m = s_ % MT16;
                constexpr int N = s_ == 0 ? 0 : (s_ == 1 ? TS::nb(0) + 1 : (s_ == 2 ? TS::nb(0) + TS::nb(1) + 2 : TS::nb(s_ - 2) + TS::nb(s_ - 1) + 2));
                if constexpr (m == 0)
                    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(ArC[s_ % 4]), "+v"(Br[ks][0]), "+v"(Br[ks][1]), "+v"(Br[ks][2]), "+v"(Br[ks][3])
                                 : "n"(N) : "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ArC[s_ % 4]) : "n"(N));
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc16[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ArC[s_ % 4]),
                                                                          __builtin_bit_cast(bf16x8, Br[ks][n]), acc16[m][n], 0, 0, 0);
                sfor<0, TS::nb(s_)>([&](auto jc) {
                    constexpr int j = TS::bfirst(s_) + decltype(jc)::value;
                    rdi(Br[1][j], bbase[j] ^ (1u << 6), std::integral_constant<int, buf * W_BYTES>{});
                });
                if constexpr (s_ + 3 < S)
                    rd_a(ArC[(s_ + 3) % 4], std::integral_constant<int, dy>{}, std::integral_constant<int, dx>{},
                         std::integral_constant<int, (s_ + 3) % MT16>{}, std::integral_constant<int, (s_ + 3) / MT16>{});
                else
                    rd_a(ArC[(s_ + 3) % 4], std::integral_constant<int, dyn>{}, std::integral_constant<int, dxn>{},
                         std::integral_constant<int, s_ + 3 - S>{}, std::integral_constant<int, 0>{});
            });
        };
        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();                       // previous chunk fully consumed
            stage_Wf(ch * 9, std::integral_constant<int, 0>{});
#ifdef IG_EXP_NO_STAGE_A
            if (ch == 0)       // EXPERIMENT (wrong results): see the branch below
#endif
            stage_Af(ch);
            __syncthreads();                       // (drains the LDS-DMA: vmcnt(0) before the barrier)
            sfor<0, 3>([&](auto mc) {
                rd_a(ArC[decltype(mc)::value], std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, mc, std::integral_constant<int, 0>{});
            });
            sfor<0, 9>([&](auto tapc) {
                constexpr int tap = decltype(tapc)::value;
                // the buffer of tap+1 was last read in tap-1, which every wave has left
                if constexpr (tap < 8) stage_Wf(ch * 9 + tap + 1, std::integral_constant<int, (tap + 1) & 1>{});
                compute_fast(tapc);
                if constexpr (tap < 8) __syncthreads();
            });
        }
    } else if (MODE == 0 && BF) {
        const int HW2 = TW + 2;
#ifdef IG_STAMPS
        unsigned long long iph[6] = {0, 0, 0, 0, 0, 0}, itprev = __builtin_amdgcn_s_memtime();
#endif
        for (int ch = 0; ch < nchunks; ++ch) {
            ISTAMP(5);
            __syncthreads();                       // previous chunk fully consumed
            ISTAMP(0);
            stage_W(ch * 9, 0);
#ifdef IG_EXP_NO_STAGE_A
            if (ch == 0)       // EXPERIMENT (wrong results): no halo staging after the first chunk = upper bound of what hiding it could gain
#endif
            stage_A(ch, 0);
            ISTAMP(1);
            __syncthreads();                       // (drains the LDS-DMA: vmcnt(0) before the barrier)
            ISTAMP(2);
#pragma unroll
            for (int m = 0; m < 3; ++m) rd128(ArC[m], a_addr16(m, 0));
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                // the buffer of tap+1 was last read in tap-1, which every wave has left
                if (tap < 8) stage_W(ch * 9 + tap + 1, (tap + 1) & 1);
                const int dy = tap / 3, dx = tap - dy * 3;
                const int tn = tap == 8 ? 0 : tap + 1, dyn = tn / 3, dxn = tn - dyn * 3;   // (after tap 8: a harmless look-ahead)
                compute_cont(dy * HW2 + dx, dyn * HW2 + dxn, tap & 1, [&]() {});
                ISTAMP(3);
#ifndef IG_EXP_NO_TAP_BARRIER          // EXPERIMENT (racy, wrong results): upper bound of what removing the per-tap weight barrier could gain
                if (tap < 8) __syncthreads();
#else
                if (tap < 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                ISTAMP(4);
            }
        }
#ifdef IG_STAMPS
        if (lane == 0 && g.dbg) for (int k = 0; k < 6; ++k) atomicAdd(&g.dbg[(wave) * 6 + k], iph[k]);
#endif
    } else if (MODE == 0) {
        const int HW2 = TW + 2;
        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();                       // previous chunk fully consumed
            stage_W(ch * 9, 0);
            stage_A(ch, 0);
            __syncthreads();
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                if (tap < 8) stage_W(ch * 9 + tap + 1, (tap + 1) & 1);
                const int dy = tap / 3, dx = tap - dy * 3;
                compute(dy * HW2 + dx, tap & 1);
                if (tap < 8) __syncthreads();
            }
        }
    } else {
        const int ntaps = MODE == 2 ? 4 : 1;
        const int nk = ntaps * nchunks;
        // kc = tap * nchunks + chunk
        if constexpr (GR) {
        // The next k-chunk is on its way while this one is multiplied: its pixels travel global -> registers (written to LDS
        // after the barrier, in the layout the LDS-DMA path writes: wave-instruction j fills 1 KB linearly, swizzle on the
        // source), its weights by LDS-DMA into the other of two weight tiles.  With both staged by DMA between two barriers
        // the chunk's whole HBM latency was exposed in every chunk (only the CU's other workgroup covered it).
        u32x4 ra[A_PASSES];
        auto load_A = [&](int kc) {
            const int tap = MODE == 2 ? kc / nchunks : 0;
            const int ch = kc - tap * nchunks;
            const int tap_off = MODE == 2 ? ((tap >> 1) * (2 * W) + (tap & 1)) * g.ldx : 0;
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                ra[i] = ((a_valid >> i) & 1) ? *reinterpret_cast<const u32x4*>(xg + a_off[i] + tap_off + ch * CK) : z;
            }
        };
        load_A(0);
        stage_W(0, 0);
        for (int kc = 0; kc < nk; ++kc) {
            __syncthreads();                       // previous k-chunk consumed; this chunk's weights have landed
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) *reinterpret_cast<u32x4*>(ldsA + (i * 4 + wave) * 1024 + lane * 16) = ra[i];
            __syncthreads();
            if (kc + 1 < nk) { load_A(kc + 1); stage_W(kc + 1, (kc + 1) & 1); }
            compute(0, kc & 1);
        }
        } else {
        for (int kc = 0; kc < nk; ++kc) {
            const int tap = MODE == 2 ? kc / nchunks : 0;
            const int ch = kc - tap * nchunks;
            const int tap_off = MODE == 2 ? ((tap >> 1) * (2 * W) + (tap & 1)) * g.ldx : 0;
            __syncthreads();                       // previous k-chunk consumed
#ifdef IG_EXP_NO_STAGE_GEMM
            if (kc == 0) {       // EXPERIMENT (wrong results): staging only for the first chunk = what hiding it completely could gain
#endif
            stage_W(kc, 0);
            stage_A(ch, tap_off);
#ifdef IG_EXP_NO_STAGE_GEMM
            }
#endif
            __syncthreads();
            compute(0, 0);
        }
        }
    }
    __syncthreads();                               // LDS is reused by the epilogue

    // ---- epilogue -----------------------------------------------------------
    const bool full = (y0 + TH <= H) && (x0 + TW <= W);
    // output addressing
    T* __restrict__ yg = (T*)g.y + (long long)n_img * g.y_img_stride;
    const int ab = EPI == 1 ? cout0 / g.Cup : 0;
    const int c_off = EPI == 1 ? cout0 - ab * g.Cup : cout0;
    auto out_pix = [&](int p, bool& ok) -> int {
        const int gy = y0 + (p >> tws), gx = x0 + (p & (TW - 1));
        ok = gy < H && gx < W;
        return EPI == 1 ? (2 * gy + (ab >> 1)) * (2 * W) + 2 * gx + (ab & 1) : gy * W + gx;
    };
    float* st = reinterpret_cast<float*>(smem + EPI_BYTES);

    if constexpr (BF) {
        // 16x16 accumulator tiles: element j of acc16[m][n] is pixel
        // wm*(MT*32) + m*16 + 4*q16 + j, column wn*64 + n*16 + r16.
        float bv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = cout0 + wn * 64 + n * 16 + r16;
            bv[n] = g.bias ? g.bias[EPI == 1 ? col % g.Cup : col] : 0.f;
        }
        // (vector adds / fmas: v_pk_add_f32, v_pk_fma_f32 -- the epilogue is VALU-issue bound on the
        // 64- and 128-channel layers, where it is a quarter of a workgroup's life)
#pragma unroll
        for (int m = 0; m < MT16; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f32x4 b4 = {bv[n], bv[n], bv[n], bv[n]};
                acc16[m][n] += b4;
                if constexpr (EPI == 3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc16[m][n][j] = fmaxf(acc16[m][n][j], 0.f);
                }
            }

        if (EPI == 0 && g.stats) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float s, q;
                if (full) {
                    f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < MT16; ++m) { s4 += acc16[m][n]; q4 = __builtin_elementwise_fma(acc16[m][n], acc16[m][n], q4); }
                    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
                    q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
                } else {
                    s = 0.f; q = 0.f;
#pragma unroll
                    for (int m = 0; m < MT16; ++m)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int p = wm * (MT * 32) + m * 16 + 4 * q16 + j;
                            const bool ok = (y0 + (p >> tws) < H) && (x0 + (p & (TW - 1)) < W);
                            const float v = ok ? acc16[m][n][j] : 0.f;
                            s += v; q += v * v;
                        }
                }
                s = row_sum4(s); q = row_sum4(q);
                if (q16 == 0) {
                    st[(wm * BN + wn * 64 + n * 16 + r16) * 2 + 0] = s;
                    st[(wm * BN + wn * 64 + n * 16 + r16) * 2 + 1] = q;
                }
            }
        }
        // stage the tile as [pixel][BN] bf16 in LDS, then store whole rows
#pragma unroll
        for (int m = 0; m < MT16; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = wm * (MT * 32) + m * 16 + 4 * q16 + j;
                    *reinterpret_cast<T*>(smem + p * EPI_STRIDE + (wn * 64 + n * 16 + r16) * 2) = from_f32<T>(acc16[m][n][j]);
                }
        __syncthreads();
        if (EPI == 0 && g.stats && tid < BN) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < WM; ++w_) { s += st[(w_ * BN + tid) * 2]; q += st[(w_ * BN + tid) * 2 + 1]; }
            float* dst = g.stats + ((long long)ptile * g.Cout + cout0 + tid) * 2;
            dst[0] = s; dst[1] = q;
        }
        constexpr int CPR = BN / 8;               // 16-byte chunks per row
        if constexpr (EPI == 2) {
            // thread -> fixed channel chunk c = tid % CPR, pixels tid / CPR + i * (256 / CPR)
            const int c = tid % CPR;
            // 8 VALU per element: sum g and sum g*(y - mean) are accumulated (centred per element: with |mean| >> std of y
            // the difference sum g*y - mean * sum g would cancel in fp32), the invstd factor is applied once per tile and
            // channel at the end (as in conv_p.hip); relu'(bn(y)) = [y*scale + shift > 0] needs no max
            float csc[8], csh[8], cmn[8], a0[8], a1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = cout0 + c * 8 + j;
                csc[j] = g.bn_scale[ch]; csh[j] = g.bn_shift[ch]; cmn[j] = g.bn_mean[ch];
                a0[j] = 0.f; a1[j] = 0.f;
            }
            const T* __restrict__ yb = (const T*)g.ybn + (long long)n_img * H * W * g.ldybn + cout0 + c * 8;
            // all y loads of the thread in flight at once (the accumulators are dead: 64 registers are free)
            constexpr int GRP = EPI2_GRP < CPR ? EPI2_GRP : CPR;
#pragma unroll
            for (int i0 = 0; i0 < CPR; i0 += GRP) {
                u32x4 yv[GRP]; int pixv[GRP]; bool okv[GRP];
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const int p = ((i0 + u) * 256 + tid) / CPR;
                    pixv[u] = out_pix(p, okv[u]);
                    yv[u] = *reinterpret_cast<const u32x4*>(yb + (long long)(okv[u] ? pixv[u] : 0) * g.ldybn);
                }
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const int p = ((i0 + u) * 256 + tid) / CPR;
                    if (okv[u]) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + p * EPI_STRIDE + c * 16);
                        *reinterpret_cast<u32x4*>(yg + (long long)pixv[u] * g.ldy + c_off + c * 8) = v;
                        const bf16x8 dv = __builtin_bit_cast(bf16x8, v), yy = __builtin_bit_cast(bf16x8, yv[u]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float yf = (float)yy[j];
                            const float gg = __builtin_fmaf(yf, csc[j], csh[j]) > 0.f ? (float)dv[j] : 0.f;   // bn_act(y) > 0 (elementwise.hip)
                            a0[j] += gg; a1[j] = __builtin_fmaf(gg, yf - cmn[j], a1[j]);
                        }
                    }
                }
            }
            // lanes with equal lane % CPR hold the same channels: sum them on the VALU
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (CPR == 8) {
                    a0[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0[j]), 0x128, 0xf, 0xf, false));
                    a1[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1[j]), 0x128, 0xf, 0xf, false));
                }
                a0[j] = row_sum4(a0[j]); a1[j] = row_sum4(a1[j]);
            }
            if (lane < CPR) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    st[((wave * BN) + lane * 8 + j) * 2 + 0] = a0[j];
                    st[((wave * BN) + lane * 8 + j) * 2 + 1] = a1[j];
                }
            }
            __syncthreads();
            if (tid < BN) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < 4; ++w_) { s0 += st[(w_ * BN + tid) * 2]; s1 += st[(w_ * BN + tid) * 2 + 1]; }
                float* dst = g.red_part + ((long long)ptile * g.Cout + cout0 + tid) * 2;
                dst[0] = s0; dst[1] = g.bn_invstd[cout0 + tid] * s1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < CPR; ++i) {
                const int q = i * 256 + tid;
                const int p = q / CPR, c = q - p * CPR;
                bool ok; const int pix = out_pix(p, ok);
                if (ok) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + p * EPI_STRIDE + c * 16);
                    *reinterpret_cast<u32x4*>(yg + (long long)pix * g.ldy + c_off + c * 8) = v;
                }
            }
            if constexpr (EPI == 3) {
                if (g.pool) {
                    // 2x2 max-pool of the staged tile (tile origins are even, H and W are even:
                    // every window lies inside one tile and is either whole or outside the image)
                    constexpr int TWP = TW / 2;
                    T* __restrict__ pg = (T*)g.pool + (long long)n_img * (H / 2) * (W / 2) * g.ldpool + cout0;
#pragma unroll
                    for (int i = 0; i < (CPR + 3) / 4; ++i) {
                        const int q = i * 256 + tid;
                        const int pp = q / CPR, c = q - pp * CPR;
                        const int py = pp / TWP, px = pp - py * TWP;
                        if (pp < 64 && y0 + 2 * py < H && x0 + 2 * px < W) {
                            const int p00 = (2 * py) * TW + 2 * px;
                            bf16x8 v[4];
                            v[0] = *reinterpret_cast<const bf16x8*>(smem + p00 * EPI_STRIDE + c * 16);
                            v[1] = *reinterpret_cast<const bf16x8*>(smem + (p00 + 1) * EPI_STRIDE + c * 16);
                            v[2] = *reinterpret_cast<const bf16x8*>(smem + (p00 + TW) * EPI_STRIDE + c * 16);
                            v[3] = *reinterpret_cast<const bf16x8*>(smem + (p00 + TW + 1) * EPI_STRIDE + c * 16);
                            bf16x8 o;
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                o[j] = (__bf16)fmaxf(fmaxf((float)v[0][j], (float)v[1][j]), fmaxf((float)v[2][j], (float)v[3][j]));
                            *reinterpret_cast<bf16x8*>(pg + ((long long)(y0 / 2 + py) * (W / 2) + x0 / 2 + px) * g.ldpool + c * 8) = o;
                        }
                    }
                }
            }
        }
    } else {
        float bv[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = cout0 + wn * 64 + nt * 32 + r;
            bv[nt] = g.bias ? g.bias[EPI == 1 ? col % g.Cup : col] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][nt][e] += bv[nt];

        if (EPI == 0 && g.stats) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = acc[m][nt][e];
                        if (!full) {
                            const int p = wm * (MT * 32) + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            const bool ok = (y0 + (p >> tws) < H) && (x0 + (p & (TW - 1)) < W);
                            v = ok ? v : 0.f;
                        }
                        s += v; q += v * v;
                    }
                s += __shfl_xor(s, 32);
                q += __shfl_xor(q, 32);
                if (h == 0) {
                    st[(wm * BN + wn * 64 + nt * 32 + r) * 2 + 0] = s;
                    st[(wm * BN + wn * 64 + nt * 32 + r) * 2 + 1] = q;
                }
            }
            __syncthreads();
            if (tid < BN) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < WM; ++w_) { s += st[(w_ * BN + tid) * 2]; q += st[(w_ * BN + tid) * 2 + 1]; }
                float* dst = g.stats + ((long long)ptile * g.Cout + cout0 + tid) * 2;
                dst[0] = s; dst[1] = q;
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int p = wm * (MT * 32) + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                bool ok; const int pix = out_pix(p, ok);
                if (ok) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        yg[(long long)pix * g.ldy + c_off + wn * 64 + nt * 32 + r] = from_f32<T>(acc[m][nt][e]);
                }
            }
    }
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
static inline int tile_shift(int W) { return W >= 32 ? 5 : 4; }

extern "C" int fea_conv_num_tiles(int N, int H, int W) {
    const int tws = tile_shift(W), TW = 1 << tws, TH = 256 >> tws;
    return N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
}

template <typename T, int MODE, int EPI>
static int launch_igemm(IgemmArgs& a, int N, hipStream_t s) {
    const int tws = tile_shift(a.W), TW = 1 << tws, TH = 256 >> tws;
    a.tw_shift = tws;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    const bool bn128 = (a.Cout % 128 == 0) && !(EPI == 1 && a.Cup % 128 != 0);   // MODE 0: must agree with fea_conv3x3_frag_pack
    const int BN = bn128 ? 128 : 64;
    a.nct = a.Cout / BN;
    const long long nblk = (long long)N * a.tiles_x * a.tiles_y * a.nct;
    FEA_CHECK(nblk > 0 && nblk < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "igemm: grid of %lld blocks", nblk);
    dim3 grid((unsigned)nblk), block(256);
#ifdef IG_STAMPS
    a.dbg = getenv("IG_DBG_PTR") ? (unsigned long long*)strtoull(getenv("IG_DBG_PTR"), nullptr, 0) : nullptr;
#endif
    if (tws == 5) {
        if (bn128) hipLaunchKernelGGL((igemm_kernel<T, MODE, 128, EPI, 5>), grid, block, 0, s, a);
        else       hipLaunchKernelGGL((igemm_kernel<T, MODE, 64, EPI, 5>), grid, block, 0, s, a);
    } else {
        if (bn128) hipLaunchKernelGGL((igemm_kernel<T, MODE, 128, EPI, 4>), grid, block, 0, s, a);
        else       hipLaunchKernelGGL((igemm_kernel<T, MODE, 64, EPI, 4>), grid, block, 0, s, a);
    }
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

template <int MODE, int EPI>
static int igemm_dispatch(IgemmArgs& a, int N, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FEA_BF16) return launch_igemm<__bf16, MODE, EPI>(a, N, s);
    if (dtype == FEA_F32) return launch_igemm<float, MODE, EPI>(a, N, s);
    fea_set_error("igemm: bad dtype %d", dtype);
    return FEA_E_BADARG;
}

static int check_common(const char* who, const void* x, int ldx, const void* w, const void* y, int ldy,
                        int N, int H, int W, int Cin, int Cout, int dtype) {
    const int CK = dtype == FEA_BF16 ? 64 : 32;
    FEA_CHECK(dtype == FEA_BF16 || dtype == FEA_F32, FEA_E_BADARG, "%s: bad dtype %d", who, dtype);
    FEA_CHECK(x && w && y, FEA_E_BADARG, "%s: null pointer", who);
    FEA_CHECK(N > 0 && H > 0 && W > 0, FEA_E_BADARG, "%s: bad dims N=%d H=%d W=%d", who, N, H, W);
    FEA_CHECK(Cin > 0 && Cin % CK == 0, FEA_E_UNSUPPORTED_SHAPE, "%s: Cin=%d must be a multiple of %d", who, Cin, CK);
    FEA_CHECK(Cout > 0 && Cout % 64 == 0, FEA_E_UNSUPPORTED_SHAPE, "%s: Cout=%d must be a multiple of 64", who, Cout);
    FEA_CHECK(ldx >= Cin && ldx % 8 == 0 && ldy % 8 == 0, FEA_E_BADARG, "%s: ldx=%d ldy=%d", who, ldx, ldy);
    FEA_CHECK(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)w % 16 == 0), FEA_E_BADARG,
              "%s: pointers must be 16-byte aligned", who);
    // (input: BYTE offsets inside one image are 32-bit -- the halo pieces of the 128-column kernel are buffer loads of an image-sized descriptor)
    FEA_CHECK((long long)H * W * ldx * (dtype == FEA_BF16 ? 2 : 4) < (1ll << 31) && (long long)4 * H * W * ldy < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE,
              "%s: image too large for 32-bit in-image offsets", who);
    return FEA_OK;
}

extern "C" int fea_conv3x3(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy, float* stats,
                           int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    int rc = check_common("fea_conv3x3", x, ldx, w, y, ldy, N, H, W, Cin, Cout, dtype);
    if (rc) return rc;
    FEA_CHECK(ldy >= Cout, FEA_E_BADARG, "fea_conv3x3: ldy=%d < Cout=%d", ldy, Cout);
    IgemmArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats;
    a.ldx = ldx; a.ldy = ldy; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Cup = Cout;
    a.x_img_stride = (long long)H * W * ldx; a.y_img_stride = (long long)H * W * ldy;
    return igemm_dispatch<0, 0>(a, N, dtype, stream);
}

extern "C" int fea_conv3x3_dgrad_bnred(const void* dy, int lddy, const void* w_packed, void* da, int ldda,
                                       const void* y, int ldy_bn, const float* scale, const float* shift,
                                       const float* save_mean, const float* save_invstd, float* part,
                                       int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    int rc = check_common("fea_conv3x3_dgrad_bnred", dy, lddy, w_packed, da, ldda, N, H, W, Cin, Cout, dtype);
    if (rc) return rc;
    FEA_CHECK(dtype == FEA_BF16, FEA_E_UNSUPPORTED_SHAPE, "fea_conv3x3_dgrad_bnred: bf16 only (fp32: fea_conv3x3 + fea_bn_relu_bwd_reduce)");
    FEA_CHECK(ldda >= Cout && y && scale && shift && save_mean && save_invstd && part, FEA_E_BADARG, "fea_conv3x3_dgrad_bnred: bad args");
    FEA_CHECK(ldy_bn >= Cout && ldy_bn % 8 == 0 && ((uintptr_t)y % 16 == 0), FEA_E_BADARG, "fea_conv3x3_dgrad_bnred: y layout");
    IgemmArgs a{};
    a.x = dy; a.w = w_packed; a.bias = nullptr; a.y = da; a.stats = nullptr;
    a.ldx = lddy; a.ldy = ldda; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Cup = Cout;
    a.x_img_stride = (long long)H * W * lddy; a.y_img_stride = (long long)H * W * ldda;
    a.ybn = y; a.ldybn = ldy_bn; a.bn_scale = scale; a.bn_shift = shift; a.bn_mean = save_mean; a.bn_invstd = save_invstd;
    a.red_part = part;
    return launch_igemm<__bf16, 0, 2>(a, N, (hipStream_t)stream);
}

// ConvTranspose2d data gradient fused with the reduction of the BatchNorm backward that
// consumes it (the layer in front of the transposed conv is conv -> BN -> ReLU, models.py:31-33).
extern "C" int fea_upconv2x2_bwd_data_bnred(const void* dy, int lddy, const void* w, void* dx, int lddx,
                                            const void* y, int ldy_bn, const float* scale, const float* shift,
                                            const float* save_mean, const float* save_invstd, float* part,
                                            int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    const char* who = "fea_upconv2x2_bwd_data_bnred";
    int rc = check_common(who, dy, lddy, w, dx, lddx, N, H, W, Cout, Cin, dtype);
    if (rc) return rc;
    FEA_CHECK(dtype == FEA_BF16, FEA_E_UNSUPPORTED_SHAPE, "%s: bf16 only (fp32: fea_upconv2x2_bwd_data + fea_bn_relu_bwd_reduce)", who);
    FEA_CHECK(lddx >= Cin && y && scale && shift && save_mean && save_invstd && part, FEA_E_BADARG, "%s: bad args", who);
    FEA_CHECK(ldy_bn >= Cin && ldy_bn % 8 == 0 && ((uintptr_t)y % 16 == 0), FEA_E_BADARG, "%s: y layout", who);
    FEA_CHECK((long long)4 * H * W * lddy < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "%s: image too large", who);
    IgemmArgs a{};
    a.x = dy; a.w = w; a.bias = nullptr; a.y = dx; a.stats = nullptr;
    a.ldx = lddy; a.ldy = lddx; a.H = H; a.W = W; a.Cin = Cout; a.Cout = Cin; a.Cup = Cin;
    a.x_img_stride = (long long)4 * H * W * lddy; a.y_img_stride = (long long)H * W * lddx;
    a.ybn = y; a.ldybn = ldy_bn; a.bn_scale = scale; a.bn_shift = shift; a.bn_mean = save_mean; a.bn_invstd = save_invstd;
    a.red_part = part;
    return launch_igemm<__bf16, 2, 2>(a, N, (hipStream_t)stream);
}

extern "C" int fea_conv3x3_first(const float* x, const void* w_packed, const float* bias, void* y, int ldy, float* stats,
                                 int N, int C, int H, int W, int Cout, int dtype, void* stream) {
    FEA_CHECK(dtype == FEA_BF16, FEA_E_UNSUPPORTED_SHAPE, "fea_conv3x3_first: bf16 only (fp32: fea_im2col3x3_nchw + fea_conv1x1)");
    FEA_CHECK(x && w_packed && y, FEA_E_BADARG, "fea_conv3x3_first: null pointer");
    FEA_CHECK(N > 0 && H > 0 && W > 0, FEA_E_BADARG, "fea_conv3x3_first: bad dims");
    FEA_CHECK(C > 0 && 9 * C <= 64, FEA_E_UNSUPPORTED_SHAPE, "fea_conv3x3_first: C=%d (9*C must be <= 64)", C);
    FEA_CHECK(Cout > 0 && Cout % 64 == 0 && ldy >= Cout && ldy % 8 == 0, FEA_E_UNSUPPORTED_SHAPE, "fea_conv3x3_first: Cout=%d ldy=%d", Cout, ldy);
    FEA_CHECK(((uintptr_t)y % 16 == 0) && ((uintptr_t)w_packed % 16 == 0) && ((uintptr_t)x % 4 == 0), FEA_E_BADARG, "fea_conv3x3_first: alignment");
    FEA_CHECK((long long)H * W * ldy < (1ll << 31) && (long long)C * H * W < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "fea_conv3x3_first: image too large");
    {   // whole 8x32 tiles, 64 channels: the persistent kernel (its partial table has fea_conv3x3_first_rows() rows)
        int rc = FEA_OK;
        if (fea_first_conv_p(x, w_packed, bias, y, ldy, stats, 0, N, C, H, W, Cout, (hipStream_t)stream, &rc)) return rc;
    }
    IgemmArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.stats = stats;
    a.ldx = 0; a.ldy = ldy; a.H = H; a.W = W; a.Cin = C; a.Cout = Cout; a.Cup = Cout;
    a.x_img_stride = (long long)C * H * W; a.y_img_stride = (long long)H * W * ldy;
    return launch_igemm<__bf16, 3, 0>(a, N, (hipStream_t)stream);
}

// Inference with BatchNorm folded into weights and bias (eval mode, models.py:28-33 collapse to
// conv + bias + ReLU); optional MaxPool2d(2) output (models.py:24,38-41).
extern "C" int fea_conv3x3_bias_relu(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                                     void* pool, int ldpool, int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    const char* who = "fea_conv3x3_bias_relu";
    int rc = check_common(who, x, ldx, w, y, ldy, N, H, W, Cin, Cout, dtype);
    if (rc) return rc;
    FEA_CHECK(dtype == FEA_BF16, FEA_E_UNSUPPORTED_SHAPE, "%s: bf16 only (fp32: fea_conv3x3 + fea_bn_eval_coeffs + fea_bn_relu_pool)", who);
    FEA_CHECK(ldy >= Cout && bias, FEA_E_BADARG, "%s: ldy=%d < Cout=%d or no bias", who, ldy, Cout);
    FEA_CHECK(!pool || (H % 2 == 0 && W % 2 == 0 && ldpool >= Cout && ldpool % 8 == 0 && (uintptr_t)pool % 16 == 0), FEA_E_BADARG,
              "%s: pooled output needs even H, W and an aligned buffer", who);
    IgemmArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = nullptr;
    a.ldx = ldx; a.ldy = ldy; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Cup = Cout;
    a.x_img_stride = (long long)H * W * ldx; a.y_img_stride = (long long)H * W * ldy;
    a.pool = pool; a.ldpool = ldpool;
    return launch_igemm<__bf16, 0, 3>(a, N, (hipStream_t)stream);
}

extern "C" int fea_conv3x3_first_bias_relu(const float* x, const void* w_packed, const float* bias, void* y, int ldy,
                                           int N, int C, int H, int W, int Cout, int dtype, void* stream) {
    const char* who = "fea_conv3x3_first_bias_relu";
    FEA_CHECK(dtype == FEA_BF16, FEA_E_UNSUPPORTED_SHAPE, "%s: bf16 only", who);
    FEA_CHECK(x && w_packed && y && bias, FEA_E_BADARG, "%s: null pointer", who);
    FEA_CHECK(N > 0 && H > 0 && W > 0, FEA_E_BADARG, "%s: bad dims", who);
    FEA_CHECK(C > 0 && 9 * C <= 64, FEA_E_UNSUPPORTED_SHAPE, "%s: C=%d (9*C must be <= 64)", who, C);
    FEA_CHECK(Cout > 0 && Cout % 64 == 0 && ldy >= Cout && ldy % 8 == 0, FEA_E_UNSUPPORTED_SHAPE, "%s: Cout=%d ldy=%d", who, Cout, ldy);
    FEA_CHECK(((uintptr_t)y % 16 == 0) && ((uintptr_t)w_packed % 16 == 0) && ((uintptr_t)x % 4 == 0), FEA_E_BADARG, "%s: alignment", who);
    FEA_CHECK((long long)H * W * ldy < (1ll << 31) && (long long)C * H * W < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "%s: image too large", who);
    {
        int rc = FEA_OK;
        if (fea_first_conv_p(x, w_packed, bias, y, ldy, nullptr, 1, N, C, H, W, Cout, (hipStream_t)stream, &rc)) return rc;
    }
    IgemmArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.stats = nullptr;
    a.ldx = 0; a.ldy = ldy; a.H = H; a.W = W; a.Cin = C; a.Cout = Cout; a.Cup = Cout;
    a.x_img_stride = (long long)C * H * W; a.y_img_stride = (long long)H * W * ldy;
    return launch_igemm<__bf16, 3, 3>(a, N, (hipStream_t)stream);
}

extern "C" int fea_conv1x1(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy, float* stats,
                           int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    int rc = check_common("fea_conv1x1", x, ldx, w, y, ldy, N, H, W, Cin, Cout, dtype);
    if (rc) return rc;
    FEA_CHECK(ldy >= Cout, FEA_E_BADARG, "fea_conv1x1: ldy=%d < Cout=%d", ldy, Cout);
    IgemmArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats;
    a.ldx = ldx; a.ldy = ldy; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Cup = Cout;
    a.x_img_stride = (long long)H * W * ldx; a.y_img_stride = (long long)H * W * ldy;
    return igemm_dispatch<1, 0>(a, N, dtype, stream);
}

extern "C" int fea_upconv2x2_fwd(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                                 int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    int rc = check_common("fea_upconv2x2_fwd", x, ldx, w, y, ldy, N, H, W, Cin, Cout, dtype);
    if (rc) return rc;
    FEA_CHECK(ldy >= Cout, FEA_E_BADARG, "fea_upconv2x2_fwd: ldy=%d < Cout=%d", ldy, Cout);
    if (dtype == FEA_BF16 && fea_upg_fwd(x, ldx, w, bias, y, ldy, N, H, W, Cin, Cout, (hipStream_t)stream, &rc)) return rc;
    IgemmArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = nullptr;
    a.ldx = ldx; a.ldy = ldy; a.H = H; a.W = W; a.Cin = Cin; a.Cout = 4 * Cout; a.Cup = Cout;
    a.x_img_stride = (long long)H * W * ldx; a.y_img_stride = (long long)4 * H * W * ldy;
    return igemm_dispatch<1, 1>(a, N, dtype, stream);
}

extern "C" int fea_upconv2x2_bwd_data(const void* dy, int lddy, const void* w, void* dx, int lddx,
                                      int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    // GEMM: K channels per tap = Cout (of the transposed conv), N = Cin
    int rc = check_common("fea_upconv2x2_bwd_data", dy, lddy, w, dx, lddx, N, H, W, Cout, Cin, dtype);
    if (rc) return rc;
    FEA_CHECK(lddx >= Cin, FEA_E_BADARG, "fea_upconv2x2_bwd_data: lddx=%d < Cin=%d", lddx, Cin);
    FEA_CHECK((long long)4 * H * W * lddy < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE, "fea_upconv2x2_bwd_data: image too large");
    IgemmArgs a{};
    a.x = dy; a.w = w; a.bias = nullptr; a.y = dx; a.stats = nullptr;
    a.ldx = lddy; a.ldy = lddx; a.H = H; a.W = W; a.Cin = Cout; a.Cout = Cin; a.Cup = Cin;
    a.x_img_stride = (long long)4 * H * W * lddy; a.y_img_stride = (long long)H * W * lddx;
    return igemm_dispatch<2, 0>(a, N, dtype, stream);
}
