// Persistent 3x3 convolution for the shallow levels (64 / 128 channels, full-resolution images), bf16, gfx950.
//
// The implicit-GEMM kernel of igemm.hip runs one 256-pixel tile per workgroup; at K = 9 x 64 that is ~4.6 k cycles of
// MFMA between a prologue and an epilogue of the same length, and its 64-column variant re-streams the whole packed
// weight (74 KB) through the vector cache for every tile.  Here ONE workgroup per CU walks a range of pixel tiles and
// keeps the weights in REGISTERS for the kernel's life:
//
//   * 4 waves = 2 (pixel halves of a 8x32 tile) x 2 (32-column halves of a 64-column group); a wave owns 128 pixels x
//     32 output channels: 8 x 2 accumulator tiles of v_mfma_f32_16x16x32_bf16, and ALL weight fragments of its 32
//     columns (K = 576: 144 VGPRs; K = 1152: 288 VGPRs -- the kernel is built for one wave per SIMD, 512 registers);
//   * the MFMA takes the WEIGHT fragment as its A operand and the pixel fragment as B, so an accumulator register holds
//     4 consecutive channels of one pixel: after bf16 packing one v_permlane16_swap per dword gives every lane 8
//     consecutive channels = one 16-byte store, with no LDS staging and no barrier in the epilogue;
//   * pixel (halo) tiles arrive by LDS-DMA into a two-stage ring: the (8+2)x(32+2) halo of the next tile (or next
//     64-channel chunk) is in flight while the current one is multiplied; one barrier per stage.  Tiles are walked
//     down the image (tile row fastest), so consecutive halos share two pixel rows in L2;
//   * BatchNorm partial sums (forward) and the fused BatchNorm-backward reduction (data gradient) are accumulated in
//     registers over ALL tiles of the workgroup and written once per wave at the end (2 rows per workgroup instead of
//     one row per tile): the finalize kernels read ~512 rows instead of 65,536.
//
// GEMM N = 128 (64 -> 128, 128 -> 128) runs as two 64-column groups on two workgroups that start on the same tiles
// (blocks b and b + 8: the same XCD under round-robin placement, so the second reader of a halo finds it in L2).
//
// Reference layers: nn.Conv2d(ci, co, 3, padding=1) of enc1 / enc2 / dec2 / dec1 (models.py:10-11,19,21,28,31), forward and
// data gradient.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

struct C3pArgs {
    const __bf16* x; const __bf16* w; const float* bias; __bf16* y; float* stats;
    long long x_img_stride, y_img_stride;     // elements per image
    int ldx, ldy;
    int H, W;
    int CoutTot;                              // GEMM N of the packed weight
    int tiles_x, tiles_y, ntiles, tiles_per_wg, npixwg, nhalf;
    int wsx, wsn;                             // WALK kernels: the cursor's stride in (x, image)
    // EPI 2
    const __bf16* ybn; int ldybn; long long ybn_img_stride;
    const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
    float* red_part;
    // EPI 3: optional 2x2 max-pool of the activated output, [N][H/2][W/2] pixels ldpool apart
    __bf16* pool; int ldpool; long long pool_img_stride;
    unsigned long long* dbg;                  // C3P_STAMPS builds only: per-workgroup phase cycle sums
};

constexpr int P_TH = 8, P_TW = 32, P_HW2 = P_TW + 2;
constexpr int P_ROWS = 352;                   // >= (8+2)*(32+2) = 340, multiple of 32
constexpr int P_STAGE = P_ROWS * 128;         // bytes of one halo stage
constexpr int P_NDMA = 11;                    // 1 KB LDS-DMA pieces per wave per stage (44 pieces)

// XOR applied to a row's 16-byte chunk index.  A ds_read_b128 is served in groups of 16 lanes = 16 consecutive halo rows, 8 of
// them reading chunk q and 8 chunk q ^ 1 (MI355X_MICROARCH.md, LDS table): bit 0 of the chunk is left alone so the two subsets
// never meet, and 8 consecutive rows get 8 distinct (row parity, XOR) pairs -- conflict-free for every base row and tap.
__device__ __forceinline__ int p_swz(int row) { return ((row >> 1) & 3) << 1; }

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// EPI 0: y = conv + bias (bias nullable); STATS 2: BatchNorm partial sums (sum y, sum y^2), STATS 1: sum y only (slot 1
//        is written as 0: the ConvTranspose2d bias gradient taken from a data-gradient launch), STATS 0: none
// EPI 2: da = conv (no bias) + fused first pass of the next BatchNorm backward (sum g, sum g*xhat)
// EPI 3: inference with BatchNorm folded into weights and bias: relu(conv + bias), plus its MaxPool2d(2) when g.pool is set
// cache-policy switches (A/B): aux operand of the halo LDS-DMA and of the y-row LDS-DMA (2 = nt), nontemporal output stores.
// All three measured SLOWER in the training step (persistent kernels 10.95 -> 11.35 / 10.99 / 11.34 ms, fused-reduction
// launches 2.81 -> 2.77 / 3.11 / 3.03): halos and freshly written rows are re-read through L2.  Defaults stay 0.
#ifndef P_AUX_X
#define P_AUX_X 0
#endif
#ifndef P_AUX_Y
#define P_AUX_Y 0
#endif
#ifndef P_NT_ST
#define P_NT_ST 0
#endif
#ifndef P_PIN2
#define P_PIN2 7
#endif
// A fragments retired per counted wait (2 or 4): the loop is ~97 % issue-bound, a wait is an issue slot
#ifndef P_WAIT_EVERY
#define P_WAIT_EVERY 4
#endif
#ifndef P_OVL2
#define P_OVL2 1        // 0: the K = 1152 layers run item() (epilogue behind the MFMAs, not between them): A/B builds
#endif
#ifndef P_WAIT_EVERY_OVL
#define P_WAIT_EVERY_OVL 4
#endif
__device__ __forceinline__ void st_out(__bf16* p, const u32x4& v) {
    if constexpr (P_NT_ST) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = v;
}
template <int NCH, int EPI, int STATS, bool WALK = false>
__global__ __launch_bounds__(256, 1) void c3p_kernel(const C3pArgs g) {
    // OVL (K = 576): a tile is multiplied as two pixel halves (accumulator tiles m = 0..3, then 4..7, 18 k-steps each);
    // the epilogue of one half is issued, a micro-operation or two at a time, between the MFMA pairs of the other half
    // (of the same tile for the first half, of the next tile for the second), so it costs issue slots but no MFMA
    // idle time and no extra registers.  With K = 1152 both LDS stages hold the current tile: no room to run ahead.
    constexpr bool OVL = NCH == 1 && EPI != 3;
    constexpr int YB = EPI == 2 ? 4 * 8192 : 0;        // per-wave private y tiles (128 px x 64 B)
    // EPI 2 with K = 1152: scale / shift / mean of the 64-column group live in LDS and are read per tile (24 registers that are
    // only live in the epilogue, where the A ring is dead) -- as persistent registers the kernel spilled 30 VGPRs
    constexpr bool COEF_LDS = EPI == 2 && NCH == 2;
    __shared__ __attribute__((aligned(128))) char smem[2 * P_STAGE + YB + 256 + (COEF_LDS ? 768 : 0)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // provably wave-uniform: LDS-DMA bases stay scalar
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, q16 = lane >> 4;
    int pw, half;
    if (g.nhalf > 1 && (g.npixwg & 7) == 0) {        // blocks b, b + 8, ... (one XCD) take the column groups of the same tiles
        half = (blockIdx.x >> 3) % g.nhalf; pw = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * g.nhalf));
    } else { half = blockIdx.x / g.npixwg; pw = blockIdx.x - half * g.npixwg; }
    const int W = g.W;
    const int cbase = half * 64 + wn * 32;              // first GEMM column of this wave

    // ---- weights: all fragments of this wave's 32 columns, for the kernel's life ------------------
    // packed [kc = chunk*9 + tap][col/16][ks][lane][8]
    u32x4 Wf[NCH][9][2][2];
    {
        const __bf16* wb = g.w + (long long)lane * 8;
        const int nt16 = g.CoutTot >> 4;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        Wf[c][tap][ks][n] = *reinterpret_cast<const u32x4*>(
                            wb + ((long long)((c * 9 + tap) * nt16 + (cbase >> 4) + n) * 2 + ks) * 512);
    }

    // The weight fragments are only ever MFMA operands: keep them in the accumulation half of the register file, so that
    // the 256 architectural registers hold the A ring, the addresses and the epilogue's temporaries without copies
    auto pin_weights = [&]() {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    asm volatile("" : "+a"(Wf[0][tap][ks][n]));
                    if constexpr (NCH == 2 && P_PIN2 > 0) if (tap < P_PIN2) asm volatile("" : "+a"(Wf[1][tap][ks][n]));
                }
    };
    pin_weights();

    // ---- A (pixel) fragment addresses: halo rows wm*4 .. wm*4+5, dx 0..2, x = r16 (odd m: +2048 bytes) ----
    uint32_t abase[6][3];
#pragma unroll
    for (int hr = 0; hr < 6; ++hr)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int row = (wm * 4 + hr) * P_HW2 + r16 + dx;
            abase[hr][dx] = row * 128 + ((q16 ^ p_swz(row)) << 4);
        }

    // ---- LDS-DMA descriptors: piece j = 4 i + wave fills LDS rows 8 j .. 8 j + 7 (lane -> row 8j + lane/8, slot lane%8)
    int rel[P_NDMA];
    unsigned fl0 = 0, fl1 = 0;       // 5 flag bits per piece: top, bottom, left, right halo line, padding row
#pragma unroll
    for (int i = 0; i < P_NDMA; ++i) {
        const int row = (4 * i + wave) * 8 + (lane >> 3);
        const int hy = row / P_HW2, hx = row - hy * P_HW2;
        const int chunk = (lane & 7) ^ p_swz(row);
        rel[i] = ((hy - 1) * W + (hx - 1)) * g.ldx * 2 + chunk * 16;
        const unsigned f = (hy == 0 ? 1u : 0u) | (hy == P_TH + 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == P_TW + 1 ? 8u : 0u) |
                           (row >= (P_TH + 2) * P_HW2 ? 16u : 0u);
        if (i < 6) fl0 |= f << (5 * i); else fl1 |= f << (5 * (i - 6));
    }
    // Pieces are buffer loads (LDS-DMA form): the descriptor covers ONE image, a lane's offset is tile origin + rel, and
    // lanes that must read zero padding get an out-of-range offset (the hardware returns 0 for them) -- no select
    // between two 64-bit addresses, 4 VALU per piece.
    const unsigned img_bytes = (unsigned)(g.x_img_stride * 2);
    auto tile_rsrc = [&](int n_img) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(g.x + (long long)n_img * g.x_img_stride), 0, (int)img_bytes, 0x00020000);
    };
    auto tile_off = [&](int tx, int ty, int c) -> int { return ((ty * P_TH) * W + tx * P_TW) * g.ldx * 2 + c * 128; };
    auto tile_edge = [&](int tx, int ty) -> unsigned {
        return (ty == 0 ? 1u : 0u) | (ty == g.tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == g.tiles_x - 1 ? 8u : 0u) | 16u;
    };
    auto issue_piece = [&](__amdgpu_buffer_rsrc_t rs, int toff, unsigned edge, int stage, auto ic) {
        constexpr int i = decltype(ic)::value;
        // bit 5 of `edge` = nothing follows this stage: every lane (interior ones too) takes the out-of-range offset, so the
        // stage that is never consumed costs no memory traffic
        const unsigned f = (((i < 6 ? fl0 >> (5 * i) : fl1 >> (5 * (i - 6))) & edge) & 31u) | (edge >> 5);
        const int voff = f ? (int)0xfffffff0u : toff + rel[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + stage * P_STAGE + (4 * i + wave) * 1024), 16, voff, 0, 0, P_AUX_X);
    };
    auto issue = [&](int n_img, int tx, int ty, int c, int stage) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(n_img);
        const int toff = tile_off(tx, ty, c);
        const unsigned edge = tile_edge(tx, ty);
        sfor<0, P_NDMA>([&](auto ic) { issue_piece(rs, toff, edge, stage, ic); });
    };

    // ---- output addressing: lane's pixel (tile row wm*4 + (m>>1), x = (m&1)*16 + r16), 8 channels from chl -------
    const int chl = (q16 & 1) * 16 + (q16 >> 1) * 8;     // q: 0 -> 0, 1 -> 16, 2 -> 8, 3 -> 24 (after the lane exchange)
    const long long y_lane = ((long long)(wm * 4) * W + r16) * g.ldy + cbase + chl;

    // ---- per-channel state -------------------------------------------------------------------------------
    // bias of the 64-column group in LDS (after the stages): read back as the C operand of a tile's first MFMAs, so
    // it occupies registers only for the first k-step of each tile
    float* const blds = reinterpret_cast<float*>(smem + 2 * P_STAGE + YB);
    if (tid < 64) blds[tid] = ((EPI == 0 || EPI == 3) && g.bias) ? g.bias[half * 64 + tid] : 0.f;
    float* const clds = blds + 64;                // [3][64]: scale, shift, mean (COEF_LDS)
    if constexpr (COEF_LDS) {
        if (tid < 192) {
            const float* src = tid < 64 ? g.bn_scale : (tid < 128 ? g.bn_shift : g.bn_mean);
            clds[tid] = src[half * 64 + (tid & 63)];
        }
    }
    __syncthreads();
    f32x4 S[2], Q[2];                 // EPI 0: sum y, sum y^2 of channels cbase + 16 n + 4 q + e
    float a0[8], a1[8], csc[8], csh[8], cmn[8];   // EPI 2: channels cbase + chl + j
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        S[n] = f32x4{0.f, 0.f, 0.f, 0.f}; Q[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a0[j] = 0.f; a1[j] = 0.f;
        csc[j] = (EPI == 2 && !COEF_LDS) ? g.bn_scale[cbase + chl + j] : 0.f;
        csh[j] = (EPI == 2 && !COEF_LDS) ? g.bn_shift[cbase + chl + j] : 0.f;
        cmn[j] = (EPI == 2 && !COEF_LDS) ? g.bn_mean[cbase + chl + j] : 0.f;
    }
    // EPI 2: the y rows arrive by LDS-DMA buffer loads (descriptor = one image): a lane's byte offset inside the tile is a
    // constant, the tile origin and the row of piece m are wave-uniform (scalar offset operand)
    int ybn_lane = 0;
    char* const ylds = smem + 2 * P_STAGE + wave * 8192;
    if constexpr (EPI == 2) ybn_lane = (((wm * 4) * W + (lane >> 2)) * g.ldybn + cbase + (lane & 3) * 8) * 2;
    auto y_rsrc = [&](int n_img) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(g.ybn + (long long)n_img * g.ybn_img_stride), 0, (int)(g.ybn_img_stride * 2), 0x00020000);
    };
    auto y_piece = [&](__amdgpu_buffer_rsrc_t rs, int tile_byte_off, int m) {
        const int soff = tile_byte_off + (((m >> 1) * W + (m & 1) * 16) * g.ldybn) * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(ylds + m * 1024), 16, ybn_lane, soff, 0, P_AUX_Y);
    };

    f32x4 acc[8][2];
    // the first k-step of a tile takes the bias as its C operand: the accumulators are never re-initialised
    const uint32_t bias_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 2 * P_STAGE + YB + (wn * 32 + 4 * q16) * 4;

    // one stage: 9 taps x 2 k-steps of 32.  The A (pixel) fragment of tile m is re-read for the NEXT k-step right after
    // its two MFMAs of this k-step, so 7 reads are always in flight behind the one an MFMA needs: the LDS reads are
    // inline asm in a fixed order (the LDS returns them in order) retired by a constant counted wait, lgkmcnt(7).
    // `between(step)` is called once per k-step behind its MFMAs (the LDS-DMA pieces of the next stage go there).
    auto lds_rd = [&](u32x4& dst, uint32_t addr, auto offc) {
#ifdef C3P_EXP_HALF_READS      // EXPERIMENT (wrong results): every second fragment read dropped.  17-28 % faster -- but NOT what a wave tile with
        // half the reads per MFMA gains: that was built (64 pixels x 64 channels per wave, 288 weight registers, each fragment feeding
        // four MFMAs) and measured 1-3 % (and spilled with the BatchNorm sums), and dropping every fragment WAIT instead
        // (C3P_EXP_NO_WAITS) changes nothing.  The MFMAs of the dropped fragments multiply the same stale registers every time:
        // operands that do not toggle draw less power, and the chip, clock-limited by its power budget under this load, runs faster.
        if constexpr ((decltype(offc)::value & 2048) != 0) return;
#endif
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(offc)::value) : "memory");
    };
    auto mma_stage = [&](auto cc, auto sc_, auto&& hook) {
        constexpr int C = decltype(cc)::value, STG = decltype(sc_)::value;
        const uint32_t st = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        u32x4 A[8], bq[2];
        if constexpr (C == 0) {      // two more reads in front of the ring: retired by the first counted wait
            lds_rd(bq[0], bias_addr, std::integral_constant<int, 0>{});
            lds_rd(bq[1], bias_addr, std::integral_constant<int, 64>{});
        }
        sfor<0, 8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            lds_rd(A[m], st + abase[m >> 1][0], std::integral_constant<int, STG * P_STAGE + (m & 1) * 2048>{});
        });
        f32x4 biasv[2];
        if constexpr (C == 0) {
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bq[0]), "+v"(bq[1]));
            biasv[0] = __builtin_bit_cast(f32x4, bq[0]); biasv[1] = __builtin_bit_cast(f32x4, bq[1]);
        }
        sfor<0, 18>([&](auto stc) {
            constexpr int step = decltype(stc)::value, tap = step >> 1, ks = step & 1;
            constexpr int nstep = step + 1, ntap = nstep >> 1, nks = nstep & 1, ndy = ntap / 3, ndx = ntap % 3;
            sfor<0, 8>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                // one wait per two MFMA pairs (fragments m and m + 1: 6 reads in flight behind them); no register operand,
                // so the compiler pads no hazard state behind the statement -- the sched_barrier keeps the MFMAs below it
                if constexpr ((m & (P_WAIT_EVERY - 1)) == 0) {
                    // (P_WAIT_EVERY fragments per wait: 8 reads are in flight, the P_WAIT_EVERY oldest are the ones needed; last k-step: 8 - m)
                    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(nstep < 18 ? 8 - P_WAIT_EVERY : (8 - m - P_WAIT_EVERY > 0 ? 8 - m - P_WAIT_EVERY : 0)) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[C][tap][ks][n]),
                                                                        __builtin_bit_cast(bf16x8, A[m]),
                                                                        (C == 0 && step == 0) ? biasv[n] : acc[m][n], 0, 0, 0);
                if constexpr (nstep < 18)
                    lds_rd(A[m], st + (abase[(m >> 1) + ndy][ndx] ^ (uint32_t)(nks << 6)),
                           std::integral_constant<int, STG * P_STAGE + (m & 1) * 2048>{});
                hook(stc, mc);
            });
        });
    };

    // ---- tile walk ----------------------------------------------------------------------------------------
    // WALK false: workgroup pw takes tiles pw * tiles_per_wg ..., consecutive (vertical neighbours one after the other: the two shared
    // halo rows are L2 hits).  WALK true (64 -> 64 layers, which are HBM-bound; needs npixwg % tiles_y == 0): tiles pw, pw + npixwg,
    // ... -- at any moment the chip writes one compact window of npixwg tiles instead of npixwg windows 8 MB apart.  A store-only
    // probe of this tile shape writes 10 % faster that way (scripts/hbm_tile_probe.hip); the 64 -> 64 launches @512^2 went
    // 1.07 -> 0.96 ms (forward) and 1.42 -> 1.30 ms (data gradient with the fused reduction), the 64 -> 128 ones did not move.
    // The cursor keeps its tile row and advances by (wsx, wsn) in (x, image); t counts the workgroup's own tiles.
    const int first = WALK ? pw : pw * g.tiles_per_wg;
    const int t0 = WALK ? 0 : first;
    const int t1 = WALK ? (first < g.ntiles ? (g.ntiles - first + g.npixwg - 1) / g.npixwg : 0) : min(first + g.tiles_per_wg, g.ntiles);
    int ty = first % g.tiles_y, tx = (first / g.tiles_y) % g.tiles_x, ni = first / (g.tiles_y * g.tiles_x);
    auto advance = [&](int& n_, int& x_, int& y_) {
        if constexpr (WALK) { x_ += g.wsx; if (x_ >= g.tiles_x) { x_ -= g.tiles_x; ++n_; } n_ += g.wsn; }
        else if (++y_ == g.tiles_y) { y_ = 0; if (++x_ == g.tiles_x) { x_ = 0; ++n_; } }
    };
    if (t0 < t1) issue(ni, tx, ty, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#ifdef C3P_STAMPS
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define STAMP(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
    // work item = (tile, chunk); stage = item & 1
    auto item = [&](int t, auto cc, auto sc_) {
        constexpr int C = decltype(cc)::value, STG = decltype(sc_)::value;
        STAMP(4);
        pin_weights();
        // this wave's DMA pieces of this stage have landed (the vmcnt(0) behind the previous stage's MFMAs / the
        // prologue); after the barrier every wave's have, and every wave is done reading the other stage
        __builtin_amdgcn_s_barrier();
        STAMP(0);
        // next item into the other stage: its 11 pieces are issued behind every second MFMA pair of the first three
        // k-steps (mma_stage), early enough to land during this stage and spread so that their address arithmetic
        // issues in the shadow of the MFMAs
        int n2 = ni, x2 = tx, y2 = ty;
        if constexpr (C + 1 == NCH) advance(n2, x2, y2);
        const bool more = (C + 1 < NCH) || (t + 1 < t1);
        const __amdgpu_buffer_rsrc_t nrs = tile_rsrc(more ? n2 : ni);
        const int ntoff = tile_off(x2, y2, (C + 1) % NCH);
        const unsigned nedge = more ? tile_edge(x2, y2) : 63u;          // nothing follows: no lane reads memory (issue_piece), the stage is zero-filled
        if constexpr (EPI == 2 && C == NCH - 1) {
            // this tile's pre-BN activations, the 64-byte rows this wave will read itself: 16 pixels x 64 B per piece
            const __amdgpu_buffer_rsrc_t yrs = y_rsrc(ni);
            const int ytile = (((ty * P_TH) * W + tx * P_TW) * g.ldybn) * 2;
#pragma unroll
            for (int m = 0; m < 8; ++m) y_piece(yrs, ytile, m);
        }
        mma_stage(cc, sc_, [&](auto stc, auto mc) {
            constexpr int step = decltype(stc)::value, m = decltype(mc)::value, piece = step * 4 + (m >> 1);
            if constexpr ((m & 1) == 1 && piece < P_NDMA) issue_piece(nrs, ntoff, nedge, STG ^ 1, std::integral_constant<int, piece>{});
        });
        // the next stage's pieces (and this tile's y pieces) were issued a whole MFMA phase ago, the previous tile's
        // stores before that: this wait is (nearly) free and the epilogue's stores below are never waited for
        STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(2);
        if constexpr (C == NCH - 1) {
            __bf16* yt = g.y + (long long)ni * g.y_img_stride + ((long long)(ty * P_TH) * W + tx * P_TW) * g.ldy + y_lane;
            if constexpr (EPI == 3) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[m][n][e] = fmaxf(acc[m][n][e], 0.f);
                if (g.pool) {
                    // window rows = accumulator tiles m and m + 2 of this lane, window columns = lanes r16 and r16 ^ 1;
                    // the even lane of a pair stores the pooled pixel (tile row wm*2 + (m>>2), x = (m&1)*8 + r16/2)
                    __bf16* pt = g.pool + (long long)ni * g.pool_img_stride +
                                 ((long long)(ty * (P_TH / 2) + wm * 2) * (W / 2) + tx * (P_TW / 2) + (r16 >> 1)) * g.ldpool + cbase + chl;
#pragma unroll
                    for (int mp = 0; mp < 4; ++mp) {
                        const int m = (mp & 1) + (mp >> 1) * 4;            // 0, 1, 4, 5
                        uint32_t pk[2][2];
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float a = fmaxf(acc[m][n][e], acc[m + 2][n][e]);
                                const float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xf, 0xf, false));
                                v[e] = fmaxf(a, b);
                            }
                            pk[n][0] = pack_bf16x2(v[0], v[1]); pk[n][1] = pack_bf16x2(v[2], v[3]);
                        }
                        const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                        const u32x4 po = {r0[0], r1[0], r0[1], r1[1]};
                        if ((r16 & 1) == 0)
                            *reinterpret_cast<u32x4*>(pt + ((long long)(mp >> 1) * (W / 2) + (mp & 1) * 8) * g.ldpool) = po;
                    }
                }
            }
            float lsc[8], lsh[8], lmn[8];
            if constexpr (COEF_LDS) {
                const float* cf = clds + wn * 32 + chl;
#pragma unroll
                for (int j = 0; j < 8; ++j) { lsc[j] = cf[j]; lsh[j] = cf[64 + j]; lmn[j] = cf[128 + j]; }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if constexpr (EPI == 0 && STATS >= 1) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {          // scalar on purpose: packed f32 VALU is slower beside MFMAs
                            S[n][e] += acc[m][n][e];
                            if constexpr (STATS == 2) Q[n][e] = __builtin_fmaf(acc[m][n][e], acc[m][n][e], Q[n][e]);
                        }
                    }
                }
                uint32_t lo0 = pack_bf16x2(acc[m][0][0], acc[m][0][1]), hi0 = pack_bf16x2(acc[m][0][2], acc[m][0][3]);
                uint32_t lo1 = pack_bf16x2(acc[m][1][0], acc[m][1][1]), hi1 = pack_bf16x2(acc[m][1][2], acc[m][1][3]);
                // odd 16-lane rows of the first operand <-> even rows of the second: every lane ends up with 8 consecutive channels
                const auto r0 = __builtin_amdgcn_permlane16_swap(lo0, lo1, false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(hi0, hi1, false, false);
                const u32x4 out = {r0[0], r1[0], r0[1], r1[1]};
                st_out(yt + ((long long)(m >> 1) * W + (m & 1) * 16) * g.ldy, out);
                if constexpr (EPI == 2) {
                    // lane (pixel r16 of tile m, channels chl..chl+7): y row at ylds + m*1024 + r16*64 + piece*16
                    const u32x4 yv = *reinterpret_cast<const u32x4*>(ylds + m * 1024 + r16 * 64 + (chl >> 3) * 16);
                    const bf16x8 yy = __builtin_bit_cast(bf16x8, yv), dd = __builtin_bit_cast(bf16x8, out);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float yf = (float)yy[j];
                        const float sc_ = COEF_LDS ? lsc[j] : csc[j], sh_ = COEF_LDS ? lsh[j] : csh[j], mn_ = COEF_LDS ? lmn[j] : cmn[j];
                        const float gg = __builtin_fmaf(yf, sc_, sh_) > 0.f ? (float)dd[j] : 0.f;
                        a0[j] += gg; a1[j] = __builtin_fmaf(gg, yf - mn_, a1[j]);
                    }
                }
            }
            advance(ni, tx, ty);
            STAMP(3);
        }
    };

    // ---- overlapped form (OVL) ----------------------------------------------------------------------------------
    // one pixel half (accumulator tiles 4 HALF .. 4 HALF + 3) over the 18 k-steps of a stage.  A fragments: ring of
    // 8 = 2 k-steps x 4 tiles, re-read for k-step s + 2 right after the MFMA pair of k-step s: 7 reads in flight
    // behind the one a pair needs.  The ring never drains: k-steps 16 and 17 refill it with k-steps 0 and 1 of the NEXT half
    // (second half of this tile, or first half of the next tile in the other stage), behind that half's two bias reads,
    // so a half starts with its operands already on their way.  hook(slot) runs behind MFMA pair slot = 4 * step + mm
    // (72 slots); before16() runs in front of k-step 16 (the second half puts the tile's barrier there).
    u32x4 Ar[8], bqr[2];
    auto prime_reads = [&](auto sc_, auto hc, auto stepc, auto mmc) {     // read of (stage, half, k-step 0 / 1, tile mm) into its ring slot
        constexpr int STG = decltype(sc_)::value, HALF = decltype(hc)::value, step = decltype(stepc)::value, mm = decltype(mmc)::value;
        constexpr int m = HALF * 4 + mm, ks = step & 1;
        const uint32_t st = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        lds_rd(Ar[step * 4 + mm], st + (abase[m >> 1][0] ^ (uint32_t)(ks << 6)), std::integral_constant<int, STG * P_STAGE + (m & 1) * 2048>{});
    };
    auto prime_bias = [&]() {
        lds_rd(bqr[0], bias_addr, std::integral_constant<int, 0>{});
        lds_rd(bqr[1], bias_addr, std::integral_constant<int, 64>{});
    };
    auto mma_half = [&](auto sc_, auto hc, auto&& hook, auto&& before16) {
        constexpr int STG = decltype(sc_)::value, HALF = decltype(hc)::value;
        constexpr int NSTG = HALF ? STG ^ 1 : STG, NHALF = HALF ^ 1;          // the half whose first two k-steps this one reads ahead
        const uint32_t st = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        // on entry: 2 bias reads, then the 8 reads of k-steps 0 and 1 are in flight (prime_* of the previous half or of the first tile)
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bqr[0]), "+v"(bqr[1]));
        const f32x4 biasv[2] = {__builtin_bit_cast(f32x4, bqr[0]), __builtin_bit_cast(f32x4, bqr[1])};
        sfor<0, 18>([&](auto stc) {
            constexpr int step = decltype(stc)::value, tap = step >> 1, ks = step & 1;
            constexpr int nstep = step + 2, ntap = nstep >> 1, nks = nstep & 1, ndy = ntap / 3, ndx = ntap % 3;
            if constexpr (step == 16) { before16(); prime_bias(); }
            sfor<0, 4>([&](auto mc) {
                constexpr int mm = decltype(mc)::value, m = HALF * 4 + mm, ai = (step & 1) * 4 + mm;
                if constexpr ((mm & (P_WAIT_EVERY_OVL - 1)) == 0) {
                    // fragments mm .. mm + P_WAIT_EVERY_OVL - 1: 8 ring reads are in flight (10 in k-steps 16 / 17, where the next half's two
                    // bias reads sit between them), the oldest ones are needed (never fewer in flight: no tail)
#ifndef C3P_EXP_NO_WAITS       // EXPERIMENT (racy): no fragment waits = what LDS latency beyond the ring's reach costs (measured: nothing)
                    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"((step >= 16 ? 10 : 8) - P_WAIT_EVERY_OVL) : "memory");
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[0][tap][ks][n]),
                                                                        __builtin_bit_cast(bf16x8, Ar[ai]),
                                                                        step == 0 ? biasv[n] : acc[m][n], 0, 0, 0);
                if constexpr (nstep < 18)
                    lds_rd(Ar[ai], st + (abase[(m >> 1) + ndy][ndx] ^ (uint32_t)(nks << 6)),
                           std::integral_constant<int, STG * P_STAGE + (m & 1) * 2048>{});
                else
                    prime_reads(std::integral_constant<int, NSTG>{}, std::integral_constant<int, NHALF>{},
                                std::integral_constant<int, nstep - 18>{}, mc);
                hook(std::integral_constant<int, step * 4 + mm>{});
            });
        });
    };
    // micro-operations of the epilogue of accumulator tile m (k = 0 .. NPER - 1)
    constexpr int NPER = EPI == 2 ? 16 : 8;
    uint32_t pk[4]; u32x4 outv; bf16x8 yyv;            // live between the micro-operations of one m
    auto epi_micro = [&](__bf16* yt, auto mcc, auto uc) {
        constexpr int m = decltype(mcc)::value, u = decltype(uc)::value;
        constexpr int UC = EPI == 2 ? 0 : 4;          // first of the four convert / exchange / store micro-operations
        if constexpr (EPI == 0 && u < 4) {
            constexpr int n = u >> 1;
            // scalar on purpose: packed f32 VALU is slower beside MFMAs (the file is built without the SLP vectoriser)
            if constexpr ((u & 1) == 0 && STATS >= 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) S[n][e] += acc[m][n][e];
            }
            if constexpr ((u & 1) == 1 && STATS == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Q[n][e] = __builtin_fmaf(acc[m][n][e], acc[m][n][e], Q[n][e]);
            }
        }
        if constexpr (u == UC)     { pk[0] = pack_bf16x2(acc[m][0][0], acc[m][0][1]); pk[1] = pack_bf16x2(acc[m][0][2], acc[m][0][3]); }
        if constexpr (u == UC + 1) { pk[2] = pack_bf16x2(acc[m][1][0], acc[m][1][1]); pk[3] = pack_bf16x2(acc[m][1][2], acc[m][1][3]); }
        if constexpr (u == UC + 2) {
            const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0], pk[2], false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(pk[1], pk[3], false, false);
            outv = u32x4{r0[0], r1[0], r0[1], r1[1]};
        }
        if constexpr (u == UC + 3) st_out(yt + ((long long)(m >> 1) * W + (m & 1) * 16) * g.ldy, outv);
        if constexpr (EPI == 2) {
            if constexpr (u == 4) yyv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ylds + m * 1024 + r16 * 64 + (chl >> 3) * 16));
            if constexpr (u >= 5 && u < 13) {
                constexpr int j = u - 5;
                const bf16x8 dd = __builtin_bit_cast(bf16x8, outv);
                const float yf = (float)yyv[j];
                const float gg = __builtin_fmaf(yf, csc[j], csh[j]) > 0.f ? (float)dd[j] : 0.f;
                a0[j] += gg; a1[j] = __builtin_fmaf(gg, yf - cmn[j], a1[j]);
            }
        }
    };
    // the 4 * NPER micro-operations of one half, spread over slots S0 .. 71 of the other half's MFMA pairs
    auto epi_slot = [&](__bf16* yt, auto hc, auto s0c, auto slotc) {
        constexpr int HALF = decltype(hc)::value, S0 = decltype(s0c)::value, slot = decltype(slotc)::value, NM = 4 * NPER;
        constexpr int lo = slot <= S0 ? -1 : ((slot - S0) * NM) / (72 - S0) - 1;
        constexpr int hi = slot < S0 ? -1 : ((slot - S0 + 1) * NM) / (72 - S0) - 1;
        sfor<0, hi - lo>([&](auto dc) {
            constexpr int k = lo + 1 + decltype(dc)::value;
            epi_micro(yt, std::integral_constant<int, HALF * 4 + k / NPER>{}, std::integral_constant<int, k % NPER>{});
        });
    };
    auto y_pieces = [&](auto hc) {      // EPI 2: this tile's pre-BN activations of one pixel half, the rows this wave reads itself
        constexpr int HALF = decltype(hc)::value;
        const __amdgpu_buffer_rsrc_t yrs = y_rsrc(ni);
        const int ytile = (((ty * P_TH) * W + tx * P_TW) * g.ldybn) * 2;
#pragma unroll
        for (int m = HALF * 4; m < HALF * 4 + 4; ++m) y_piece(yrs, ytile, m);
    };
    // stores among the first-half epilogue's micro-operations that are issued before slot `lim` of the second half
    constexpr int S0B = 8, NMB = 4 * NPER;
    auto stores_before = [](int lim) constexpr {
        int n = 0;
        for (int slot = S0B; slot < lim; ++slot) {
            const int lo = slot <= S0B ? -1 : ((slot - S0B) * NMB) / (72 - S0B) - 1, hi = ((slot - S0B + 1) * NMB) / (72 - S0B) - 1;
            for (int k = lo + 1; k <= hi; ++k) n += (k % NPER) == (EPI == 2 ? 3 : 7) ? 1 : 0;
        }
        return n;
    };
    constexpr int NSTB = stores_before(64);          // k-step 16 = slot 64
    __bf16* prev_yt = nullptr;
    auto item_ovl = [&](int t, auto sc_, auto fc) {
        constexpr int STG = decltype(sc_)::value;
        constexpr bool FIRST = decltype(fc)::value;       // no previous tile: no second-half epilogue to interleave
        pin_weights();
        if constexpr (FIRST) {
            // the first tile's pieces were waited for in the prologue; every later tile's barrier sits at k-step 16 of the
            // tile before it (below), and its ring is primed from there
            __builtin_amdgcn_s_barrier();
            prime_bias();
            sfor<0, 8>([&](auto ic) {
                prime_reads(sc_, std::integral_constant<int, 0>{}, std::integral_constant<int, (decltype(ic)::value >> 2)>{},
                            std::integral_constant<int, (decltype(ic)::value & 3)>{});
            });
        }
        int n2 = ni, x2 = tx, y2 = ty;
        advance(n2, x2, y2);
        const bool more = t + 1 < t1;
        const __amdgpu_buffer_rsrc_t nrs = tile_rsrc(more ? n2 : ni);
        const int ntoff = tile_off(x2, y2, 0);
        const unsigned nedge = more ? tile_edge(x2, y2) : 63u;          // nothing follows: no lane reads memory; zeros land (the ring still reads that stage ahead)
        __bf16* const yt = g.y + (long long)ni * g.y_img_stride + ((long long)(ty * P_TH) * W + tx * P_TW) * g.ldy + y_lane;
        if constexpr (EPI == 2) y_pieces(std::integral_constant<int, 0>{});
        // first half: the next tile's 11 DMA pieces behind slots 1, 3, .. 21; from slot 24 on the previous tile's second-half epilogue
        mma_half(sc_, std::integral_constant<int, 0>{}, [&](auto slotc) {
            constexpr int slot = decltype(slotc)::value;
            if constexpr ((slot & 1) == 1 && (slot >> 1) < P_NDMA) issue_piece(nrs, ntoff, nedge, STG ^ 1, std::integral_constant<int, (slot >> 1)>{});
            if constexpr (!FIRST) epi_slot(prev_yt, std::integral_constant<int, 1>{}, std::integral_constant<int, 24>{}, slotc);
        }, [] {});
        if constexpr (EPI == 2) {
            // VM operations of this tile so far, in issue order: 4 y pieces, 11 DMA pieces, then (not FIRST) 4 stores
            if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else                 asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            y_pieces(std::integral_constant<int, 1>{});          // second-half rows: last read by the epilogue just issued
        }
        // second half: this tile's first-half epilogue from slot 8 on.  In front of k-step 16 the tile's barrier: every
        // VM operation but the NSTB stores issued since (so the next stage's pieces and the second-half y rows) has landed,
        // and every LDS read of this stage has completed (the last were issued two k-steps ago) -- after the barrier the
        // other stage may be read (ring refill of k-steps 16 / 17) and this one overwritten (the next tile's DMA pieces).
        mma_half(sc_, std::integral_constant<int, 1>{}, [&](auto slotc) {
            epi_slot(yt, std::integral_constant<int, 0>{}, std::integral_constant<int, S0B>{}, slotc);
        }, [&] {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NSTB) : "memory");
            __builtin_amdgcn_s_barrier();
        });
        prev_yt = yt;
        advance(ni, tx, ty);
    };

    // ---- overlapped form for K = 1152 (OVL2: forward and plain data gradient of the 128-input-channel layers) ----------------
    // Phases of a tile: (chunk 0, half A) (chunk 0, half B) (chunk 1, half A) (chunk 1, half B), chunk c in stage c.  Both halves'
    // accumulators stay live through the tile as before; what moves is the epilogue: half A's runs between the MFMA pairs of the
    // fourth phase, half B's between those of the NEXT tile's first phase (item() left the matrix pipe idle for ~1,660 of a tile's
    // ~12,900 cycles, scripts/c3p_stamps.py).  LDS-DMA: this tile's chunk 1 goes to stage 1 during the first phase (stage 1 was last
    // read before the previous tile's barrier), the next tile's chunk 0 to stage 0 during the third (stage 0 was last read before
    // this tile's first barrier); the two barriers sit in front of k-step 16 of the second and the fourth phase, where the ring
    // starts reading the other stage.  The bias is the C operand of both halves' first k-step of chunk 0.
    constexpr bool OVL2 = P_OVL2 && NCH == 2 && EPI == 0;
    auto mma_phase = [&](auto pc, auto&& hook, auto&& before16) {
        constexpr int P = decltype(pc)::value, C = P >> 1, HALF = P & 1, STG = C;
        constexpr int NP = (P + 1) & 3, NSTG = NP >> 1, NHALF = NP & 1;
        constexpr bool BIAS = C == 0, NBIAS = NSTG == 0;
        constexpr int CW = NCH == 2 ? C : 0;
        const uint32_t st = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        f32x4 biasv[2];
        if constexpr (BIAS) {
            // on entry: 2 bias reads, then the 8 reads of k-steps 0 and 1 are in flight
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bqr[0]), "+v"(bqr[1]));
            biasv[0] = __builtin_bit_cast(f32x4, bqr[0]); biasv[1] = __builtin_bit_cast(f32x4, bqr[1]);
        }
        sfor<0, 18>([&](auto stc) {
            constexpr int step = decltype(stc)::value, tap = step >> 1, ks = step & 1;
            constexpr int nstep = step + 2, ntap = nstep >> 1, nks = nstep & 1, ndy = ntap / 3, ndx = ntap % 3;
            if constexpr (step == 16) { before16(); if constexpr (NBIAS) prime_bias(); }
            sfor<0, 4>([&](auto mc) {
                constexpr int mm = decltype(mc)::value, m = HALF * 4 + mm, ai = (step & 1) * 4 + mm;
                if constexpr ((mm & (P_WAIT_EVERY_OVL - 1)) == 0) {
                    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"((step >= 16 && NBIAS ? 10 : 8) - P_WAIT_EVERY_OVL) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Wf[CW][tap][ks][n]),
                                                                        __builtin_bit_cast(bf16x8, Ar[ai]),
                                                                        (BIAS && step == 0) ? biasv[n] : acc[m][n], 0, 0, 0);
                if constexpr (nstep < 18)
                    lds_rd(Ar[ai], st + (abase[(m >> 1) + ndy][ndx] ^ (uint32_t)(nks << 6)),
                           std::integral_constant<int, STG * P_STAGE + (m & 1) * 2048>{});
                else
                    prime_reads(std::integral_constant<int, NSTG>{}, std::integral_constant<int, NHALF>{},
                                std::integral_constant<int, nstep - 18>{}, mc);
                hook(std::integral_constant<int, step * 4 + mm>{});
            });
        });
    };
    auto item_ovl2 = [&](int t, auto fc) {
        constexpr bool FIRST = decltype(fc)::value;       // no previous tile: no second-half epilogue to interleave
        pin_weights();
        if constexpr (FIRST) {
            __builtin_amdgcn_s_barrier();                 // chunk 0 of the first tile (prologue) has landed for every wave
            prime_bias();
            sfor<0, 8>([&](auto ic) {
                prime_reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, (decltype(ic)::value >> 2)>{},
                            std::integral_constant<int, (decltype(ic)::value & 3)>{});
            });
        }
        int n2 = ni, x2 = tx, y2 = ty;
        advance(n2, x2, y2);
        const bool more = t + 1 < t1;
        const __amdgpu_buffer_rsrc_t crs = tile_rsrc(ni);
        const int ctoff = tile_off(tx, ty, 1);
        const unsigned cedge = tile_edge(tx, ty);
        const __amdgpu_buffer_rsrc_t nrs = tile_rsrc(more ? n2 : ni);
        const int ntoff = tile_off(x2, y2, 0);
        const unsigned nedge = more ? tile_edge(x2, y2) : 63u;          // nothing follows: zeros land, no lane reads memory
        __bf16* const yt = g.y + (long long)ni * g.y_img_stride + ((long long)(ty * P_TH) * W + tx * P_TW) * g.ldy + y_lane;
        // (chunk 0, A): this tile's chunk 1 -> stage 1 behind slots 1, 3, .. 21; from slot 24 on the previous tile's half-B epilogue
        mma_phase(std::integral_constant<int, 0>{}, [&](auto slotc) {
            constexpr int slot = decltype(slotc)::value;
            if constexpr ((slot & 1) == 1 && (slot >> 1) < P_NDMA) issue_piece(crs, ctoff, cedge, 1, std::integral_constant<int, (slot >> 1)>{});
            if constexpr (!FIRST) epi_slot(prev_yt, std::integral_constant<int, 1>{}, std::integral_constant<int, 24>{}, slotc);
        }, [] {});
        // (chunk 0, B): in front of k-step 16 the pieces of chunk 1 have landed (behind them only the 4 stores of that epilogue) and
        // every read of stage 0 has completed; after the barrier stage 1 is read and stage 0 may be overwritten
        mma_phase(std::integral_constant<int, 1>{}, [](auto) {}, [&] {
            if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else                 asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        });
        // (chunk 1, A): the next tile's chunk 0 -> stage 0
        mma_phase(std::integral_constant<int, 2>{}, [&](auto slotc) {
            constexpr int slot = decltype(slotc)::value;
            if constexpr ((slot & 1) == 1 && (slot >> 1) < P_NDMA) issue_piece(nrs, ntoff, nedge, 0, std::integral_constant<int, (slot >> 1)>{});
        }, [] {});
        // (chunk 1, B): this tile's half-A epilogue from slot 8 on; the tile's second barrier in front of k-step 16 (behind the
        // next tile's pieces only the NSTB stores issued since)
        mma_phase(std::integral_constant<int, 3>{}, [&](auto slotc) {
            epi_slot(yt, std::integral_constant<int, 0>{}, std::integral_constant<int, S0B>{}, slotc);
        }, [&] {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NSTB) : "memory");
            __builtin_amdgcn_s_barrier();
        });
        prev_yt = yt;
        advance(ni, tx, ty);
    };

    if constexpr (OVL2) {
        if (t0 < t1) {
            item_ovl2(t0, std::true_type{});
            for (int t = t0 + 1; t < t1; ++t) item_ovl2(t, std::false_type{});
            sfor<0, 4 * NPER>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                epi_micro(prev_yt, std::integral_constant<int, 4 + k / NPER>{}, std::integral_constant<int, k % NPER>{});
            });
        }
    } else
    if constexpr (OVL) {
        if (t0 < t1) {
            item_ovl(t0, std::integral_constant<int, 0>{}, std::true_type{});
            int t = t0 + 1;
            while (t < t1) {
                item_ovl(t, std::integral_constant<int, 1>{}, std::false_type{});
                if (++t >= t1) break;
                item_ovl(t, std::integral_constant<int, 0>{}, std::false_type{});
                ++t;
            }
            // the last tile's second-half epilogue
            sfor<0, 4 * NPER>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                epi_micro(prev_yt, std::integral_constant<int, 4 + k / NPER>{}, std::integral_constant<int, k % NPER>{});
            });
        }
    } else
    if constexpr (NCH == 1) {
        for (int t = t0; t < t1; t += 2) {
            item(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            if (t + 1 < t1) item(t + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            else break;
        }
    } else {
        for (int t = t0; t < t1; ++t) {
            item(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            item(t, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        }
    }

#ifdef C3P_STAMPS
    if (lane == 0 && g.dbg) for (int k = 0; k < 5; ++k) g.dbg[(blockIdx.x * 4 + wave) * 5 + k] = ph[k];
#endif
    // ---- per-wave partial sums: row 2 pw + wm, this wave's 32 columns ----------------------------------------
    const int row = pw * 2 + wm;
    if constexpr (EPI == 0) {
        if constexpr (STATS >= 1) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = S[n][e], q = Q[n][e];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
                    if (r16 == 0) {
                        float* dst = g.stats + ((long long)row * g.CoutTot + cbase + 16 * n + 4 * q16 + e) * 2;
                        dst[0] = s; dst[1] = q;
                    }
                }
        }
    } else if constexpr (EPI == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s0 = a0[j], s1 = a1[j];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
            if (r16 == 0) {
                const int ch = cbase + chl + j;
                // sum g*xhat = invstd * sum g*(y - mean): centred per element, so |mean| >> std of y does not cancel
                float* dst = g.red_part + ((long long)row * g.CoutTot + ch) * 2;
                dst[0] = s0; dst[1] = g.bn_invstd[ch] * s1;
            }
        }
    }
}

// ---------------------------------------------------------------------------
struct C3pPlan { int tiles_x, tiles_y, ntiles, nhalf, npixwg, tiles_per_wg; };

static bool c3p_plan(int N, int H, int W, int Cin, int Cout, int dtype, C3pPlan& p) {
    if (dtype != FEA_BF16 || (Cin != 64 && Cin != 128) || (Cout != 64 && Cout != 128 && Cout != 256)) return false;
    if (H % P_TH || W % P_TW || N <= 0) return false;
    p.tiles_x = W / P_TW; p.tiles_y = H / P_TH;
    const long long nt = (long long)N * p.tiles_x * p.tiles_y;
    if (nt >= (1ll << 30)) return false;
    p.ntiles = (int)nt;
    p.nhalf = Cout / 64;
    int want = 256 / p.nhalf * fea_wg_split();      // one workgroup per CU (x fea_set_persistent_split when CUs may be taken away)
    if (want > p.ntiles) want = p.ntiles;
    p.tiles_per_wg = (p.ntiles + want - 1) / want;
    p.npixwg = (p.ntiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    return true;
}

// rows of the partial-sum tables (stats / part) the persistent kernels write; 0 = shape not served by them
extern "C" int fea_conv3x3p_rows(int N, int H, int W, int Cin, int Cout, int dtype) {
    C3pPlan p;
    if (!c3p_plan(N, H, W, Cin, Cout, dtype, p)) return 0;
    return p.npixwg * 2;
}

static int c3p_check(const char* who, const void* x, int ldx, const void* w, const void* y, int ldy, int N, int H, int W,
                     int Cin, int Cout, int dtype, C3pPlan& p) {
    FEA_CHECK(x && w && y, FEA_E_BADARG, "%s: null pointer", who);
    FEA_CHECK(c3p_plan(N, H, W, Cin, Cout, dtype, p), FEA_E_UNSUPPORTED_SHAPE,
              "%s: serves bf16, Cin in {64,128}, Cout in {64,128,256}, H %% 8 == 0, W %% 32 == 0 (got N=%d H=%d W=%d Cin=%d Cout=%d): use fea_conv3x3",
              who, N, H, W, Cin, Cout);
    FEA_CHECK(ldx >= Cin && ldy >= Cout && ldx % 8 == 0 && ldy % 8 == 0, FEA_E_BADARG, "%s: ldx=%d ldy=%d", who, ldx, ldy);
    FEA_CHECK(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)w % 16 == 0), FEA_E_BADARG, "%s: pointers must be 16-byte aligned", who);
    FEA_CHECK((long long)(H + 2) * W * ldx * 2 < (1ll << 31) && (long long)H * W * ldy * 2 < (1ll << 31), FEA_E_UNSUPPORTED_SHAPE,
              "%s: image too large for 32-bit in-image offsets", who);
    return FEA_OK;
}

static void c3p_fill(C3pArgs& a, const C3pPlan& p, const void* x, int ldx, const void* w, void* y, int ldy, int H, int W, int Cout) {
    a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.y = (__bf16*)y;
    a.ldx = ldx; a.ldy = ldy; a.H = H; a.W = W; a.CoutTot = Cout;
    a.x_img_stride = (long long)H * W * ldx; a.y_img_stride = (long long)H * W * ldy;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.tiles_per_wg = p.tiles_per_wg;
    a.npixwg = p.npixwg; a.nhalf = p.nhalf;
    a.wsx = (p.npixwg / p.tiles_y) % p.tiles_x; a.wsn = p.npixwg / (p.tiles_y * p.tiles_x);
}

// the interleaved tile walk (c3p_kernel's WALK): the 64 -> 64 layers on whole-column strides.  FEA_AMD_P_WALK=0 switches it off (A/B runs).
static bool c3p_walk(const C3pPlan& p, int Cin, int Cout) {
    static const bool off = getenv("FEA_AMD_P_WALK") && getenv("FEA_AMD_P_WALK")[0] == '0';
    return !off && Cin == 64 && Cout == 64 && p.npixwg % p.tiles_y == 0;
}

extern "C" int fea_conv3x3p(const void* x, int ldx, const void* w_frag, const float* bias, void* y, int ldy, float* stats,
                            int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    C3pPlan p;
    int rc = c3p_check("fea_conv3x3p", x, ldx, w_frag, y, ldy, N, H, W, Cin, Cout, dtype, p);
    if (rc) return rc;
    C3pArgs a{};
    c3p_fill(a, p, x, ldx, w_frag, y, ldy, H, W, Cout);
    a.bias = bias; a.stats = stats;
#ifdef C3P_STAMPS
    a.dbg = (unsigned long long*)getenv("C3P_DBG_PTR") ? (unsigned long long*)strtoull(getenv("C3P_DBG_PTR"), nullptr, 0) : nullptr;
#endif
    dim3 grid((unsigned)(p.npixwg * p.nhalf)), block(256);
    // forward (bias given): sum y and sum y^2; data gradient (no bias): sum only (the ConvTranspose2d bias gradient)
    const int st = !stats ? 0 : (bias ? 2 : 1);
#define C3P_LAUNCH(NCHV, STV, WV) hipLaunchKernelGGL((c3p_kernel<NCHV, 0, STV, WV>), grid, block, 0, (hipStream_t)stream, a)
    if (c3p_walk(p, Cin, Cout)) { if (st == 2) C3P_LAUNCH(1, 2, true); else if (st == 1) C3P_LAUNCH(1, 1, true); else C3P_LAUNCH(1, 0, true); }
    else if (Cin == 64) { if (st == 2) C3P_LAUNCH(1, 2, false); else if (st == 1) C3P_LAUNCH(1, 1, false); else C3P_LAUNCH(1, 0, false); }
    else                { if (st == 2) C3P_LAUNCH(2, 2, false); else if (st == 1) C3P_LAUNCH(2, 1, false); else C3P_LAUNCH(2, 0, false); }
#undef C3P_LAUNCH
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

extern "C" int fea_conv3x3p_dgrad_bnred(const void* dy, int lddy, const void* w_frag, void* da, int ldda,
                                        const void* y, int ldy_bn, const float* scale, const float* shift,
                                        const float* save_mean, const float* save_invstd, float* part,
                                        int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    C3pPlan p;
    const char* who = "fea_conv3x3p_dgrad_bnred";
    int rc = c3p_check(who, dy, lddy, w_frag, da, ldda, N, H, W, Cin, Cout, dtype, p);
    if (rc) return rc;
    FEA_CHECK(y && scale && shift && save_mean && save_invstd && part, FEA_E_BADARG, "%s: null pointer", who);
    FEA_CHECK(ldy_bn >= Cout && ldy_bn % 8 == 0 && ((uintptr_t)y % 16 == 0) && (long long)H * W * ldy_bn * 2 < (1ll << 31), FEA_E_BADARG, "%s: y layout", who);
    C3pArgs a{};
    c3p_fill(a, p, dy, lddy, w_frag, da, ldda, H, W, Cout);
    a.ybn = (const __bf16*)y; a.ldybn = ldy_bn; a.ybn_img_stride = (long long)H * W * ldy_bn;
    a.bn_scale = scale; a.bn_shift = shift; a.bn_mean = save_mean; a.bn_invstd = save_invstd; a.red_part = part;
    dim3 grid((unsigned)(p.npixwg * p.nhalf)), block(256);
    if (c3p_walk(p, Cin, Cout)) hipLaunchKernelGGL((c3p_kernel<1, 2, 0, true>), grid, block, 0, (hipStream_t)stream, a);
    else if (Cin == 64) hipLaunchKernelGGL((c3p_kernel<1, 2, 0>), grid, block, 0, (hipStream_t)stream, a);
    else                hipLaunchKernelGGL((c3p_kernel<2, 2, 0>), grid, block, 0, (hipStream_t)stream, a);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}

// Eval-mode inference with BatchNorm folded into the packed weights and the bias (see fea_conv3x3_bias_relu), persistent form.
extern "C" int fea_conv3x3p_bias_relu(const void* x, int ldx, const void* w_frag, const float* bias, void* y, int ldy,
                                      void* pool, int ldpool, int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    C3pPlan p;
    const char* who = "fea_conv3x3p_bias_relu";
    int rc = c3p_check(who, x, ldx, w_frag, y, ldy, N, H, W, Cin, Cout, dtype, p);
    if (rc) return rc;
    FEA_CHECK(bias, FEA_E_BADARG, "%s: no bias", who);
    FEA_CHECK(!pool || (ldpool >= Cout && ldpool % 8 == 0 && (uintptr_t)pool % 16 == 0), FEA_E_BADARG, "%s: pooled output layout", who);
    C3pArgs a{};
    c3p_fill(a, p, x, ldx, w_frag, y, ldy, H, W, Cout);
    a.bias = bias; a.pool = (__bf16*)pool; a.ldpool = ldpool; a.pool_img_stride = (long long)(H / 2) * (W / 2) * ldpool;
    dim3 grid((unsigned)(p.npixwg * p.nhalf)), block(256);
    if (c3p_walk(p, Cin, Cout)) hipLaunchKernelGGL((c3p_kernel<1, 3, 0, true>), grid, block, 0, (hipStream_t)stream, a);
    else if (Cin == 64) hipLaunchKernelGGL((c3p_kernel<1, 3, 0>), grid, block, 0, (hipStream_t)stream, a);
    else                hipLaunchKernelGGL((c3p_kernel<2, 3, 0>), grid, block, 0, (hipStream_t)stream, a);
    FEA_LAUNCH_CHECK();
    return FEA_OK;
}
