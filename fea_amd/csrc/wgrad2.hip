// conv3x3 weight gradient, bf16, LDS-DMA pipelined (gfx950).
//
//   dW[co][ci][tap] = sum_p dy[p][co] * x[p + tap][ci]        (models.py:28,31 backward)
//
// Same math and LDS image as wgrad.hip MODE 0 (P = dy tile, Q = x halo tile,
// both [pixel][64 ch] 128-byte rows read with ds_read_b64_tr_b16), restructured
// so that the matrix pipe never waits for staging:
//   * one 512-thread workgroup per CU owns a 64(co) x 64(ci) x 9 output tile and
//     walks a range of 256-pixel tiles; the two wave groups (waves 0-3 / 4-7)
//     take alternate tile rows (a split of the reduction inside the workgroup,
//     two waves per SIMD for latency hiding);
//   * tiles are staged by global_load_lds (16 B per lane, no VGPR round trip)
//     into a two-stage LDS ring: the loads of tile t+1 are in flight while
//     tile t is multiplied; one barrier per tile.  The XOR swizzle is applied
//     on the per-lane SOURCE address (the LDS-DMA destination is lane-linear);
//     out-of-image halo rows read from a zero line.
//   * epilogue: the two groups are summed through LDS and the tile is written
//     in the parameter's own OIHW order ([co][ci][9], 16-byte stores), so the
//     cross-workgroup reduction is a plain vector sum.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;


struct Wgrad2Args {
    const __bf16* p; const __bf16* q; float* out;     // out: slab [split][M][Nn][9] (or dW itself when splits == 1)
    long long p_img_stride, q_img_stride, out_split_stride;
    int ldp, ldq;
    int H, W;
    int M, Nn;
    int tiles_x, tiles_y, ntiles, tiles_per_split;
    int nmt, nnt;
    int wstride, wsx, wsy, wsn;   // wgrad4_kernel: split s takes tiles s * tiles_per_split ... (wstride 1) or s, s + wstride, ...; (wsx, wsy, wsn) = the stride in (x, y, image)
    unsigned long long* dbg;      // WG2_STAMPS builds only: per-wave phase cycle sums [workgroup][wave][6]
};

__device__ __forceinline__ int swzw2(int x) {      // bit-reverse of bits 1..3 of x (as wgrad.hip)
    const int u = (x >> 1) & 7;
    return ((u & 1) << 2) | (u & 2) | (u >> 2);
}

// LDS-DMA loads of one 256-pixel tile (P rows then Q halo rows) into `sb`: piece IT of this wave (0..3: P, 4..9: Q).
// A CU takes ~25 cycles per 1 KB piece; with all 8 waves issuing their ten pieces at the top of a tile that was ~2,100
// cycles per tile in which no MFMA ran (cycle stamps, scripts/wgrad2_stamps.py), so the pieces are issued one per
// k-step behind the second MFMA of k-steps 0..9 instead (wg2_kstep), where the other wave of the SIMD covers them.
//
// FAST (whole tiles only, images below 2 GB): pdesc / qdesc hold each lane's byte offset from the tile origin, the source is
// a buffer descriptor of ONE image and a lane that must read zero padding (halo outside the image: the piece's 5 flag bits
// in qfl, masked by the tile's edge bits) gets an out-of-range offset, for which the hardware returns 0 -- 1 VALU per
// P piece and 4 per Q piece instead of ~20 (64-bit address arithmetic and selects): the issue sits in the MFMA stream.
template <int TWS, int IT, bool FAST>
__device__ __forceinline__ void wg2_issue_piece(const Wgrad2Args& g, const __bf16* pg, const __bf16* qg,
                                                const int (&pdesc)[4], const int (&qdesc)[6], unsigned qfl, int wave,
                                                int tile_x, int tile_y, int n_img, char* sb) {
    constexpr int TW = 1 << TWS, TH = 256 / TW;
    constexpr int QROWS = 352, PBYTES = 256 * 128;
    const int H = g.H, W = g.W;
    const int y0 = tile_y * TH, x0 = tile_x * TW;
    if constexpr (FAST) {
        if constexpr (IT < 4) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(pg + (long long)n_img * g.p_img_stride), 0,
                                                                                (int)(g.p_img_stride * 2), 0x00020000);
            const int voff = (y0 * W + x0) * g.ldp * 2 + pdesc[IT];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(sb + (IT * 8 + wave) * 1024), 16, voff, 0, 0, 0);
        } else {
            constexpr int it = IT - 4;
            if (it * 8 + wave < QROWS / 8) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(qg + (long long)n_img * g.q_img_stride), 0,
                                                                                    (int)(g.q_img_stride * 2), 0x00020000);
                const unsigned edge = (tile_y == 0 ? 1u : 0u) | (tile_y == g.tiles_y - 1 ? 2u : 0u) | (tile_x == 0 ? 4u : 0u) |
                                      (tile_x == g.tiles_x - 1 ? 8u : 0u) | 16u;
                const unsigned f = (qfl >> (5 * it)) & edge & 31u;
                const int voff = f ? (int)0xfffffff0u : (y0 * W + x0) * g.ldq * 2 + qdesc[it];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(sb + PBYTES + (it * 8 + wave) * 1024), 16, voff, 0, 0, 0);
            }
        }
        return;
    }
    if constexpr (IT < 4) {
        constexpr int it = IT;
        const __bf16* pimg = pg + (long long)n_img * g.p_img_stride;
        const int d = pdesc[it];
        const int gy = y0 + (d & 0xff) - 1, gx = x0 + ((d >> 8) & 0xff) - 1;
        const bool ok = gy < H && gx < W;
        const __bf16* src = ok ? pimg + (long long)(gy * W + gx) * g.ldp + (d >> 24) * 8
                               : reinterpret_cast<const __bf16*>(fea_zero_line);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sb + (it * 8 + wave) * 1024), 16, 0, 0);
    } else {
        constexpr int it = IT - 4;
        if (it * 8 + wave < QROWS / 8) {
            const __bf16* qimg = qg + (long long)n_img * g.q_img_stride;
            const int d = qdesc[it];
            const int gy = y0 + (d & 0xff) - 1, gx = x0 + ((d >> 8) & 0xff) - 1;
            const bool ok = !((d >> 16) & 1) && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const __bf16* src = ok ? qimg + (long long)(gy * W + gx) * g.ldq + (d >> 24) * 8
                                   : reinterpret_cast<const __bf16*>(fea_zero_line);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sb + PBYTES + (it * 8 + wave) * 1024), 16, 0, 0);
        }
    }
}
// FAST pieces with everything that does not depend on the piece made ONCE per tile (Wg2Tile): the image descriptors, the tile's
// byte offsets and its edge bits.  As separate calls of wg2_issue_piece every piece redid them under its own `if (has_next)`
// -- ~25 scalar instructions and a branch per piece, ten pieces per tile, in the in-order stream between the MFMAs.  "No next
// tile" is folded into the offsets (far out of range: the lanes receive zeros into the stage nobody reads): no branch at all.
struct Wg2Tile { __amdgpu_buffer_rsrc_t prs, qrs; int ptoff, qtoff; unsigned edge; };
template <int TWS>
__device__ __forceinline__ Wg2Tile wg2_tile(const Wgrad2Args& g, const __bf16* pg, const __bf16* qg, int tile_x, int tile_y, int n_img, bool live) {
    constexpr int TW = 1 << TWS, TH = 256 / TW;
    Wg2Tile t;
    t.prs = __builtin_amdgcn_make_buffer_rsrc((void*)(pg + (long long)n_img * g.p_img_stride), 0, (int)(g.p_img_stride * 2), 0x00020000);
    t.qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(qg + (long long)n_img * g.q_img_stride), 0, (int)(g.q_img_stride * 2), 0x00020000);
    const int org = (tile_y * TH) * g.W + tile_x * TW;
    t.ptoff = live ? org * g.ldp * 2 : 0x7ff00000;
    t.qtoff = live ? org * g.ldq * 2 : 0x7ff00000;
    t.edge = (tile_y == 0 ? 1u : 0u) | (tile_y == g.tiles_y - 1 ? 2u : 0u) | (tile_x == 0 ? 4u : 0u) | (tile_x == g.tiles_x - 1 ? 8u : 0u) | 16u;
    return t;
}
template <int IT>
__device__ __forceinline__ void wg2_piece_fast(const Wg2Tile& t, const int (&pdesc)[4], const int (&qdesc)[6], unsigned qfl, int wave, char* sb) {
    constexpr int QROWS = 352, PBYTES = 256 * 128;
    if constexpr (IT < 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t.prs, (lds_void_t*)(sb + (IT * 8 + wave) * 1024), 16, t.ptoff + pdesc[IT], 0, 0, 0);
    } else {
        constexpr int it = IT - 4;
        if (it * 8 + wave < QROWS / 8) {
            const unsigned f = (qfl >> (5 * it)) & t.edge & 31u;
            const int voff = f ? (int)0xfffffff0u : t.qtoff + qdesc[it];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(t.qrs, (lds_void_t*)(sb + PBYTES + (it * 8 + wave) * 1024), 16, voff, 0, 0, 0);
        }
    }
}

template <int TWS, bool FAST>
__device__ __forceinline__ void wg2_issue(const Wgrad2Args& g, const __bf16* pg, const __bf16* qg,
                                          const int (&pdesc)[4], const int (&qdesc)[6], unsigned qfl, int wave,
                                          int tile_x, int tile_y, int n_img, char* sb) {
    sfor<0, 10>([&](auto ic) { wg2_issue_piece<TWS, decltype(ic)::value, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, tile_x, tile_y, n_img, sb); });
}

// Transposing LDS read issued as inline asm: hipcc neither reorders it nor
// waits for it -- the waits are the counted s_waitcnt of wg2_wait below.
template <int OFF>
__device__ __forceinline__ void lds_tr(u32x2& dst, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// s_waitcnt lgkmcnt(N), tied to the fragments it retires so that the MFMA that
// consumes them cannot be hoisted above it.
// (No register operands: with "+v" ties hipcc pads a hazard s_nop in front of every MFMA -- 80 per tile and wave in a loop that
// is short of issue slots.  The sched_barrier keeps the consuming MFMA, and everything else, on its side of the wait.)
#ifndef WG2_TIED_WAITS
#define WG2_TIED_WAITS 0
#endif
template <int N>
__device__ __forceinline__ void wg2_wait(u32x2& a0, u32x2& a1, u32x2& b0, u32x2& b1) {
    if constexpr (WG2_TIED_WAITS) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : "n"(N));
    else { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }
}
template <int N>
__device__ __forceinline__ void wg2_wait(u32x2& b0, u32x2& b1) {
    if constexpr (WG2_TIED_WAITS) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(N));
    else { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }
}

template <int NT> struct Wg2Frag { u32x2 a[2]; u32x2 b[NT][2]; };

// reads issued after MFMA t of a step (the NRD = 2 + 2 NT reads of the NEXT step)
template <int NT> __device__ constexpr int wg2_cnt(int t) { return (2 + 2 * NT + NT - 1 - t) / NT; }
template <int NT> __device__ constexpr int wg2_start(int t) { int s = 0; for (int u = 0; u < t; ++u) s += wg2_cnt<NT>(u); return s; }

// read number R (0,1 = A halves; 2+2t, 3+2t = B halves of tap t) of k-step S
template <int TWS, int TAP0, int NT, int S, int R>
__device__ __forceinline__ void wg2_read(Wg2Frag<NT>& f, const uint32_t (&pa)[(1 << TWS) / 16][2],
                                         const uint32_t (&qa)[(1 << TWS) / 16][2][3]) {
    constexpr int TW = 1 << TWS, HW2 = TW + 2, XSEG = TW / 16;
    constexpr int ty = S / XSEG, xs = S % XSEG;
    if constexpr (R < 2) {
        lds_tr<ty * TW * 128>(f.a[R], pa[xs][R]);
    } else {
        constexpr int t = (R - 2) / 2, hf = (R - 2) % 2, tap = TAP0 + t, dy = tap / 3, dx = tap % 3;
        lds_tr<(ty + dy) * HW2 * 128>(f.b[t][hf], qa[xs][hf][dx]);
    }
}

// k-step S: its fragments are in `cur`, those of S + 1 are in flight into `nxt` (issued during S - 1), and the reads of
// S + 2 are issued into `far` between the MFMAs of this step: an LDS read has two k-steps (>= 256 cycles of MFMA) to
// return.  With one step of distance (round 1) a wave alone reached 63 cycles per MFMA, i.e. every k-step exposed ~150
// cycles of LDS latency, and the second wave of the SIMD did not fill them (cycle stamps, scripts/wgrad2_stamps.py).
#ifdef WG2_STAMPS
#define WG2_PH_PARAM , unsigned long long (&ph)[6], unsigned long long& tprev
#define WG2_PH_ARG , ph, tprev
#define WSTAMP(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define WG2_PH_PARAM
#define WG2_PH_ARG
#define WSTAMP(k) do {} while (0)
#endif
template <int TWS, int TAP0, int NT, int S, typename PieceFn>
__device__ __forceinline__ void wg2_kstep(Wg2Frag<NT>& cur, Wg2Frag<NT>& nxt, Wg2Frag<NT>& far, const uint32_t (&pa)[(1 << TWS) / 16][2],
                                          const uint32_t (&qa)[(1 << TWS) / 16][2][3], f32x16 (&acc)[NT], PieceFn&& piece WG2_PH_PARAM) {
    if constexpr (S == 8) WSTAMP(4);
    constexpr int NSTEP = 256 / 16, NRD = 2 + 2 * NT;
#ifndef WG2_DEPTH
#define WG2_DEPTH 1
#endif
    constexpr int D = WG2_DEPTH;                       // k-steps of read-ahead (1 or 2)
    constexpr bool more1 = S + 1 < NSTEP, more2 = S + D < NSTEP;
    sfor<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        // reads in flight behind the ones this MFMA needs: the rest of this step's, all of step S + 1, and step S + 2 so far
        constexpr int behind = (D == 2 && more1 ? NRD : 0) + (more2 ? wg2_start<NT>(t) : 0);
        // (lgkmcnt is a 4-bit counter: at most 15 can be named, and a wave cannot have more LDS operations in flight)
        constexpr int n0 = NRD - 4 + behind, nt = NRD - (4 + 2 * t) + behind;
        if constexpr (t == 0) wg2_wait<(n0 < 15 ? n0 : 15)>(cur.a[0], cur.a[1], cur.b[0][0], cur.b[0][1]);
        else                  wg2_wait<(nt < 15 ? nt : 15)>(cur.b[t][0], cur.b[t][1]);
        const u32x4 af = {cur.a[0].x, cur.a[0].y, cur.a[1].x, cur.a[1].y};
        const u32x4 bf = {cur.b[t][0].x, cur.b[t][0].y, cur.b[t][1].x, cur.b[t][1].y};
        mma16(acc[t], af, bf, (const __bf16*)nullptr);
        if constexpr (more2) {
            sfor<0, wg2_cnt<NT>(t)>([&](auto rc) {
                wg2_read<TWS, TAP0, NT, S + D, wg2_start<NT>(t) + decltype(rc)::value>(D == 2 ? far : nxt, pa, qa);
            });
        }
#ifndef WG2_PPS
#define WG2_PPS 1
#endif
        // the ten LDS-DMA pieces of the next tile: WG2_PPS per k-step, behind the MFMAs of taps 1 (and 3)
        if constexpr (WG2_PPS == 1) { if constexpr (t == 1 && S < 10) piece(std::integral_constant<int, S>{}); }
        else {
            if constexpr (t == 1 && S < 5) piece(std::integral_constant<int, 2 * S>{});
            if constexpr (t == 3 && S < 5) piece(std::integral_constant<int, 2 * S + 1>{});
        }
    });
    if constexpr (more1) {
        if constexpr (D == 2) wg2_kstep<TWS, TAP0, NT, S + 1>(nxt, far, cur, pa, qa, acc, piece WG2_PH_ARG);
        else                  wg2_kstep<TWS, TAP0, NT, S + 1>(nxt, cur, far, pa, qa, acc, piece WG2_PH_ARG);
    }
}

// One pipeline step: start the LDS-DMA loads of `next_tile` into the other
// stage, then multiply the tile resident at LDS byte address `rd` for taps
// TAP0 .. TAP0+NT-1.  The sixteen 16-pixel k-steps are fully unrolled and
// software-pipelined by hand: the fragments of step s+1 are read from LDS
// (inline asm) between the MFMAs of step s and retired by counted lgkmcnt.
template <int TWS, int TAP0, int NT, bool FAST>
__device__ __forceinline__ void wg2_step(const Wgrad2Args& g, const __bf16* pg, const __bf16* qg,
                                         const int (&pdesc)[4], const int (&qdesc)[6], unsigned qfl, int wave, bool has_next, int ntx, int nty, int nn,
                                         char* dma, uint32_t rd,
                                         const int (&pcol)[(1 << TWS) / 16][2], const int (&qcol)[(1 << TWS) / 16][2][3],
                                         f32x16 (&acc)[NT] WG2_PH_PARAM) {
    constexpr int XSEG = (1 << TWS) / 16, PBYTES = 256 * 128;
    Wg2Tile nt{};
    if constexpr (FAST) nt = wg2_tile<TWS>(g, pg, qg, ntx, nty, has_next ? nn : 0, has_next);
    auto piece = [&](auto ic) {
        if constexpr (FAST) wg2_piece_fast<decltype(ic)::value>(nt, pdesc, qdesc, qfl, wave, dma);
        else { if (has_next) wg2_issue_piece<TWS, decltype(ic)::value, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, ntx, nty, nn, dma); }
    };
    WSTAMP(2);
    uint32_t pa[XSEG][2], qa[XSEG][2][3];
#pragma unroll
    for (int xs = 0; xs < XSEG; ++xs)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            pa[xs][t] = rd + pcol[xs][t];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) qa[xs][t][dx] = rd + PBYTES + qcol[xs][t][dx];
        }
    Wg2Frag<NT> f0, f1, f2;
    sfor<0, 2 + 2 * NT>([&](auto rc) { wg2_read<TWS, TAP0, NT, 0, decltype(rc)::value>(f0, pa, qa); });
    if constexpr (WG2_DEPTH == 2) sfor<0, 2 + 2 * NT>([&](auto rc) { wg2_read<TWS, TAP0, NT, 1, decltype(rc)::value>(f1, pa, qa); });
    wg2_kstep<TWS, TAP0, NT, 0>(f0, f1, f2, pa, qa, acc, piece WG2_PH_ARG);
}

// Tile loop + epilogue staging for one wave group (taps TAP0 .. TAP0+NT-1).
template <int TWS, int TAP0, int NT, bool FAST>
__device__ __forceinline__ void wg2_run(const Wgrad2Args& g, const __bf16* pg, const __bf16* qg,
                                        const int (&pdesc)[4], const int (&qdesc)[6], unsigned qfl, int wave, int t_begin, int t_end,
                                        char* smem, const int (&pcol)[(1 << TWS) / 16][2], const int (&qcol)[(1 << TWS) / 16][2][3],
                                        int orow, int ocol, int h) {
    constexpr int STAGE = (256 + 352) * 128;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // coordinates of the NEXT tile, advanced on the scalar unit (a division per tile costs ~60 instructions)
    int ntx, nty, nn;
    {
        int tt = t_begin;
        ntx = tt % g.tiles_x; tt /= g.tiles_x;
        nty = tt % g.tiles_y; nn = tt / g.tiles_y;
    }
#ifdef WG2_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int stage = (tile - t_begin) & 1;
        if (++ntx == g.tiles_x) { ntx = 0; if (++nty == g.tiles_y) { nty = 0; ++nn; } }
        WSTAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WSTAMP(0);
        __syncthreads();                 // this wave's LDS-DMA has landed; barrier: tile complete, other stage free
        WSTAMP(1);
        wg2_step<TWS, TAP0, NT, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, tile + 1 < t_end, ntx, nty, nn,
                                smem + (stage ^ 1) * STAGE, lds_base + stage * STAGE, pcol, qcol, acc WG2_PH_ARG);
        WSTAMP(5);
    }
#ifdef WG2_STAMPS
    if ((threadIdx.x & 63) == 0 && g.dbg) for (int k = 0; k < 6; ++k) g.dbg[(blockIdx.x * 8 + wave) * 6 + k] = ph[k];
#endif
    __syncthreads();                     // every wave is done reading the stages
    float* ot = reinterpret_cast<float*>(smem);       // [64 m][64 n][9]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int mr = orow + (e & 3) + 8 * (e >> 2) + 4 * h;
            ot[(mr * 64 + ocol) * 9 + TAP0 + t] = acc[t][e];
        }
}

template <int TWS, bool FAST>
__global__ __launch_bounds__(512, 2) void wgrad2_kernel(const Wgrad2Args g) {
    constexpr int TW = 1 << TWS, TH = 256 / TW, HW2 = TW + 2;
    constexpr int QROWS = 352;                       // >= (TH+2)*HW2 (340 / 324), multiple of 8
    constexpr int PBYTES = 256 * 128, QBYTES = QROWS * 128, STAGE = PBYTES + QBYTES;
    constexpr int NPI = 4, NQI = 6;                  // glds wave-instructions per wave: P rows 0..255, Q rows 0..351
    constexpr int XSEG = TW / 16;
    __shared__ __attribute__((aligned(128))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = wave >> 2;                        // tap group
    const int wa = (wave >> 1) & 1, wb = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntp = g.nmt * g.nnt;
    const int tp = lid % ntp, split = lid / ntp;
    const int m0 = (tp / g.nnt) * 64, n0 = (tp % g.nnt) * 64;
    const int H = g.H, W = g.W;

    // ---- fragment offsets (identical to wgrad.hip) -------------------------
    int pcol[XSEG][2], qcol[XSEG][2][3];
    {
        const int gi = lane >> 4, li = lane & 15;
        const int chl = 16 * (gi & 1) + 4 * (li & 3);
        const int qrow = li >> 2;
#pragma unroll
        for (int xs = 0; xs < XSEG; ++xs)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int x = xs * 16 + 8 * h + 4 * t + qrow;
                const int chp = wa * 32 + chl, chq = wb * 32 + chl;
                pcol[xs][t] = x * 128 + ((((chp >> 3) ^ swzw2(x))) << 4) + (chp & 7) * 2;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int hx = x + dx;
                    qcol[xs][t][dx] = hx * 128 + ((((chq >> 3) ^ swzw2(hx))) << 4) + (chq & 7) * 2;
                }
            }
    }

    // ---- LDS-DMA descriptors: instruction it of this wave moves 8 rows ------
    // packed: (ry+1) | (rx+1) << 8 | invalid << 16 | source chunk << 24
    // FAST: byte offsets from the tile origin instead, and 5 flag bits per Q piece (top, bottom, left, right halo line, padding row)
    int pdesc[NPI], qdesc[NQI];
    unsigned qfl = 0;
    {
        const int rsub = lane >> 3, cpos = lane & 7;
#pragma unroll
        for (int it = 0; it < NPI; ++it) {
            const int row = (it * 8 + wave) * 8 + rsub;
            const int ty = row >> TWS, tx = row & (TW - 1);
            if constexpr (FAST) pdesc[it] = (ty * g.W + tx) * g.ldp * 2 + ((cpos ^ swzw2(tx)) << 4);
            else pdesc[it] = (ty + 1) | ((tx + 1) << 8) | ((cpos ^ swzw2(tx)) << 24);
        }
#pragma unroll
        for (int it = 0; it < NQI; ++it) {
            const int row = (it * 8 + wave) * 8 + rsub;
            const int hy = row / HW2, hx = row - hy * HW2;
            const int inval = hy >= TH + 2 ? 1 : 0;
            if constexpr (FAST) {
                qdesc[it] = ((hy - 1) * g.W + (hx - 1)) * g.ldq * 2 + ((cpos ^ swzw2(hx)) << 4);
                const unsigned f = (hy == 0 ? 1u : 0u) | (hy == TH + 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == TW + 1 ? 8u : 0u) | (inval ? 16u : 0u);
                qfl |= f << (5 * it);
            } else qdesc[it] = hy | (hx << 8) | (inval << 16) | ((cpos ^ swzw2(hx)) << 24);   // ry = hy-1, rx = hx-1
        }
    }

    const __bf16* __restrict__ pg = g.p + m0;
    const __bf16* __restrict__ qg = g.q + n0;
    const int t_begin = split * g.tiles_per_split;
    const int t_end = min(t_begin + g.tiles_per_split, g.ntiles);

    if (t_begin < t_end) {
        int tt = t_begin;
        const int tx = tt % g.tiles_x; tt /= g.tiles_x;
        wg2_issue<TWS, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, tx, tt % g.tiles_y, tt / g.tiles_y, smem);
    }
    // waves 0-3 take taps 0..4, waves 4-7 taps 5..8: every SIMD hosts one wave of each group
#ifndef WG2_PRIO
#define WG2_PRIO 1
#endif
    if (WG2_PRIO && kg == 1) __builtin_amdgcn_s_setprio(WG2_PRIO);
    if (kg == 0) wg2_run<TWS, 0, 5, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, t_begin, t_end, smem, pcol, qcol, wa * 32, wb * 32 + r, h);
    else         wg2_run<TWS, 5, 4, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, t_begin, t_end, smem, pcol, qcol, wa * 32, wb * 32 + r, h);
    __syncthreads();
    const float* ot = reinterpret_cast<const float*>(smem);
    float* dst = g.out + (long long)split * g.out_split_stride + ((long long)m0 * g.Nn + n0) * 9;
    // 64 rows x 576 floats; 144 float4 per row
    for (int i = tid; i < 64 * 144; i += 512) {
        const int row = i / 144, c4 = i - row * 144;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * 576 + c4 * 4);
        *reinterpret_cast<f32x4*>(dst + (long long)row * g.Nn * 9 + c4 * 4) = v;
    }
}

// ---------------------------------------------------------------------------
// wgrad3: the same kernel on v_mfma_f32_16x16x32_bf16 (8x32-pixel tiles only).
//
// Same workgroup tile (64 co x 64 ci x 9 taps over a range of 256-pixel tiles), same wave roles (wa: co half, wb: ci half,
// waves 0-3 taps 0-4, waves 4-7 taps 5-8), same LDS image and LDS-DMA pieces as wgrad2_kernel<5, *>; the wave's 32 x 32 block
// of a tap is four 16 x 16 accumulator tiles and a k-step is one tile row of 32 pixels (8 k-steps per tile).  Operand bytes
// read from LDS per FLOP are unchanged; the chip holds a higher clock on the 16x16x32 form of an LDS-fed bf16 loop
// (MI355X_MICROARCH.md, 'DVFS give-back' item 7; the persistent conv kernel measured 10-20 % between the two forms).
//
// Fragment of a 16-row operand tile for k = pixels 8 g .. 8 g + 7 (g = lane >> 4): two ds_read_b64_tr_b16, read t = 0 / 1
// fetches pixels 8 g + 4 t + (li >> 2), channels 4 (li & 3) .. + 3 of the tile (li = lane & 15) and returns, transposed within
// the 16 lanes, channel li at pixels 8 g + 4 t .. + 3.  Lanes 0-31 of one read touch 8 pixel rows {b .. b+3, b+8 .. b+11} x 32 B:
// the XOR below (chunk bits 2, 1 <- pixel bits 1, 3; bit 0 of the chunk untouched so the 32 B stay together) puts them in
// 8 different 32-byte bank groups for every b, i.e. for every halo shift dx.
__device__ __forceinline__ int swzw3(int x) { return (((x >> 1) & 1) << 2) | (((x >> 3) & 1) << 1); }

template <int NT> struct Wg3Frag { u32x2 a[2][2]; u32x2 b[NT][2][2]; };      // [m or n tile][read t]

template <int NT> __device__ constexpr int wg3_cnt(int t) { return (4 + 4 * NT + NT - 1 - t) / NT; }
template <int NT> __device__ constexpr int wg3_start(int t) { int s = 0; for (int u = 0; u < t; ++u) s += wg3_cnt<NT>(u); return s; }

// read number R of k-step S: 0..3 = A (m tile R >> 1, half R & 1); 4 + 4 t + 2 n + hf = B of tap t, n tile n, half hf
template <int TAP0, int NT, int S, int R>
__device__ __forceinline__ void wg3_read(Wg3Frag<NT>& f, const uint32_t (&pa)[2][2], const uint32_t (&qa)[2][2][3]) {
    constexpr int HW2 = 34;
    if constexpr (R < 4) {
        lds_tr<S * 32 * 128>(f.a[R >> 1][R & 1], pa[R >> 1][R & 1]);
    } else {
        constexpr int t = (R - 4) / 4, n = ((R - 4) / 2) % 2, hf = (R - 4) % 2, tap = TAP0 + t, dy = tap / 3, dx = tap % 3;
        lds_tr<(S + dy) * HW2 * 128>(f.b[t][n][hf], qa[n][hf][dx]);
    }
}

template <int N>
__device__ __forceinline__ void wg3_wait_ab(u32x2 (&a)[2][2], u32x2 (&b)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wg3_wait_b(u32x2 (&b)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0);
}

// k-step S: fragments in `cur`; those of S + 1 are read into `nxt` between the MFMAs of this step.
template <int TAP0, int NT, int S, typename PieceFn>
__device__ __forceinline__ void wg3_kstep(Wg3Frag<NT>& cur, Wg3Frag<NT>& nxt, const uint32_t (&pa)[2][2], const uint32_t (&qa)[2][2][3],
                                          f32x4 (&acc)[NT][2][2], PieceFn&& piece) {
    constexpr int NSTEP = 8, NRD = 4 + 4 * NT;
    constexpr bool more = S + 1 < NSTEP;
    sfor<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        // reads issued behind the ones this tap needs: the rest of this step's (taps t + 1 ..) and the next step's so far
        constexpr int behind = NRD - (8 + 4 * t) + (more ? wg3_start<NT>(t) : 0);
        if constexpr (t == 0) wg3_wait_ab<(behind < 15 ? behind : 15)>(cur.a, cur.b[0]);
        else                  wg3_wait_b<(behind < 15 ? behind : 15)>(cur.b[t]);
        u32x4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = u32x4{cur.a[i][0].x, cur.a[i][0].y, cur.a[i][1].x, cur.a[i][1].y};
            bf[i] = u32x4{cur.b[t][i][0].x, cur.b[t][i][0].y, cur.b[t][i][1].x, cur.b[t][i][1].y};
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[t][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[mi]), __builtin_bit_cast(bf16x8, bf[ni]),
                                                                         acc[t][mi][ni], 0, 0, 0);
        if constexpr (more) {
            sfor<0, wg3_cnt<NT>(t)>([&](auto rc) {
                wg3_read<TAP0, NT, S + 1, wg3_start<NT>(t) + decltype(rc)::value>(nxt, pa, qa);
            });
        }
        // the ten LDS-DMA pieces of the next tile: one per k-step behind tap 1, the last two behind tap 3 of k-steps 0 and 1
        if constexpr (t == 1) piece(std::integral_constant<int, S>{});
        if constexpr (t == 3 && S < 2) piece(std::integral_constant<int, 8 + S>{});
    });
    if constexpr (more) wg3_kstep<TAP0, NT, S + 1>(nxt, cur, pa, qa, acc, piece);
}

template <int TAP0, int NT, bool FAST>
__device__ __forceinline__ void wg3_run(const Wgrad2Args& g, const __bf16* pg, const __bf16* qg,
                                        const int (&pdesc)[4], const int (&qdesc)[6], unsigned qfl, int wave, int t_begin, int t_end,
                                        char* smem, const int (&pcol)[2][2], const int (&qcol)[2][2][3], int orow, int ocol) {
    constexpr int STAGE = (256 + 352) * 128, PBYTES = 256 * 128;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    f32x4 acc[NT][2][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    int ntx, nty, nn;
    {
        int tt = t_begin;
        ntx = tt % g.tiles_x; tt /= g.tiles_x;
        nty = tt % g.tiles_y; nn = tt / g.tiles_y;
    }
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int stage = (tile - t_begin) & 1;
        if (++ntx == g.tiles_x) { ntx = 0; if (++nty == g.tiles_y) { nty = 0; ++nn; } }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // this wave's LDS-DMA has landed; barrier: tile complete, other stage free
        const bool has_next = tile + 1 < t_end;
        char* dma = smem + (stage ^ 1) * STAGE;
        const uint32_t rd = lds_base + stage * STAGE;
        Wg2Tile nt{};
        if constexpr (FAST) nt = wg2_tile<5>(g, pg, qg, ntx, nty, has_next ? nn : 0, has_next);
        auto piece = [&](auto ic) {
            if constexpr (FAST) wg2_piece_fast<decltype(ic)::value>(nt, pdesc, qdesc, qfl, wave, dma);
            else { if (has_next) wg2_issue_piece<5, decltype(ic)::value, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, ntx, nty, nn, dma); }
        };
        uint32_t pa[2][2], qa[2][2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                pa[i][t] = rd + pcol[i][t];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) qa[i][t][dx] = rd + PBYTES + qcol[i][t][dx];
            }
        Wg3Frag<NT> f0, f1;
        sfor<0, 4 + 4 * NT>([&](auto rc) { wg3_read<TAP0, NT, 0, decltype(rc)::value>(f0, pa, qa); });
        wg3_kstep<TAP0, NT, 0>(f0, f1, pa, qa, acc, piece);
    }
    __syncthreads();                     // every wave is done reading the stages
    float* ot = reinterpret_cast<float*>(smem);       // [64 m][64 n][9]
    const int gi = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int mr = orow + mi * 16 + 4 * gi + e;
                    ot[(mr * 64 + ocol + ni * 16) * 9 + TAP0 + t] = acc[t][mi][ni][e];
                }
}

template <bool FAST>
__global__ __launch_bounds__(512, 2) void wgrad3_kernel(const Wgrad2Args g) {
    constexpr int TW = 32, TH = 8, HW2 = TW + 2;
    constexpr int QROWS = 352;
    constexpr int PBYTES = 256 * 128, QBYTES = QROWS * 128, STAGE = PBYTES + QBYTES;
    constexpr int NPI = 4, NQI = 6;
    __shared__ __attribute__((aligned(128))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = wave >> 2;
    const int wa = (wave >> 1) & 1, wb = wave & 1;
    const int gi = lane >> 4, li = lane & 15;

    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntp = g.nmt * g.nnt;
    const int tp = lid % ntp, split = lid / ntp;
    const int m0 = (tp / g.nnt) * 64, n0 = (tp % g.nnt) * 64;

    // ---- fragment offsets -------------------------------------------------
    int pcol[2][2], qcol[2][2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int x = 8 * gi + 4 * t + (li >> 2);
            const int chp = wa * 32 + i * 16 + 4 * (li & 3), chq = wb * 32 + i * 16 + 4 * (li & 3);
            pcol[i][t] = x * 128 + (((chp >> 3) ^ swzw3(x)) << 4) + (chp & 7) * 2;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int hx = x + dx;
                qcol[i][t][dx] = hx * 128 + (((chq >> 3) ^ swzw3(hx)) << 4) + (chq & 7) * 2;
            }
        }

    // ---- LDS-DMA descriptors (as wgrad2_kernel, with this kernel's swizzle) -------
    int pdesc[NPI], qdesc[NQI];
    unsigned qfl = 0;
    {
        const int rsub = lane >> 3, cpos = lane & 7;
#pragma unroll
        for (int it = 0; it < NPI; ++it) {
            const int row = (it * 8 + wave) * 8 + rsub;
            const int ty = row >> 5, tx = row & (TW - 1);
            if constexpr (FAST) pdesc[it] = (ty * g.W + tx) * g.ldp * 2 + ((cpos ^ swzw3(tx)) << 4);
            else pdesc[it] = (ty + 1) | ((tx + 1) << 8) | ((cpos ^ swzw3(tx)) << 24);
        }
#pragma unroll
        for (int it = 0; it < NQI; ++it) {
            const int row = (it * 8 + wave) * 8 + rsub;
            const int hy = row / HW2, hx = row - hy * HW2;
            const int inval = hy >= TH + 2 ? 1 : 0;
            if constexpr (FAST) {
                qdesc[it] = ((hy - 1) * g.W + (hx - 1)) * g.ldq * 2 + ((cpos ^ swzw3(hx)) << 4);
                const unsigned f = (hy == 0 ? 1u : 0u) | (hy == TH + 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == TW + 1 ? 8u : 0u) | (inval ? 16u : 0u);
                qfl |= f << (5 * it);
            } else qdesc[it] = hy | (hx << 8) | (inval << 16) | ((cpos ^ swzw3(hx)) << 24);
        }
    }

    const __bf16* __restrict__ pg = g.p + m0;
    const __bf16* __restrict__ qg = g.q + n0;
    const int t_begin = split * g.tiles_per_split;
    const int t_end = min(t_begin + g.tiles_per_split, g.ntiles);

    if (t_begin < t_end) {
        int tt = t_begin;
        const int tx = tt % g.tiles_x; tt /= g.tiles_x;
        wg2_issue<5, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, tx, tt % g.tiles_y, tt / g.tiles_y, smem);
    }
    if (WG2_PRIO && kg == 1) __builtin_amdgcn_s_setprio(WG2_PRIO);
    if (kg == 0) wg3_run<0, 5, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, t_begin, t_end, smem, pcol, qcol, wa * 32, wb * 32 + li);
    else         wg3_run<5, 4, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, t_begin, t_end, smem, pcol, qcol, wa * 32, wb * 32 + li);
    __syncthreads();
    const float* ot = reinterpret_cast<const float*>(smem);
    float* dst = g.out + (long long)split * g.out_split_stride + ((long long)m0 * g.Nn + n0) * 9;
    for (int i = tid; i < 64 * 144; i += 512) {
        const int row = i / 144, c4 = i - row * 144;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * 576 + c4 * 4);
        *reinterpret_cast<f32x4*>(dst + (long long)row * g.Nn * 9 + c4 * 4) = v;
    }
}

// ---------------------------------------------------------------------------
// wgrad4: the 16x16x32 kernel re-tiled for fewer LDS reads per MFMA (within what 8 waves of 256 registers hold).
//
// Same workgroup tile, LDS image, swizzle and LDS-DMA pieces as wgrad3.  Wave w: ci half h = w & 1 (two 16-column N tiles), ALL 64
// co (four 16-row M tiles), tap group G = w >> 1: taps 2G and 2G + 1 in every k-step, and tap 8 in the k-steps S with
// (S & 3) == G (each of the four groups sums two of the eight tile rows of tap 8; the four partial sums meet in the epilogue
// in a fixed order).  Every wave thus multiplies 18 (tap, k-step) pairs of 8 MFMAs per tile -- balanced -- and reads 8 A + 8 or 12
// B fragments per k-step: 0.94 transposing reads per MFMA against 1.2-1.25 in wgrad2 / wgrad3, whose waves own 32 co only.
template <int G, int S> __device__ constexpr int wg4_nt() { return 2 + (((S & 3) == G) ? 1 : 0); }
template <int G> __device__ constexpr int wg4_tap(int slot) { return slot < 2 ? 2 * G + slot : 8; }
struct Wg4Frag { u32x2 a[4][2]; u32x2 b[3][2][2]; };      // A: [M tile][read], B: [tap slot][N tile][read]

template <int G, int S, int R>
__device__ __forceinline__ void wg4_read(Wg4Frag& f, const uint32_t (&pa)[4][2], const uint32_t (&qa)[2][2][3]) {
    constexpr int HW2 = 34;
    if constexpr (R < 8) {
        lds_tr<S * 32 * 128>(f.a[R >> 1][R & 1], pa[R >> 1][R & 1]);
    } else {
        constexpr int sl = (R - 8) / 4, n = ((R - 8) / 2) % 2, hf = (R - 8) % 2, tap = wg4_tap<G>(sl), dy = tap / 3, dx = tap % 3;
        lds_tr<(S + dy) * HW2 * 128>(f.b[sl][n][hf], qa[n][hf][dx]);
    }
}
template <int N> __device__ __forceinline__ void wg4_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int G, int S, typename PieceFn>
__device__ __forceinline__ void wg4_kstep(Wg4Frag& cur, Wg4Frag& nxt, const uint32_t (&pa)[4][2], const uint32_t (&qa)[2][2][3],
                                          f32x4 (&acc)[3][4][2], PieceFn&& piece) {
    constexpr int NT = wg4_nt<G, S>(), NRD = 8 + 4 * NT;
    constexpr bool more = S + 1 < 8;
    constexpr int NRDN = more ? 8 + 4 * wg4_nt<G, (S + 1) & 7>() : 0;
    sfor<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        // the next step's reads are dealt over this step's taps; `behind` = reads issued after the ones tap t needs
        constexpr int cnt = (NRDN + NT - 1 - t) / NT;
        constexpr int start = [] { int s_ = 0; for (int u = 0; u < t; ++u) s_ += (NRDN + NT - 1 - u) / NT; return s_; }();
        constexpr int behind = NRD - (12 + 4 * t) + start;
        wg4_wait<(behind < 15 ? behind : 15)>();
        u32x4 af[4], bf[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = u32x4{cur.a[i][0].x, cur.a[i][0].y, cur.a[i][1].x, cur.a[i][1].y};
#pragma unroll
        for (int i = 0; i < 2; ++i) bf[i] = u32x4{cur.b[t][i][0].x, cur.b[t][i][0].y, cur.b[t][i][1].x, cur.b[t][i][1].y};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[t][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[mi]), __builtin_bit_cast(bf16x8, bf[ni]),
                                                                         acc[t][mi][ni], 0, 0, 0);
        if constexpr (more) {
            sfor<0, cnt>([&](auto rc) { wg4_read<G, S + 1, start + decltype(rc)::value>(nxt, pa, qa); });
        }
        // the ten LDS-DMA pieces of the next tile: WG4_PIECE_STEPS k-steps carry them, so that the last ones have
        // 8 - WG4_PIECE_STEPS k-steps (~300 cycles each) to land before the tile's barrier
#ifndef WG4_PIECE_STEPS
#define WG4_PIECE_STEPS 3        // 5 (two pieces per k-step) measured 0-1.5 % slower, 8 (one per k-step: the last land too late) 4-6 % slower
#endif
        if constexpr (WG4_PIECE_STEPS == 8) {
            if constexpr (t == 1) piece(std::integral_constant<int, S>{});
            if constexpr (t == 0 && S < 2) piece(std::integral_constant<int, 8 + S>{});
        } else if constexpr (WG4_PIECE_STEPS == 3) {
            if constexpr (t < 2 && 4 * S + 2 * t < 10) piece(std::integral_constant<int, 4 * S + 2 * t>{});
            if constexpr (t < 2 && 4 * S + 2 * t + 1 < 10) piece(std::integral_constant<int, 4 * S + 2 * t + 1>{});
        } else {
            if constexpr (t < 2 && 2 * S + t < 10) piece(std::integral_constant<int, 2 * S + t>{});
        }
    });
    if constexpr (more) wg4_kstep<G, S + 1>(nxt, cur, pa, qa, acc, piece);
}

template <int G, bool FAST>
__device__ __forceinline__ void wg4_run(const Wgrad2Args& g, const __bf16* pg, const __bf16* qg,
                                        const int (&pdesc)[4], const int (&qdesc)[6], unsigned qfl, int wave, int first, int t_begin, int t_end,
                                        char* smem, uint32_t (&pa)[4][2], uint32_t (&qa)[2][2][3], int h) {
    // (t_begin .. t_end count this workgroup's own tiles; `first` is the tile the count starts at, the cursor advances by the walk's stride)
    constexpr int STAGE = (256 + 352) * 128;
    f32x4 acc[3][4][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[t][i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    int ntx, nty, nn;
    {
        int tt = first;
        ntx = tt % g.tiles_x; tt /= g.tiles_x;
        nty = tt % g.tiles_y; nn = tt / g.tiles_y;
    }
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int stage = (tile - t_begin) & 1;
        ntx += g.wsx; if (ntx >= g.tiles_x) { ntx -= g.tiles_x; ++nty; }
        nty += g.wsy; if (nty >= g.tiles_y) { nty -= g.tiles_y; ++nn; }
        nn += g.wsn;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef WG4_EXP_NO_BARRIER       // EXPERIMENT (racy): no per-tile barrier
        __syncthreads();                 // this wave's LDS-DMA has landed; barrier: tile complete, other stage free
#endif
        const bool has_next = tile + 1 < t_end;
        char* dma = smem + (stage ^ 1) * STAGE;
        Wg2Tile nt{};
#ifdef WG4_EXP_HOT_DMA          // EXPERIMENT (wrong results): every tile's pieces come from tile 1 of image 0 (cache-hot): issue + LDS-write cost without the memory system
        if constexpr (FAST) nt = wg2_tile<5>(g, pg, qg, 1, 1, 0, true);
#else
        if constexpr (FAST) nt = wg2_tile<5>(g, pg, qg, ntx, nty, has_next ? nn : 0, has_next);
#endif
        auto piece = [&](auto ic) {
#ifndef WG4_EXP_NO_DMA       // EXPERIMENT (wrong results): no LDS-DMA after the first tile
            if constexpr (FAST) wg2_piece_fast<decltype(ic)::value>(nt, pdesc, qdesc, qfl, wave, dma);
            else { if (has_next) wg2_issue_piece<5, decltype(ic)::value, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, ntx, nty, nn, dma); }
#endif
        };
        Wg4Frag f0, f1;
        sfor<0, 8 + 4 * wg4_nt<G, 0>()>([&](auto rc) { wg4_read<G, 0, decltype(rc)::value>(f0, pa, qa); });
        wg4_kstep<G, 0>(f0, f1, pa, qa, acc, piece);
        // the fragment addresses follow the stage ring in place (no second set of address registers)
        const int delta = stage ? -STAGE : STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) pa[i][t] += delta;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) qa[i][t][dx] += delta;
    }
    __syncthreads();                     // every wave is done reading the stages
    float* ot = reinterpret_cast<float*>(smem);       // [64 co][64 ci][9]
    const int lane = threadIdx.x & 63, gi = lane >> 4, li = lane & 15;
    auto put = [&](int slot, int tap, bool add) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* o = ot + ((mi * 16 + 4 * gi + e) * 64 + h * 32 + ni * 16 + li) * 9 + tap;
                    *o = add ? *o + acc[slot][mi][ni][e] : acc[slot][mi][ni][e];
                }
    };
    put(0, 2 * G, false);
    put(1, 2 * G + 1, false);
    // tap 8: the four groups' partial sums, added in group order (fixed: bit-identical from run to run)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r == G) put(2, 8, r != 0);
        __syncthreads();
    }
}

template <bool FAST>
__global__ __launch_bounds__(512, 2) void wgrad4_kernel(const Wgrad2Args g) {
    constexpr int TW = 32, TH = 8, HW2 = TW + 2;
    constexpr int QROWS = 352;
    constexpr int PBYTES = 256 * 128, QBYTES = QROWS * 128, STAGE = PBYTES + QBYTES;
    constexpr int NPI = 4, NQI = 6;
    __shared__ __attribute__((aligned(128))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = wave & 1, grp = wave >> 1;
    const int gi = lane >> 4, li = lane & 15;

    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ntp = g.nmt * g.nnt;
    const int tp = lid % ntp, split = lid / ntp;
    const int m0 = (tp / g.nnt) * 64, n0 = (tp % g.nnt) * 64;

    // ---- fragment addresses of stage 0 (kept up to date in place by wg4_run) --------------------
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t pa[4][2], qa[2][2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int x = 8 * gi + 4 * t + (li >> 2);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int chp = mi * 16 + 4 * (li & 3);
            pa[mi][t] = lds_base + x * 128 + (((chp >> 3) ^ swzw3(x)) << 4) + (chp & 7) * 2;
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int chq = h * 32 + ni * 16 + 4 * (li & 3);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int hx = x + dx;
                qa[ni][t][dx] = lds_base + PBYTES + hx * 128 + (((chq >> 3) ^ swzw3(hx)) << 4) + (chq & 7) * 2;
            }
        }
    }

    // ---- LDS-DMA descriptors (as wgrad3_kernel) -------
    int pdesc[NPI], qdesc[NQI];
    unsigned qfl = 0;
    {
        const int rsub = lane >> 3, cpos = lane & 7;
#pragma unroll
        for (int it = 0; it < NPI; ++it) {
            const int row = (it * 8 + wave) * 8 + rsub;
            const int ty = row >> 5, tx = row & (TW - 1);
            if constexpr (FAST) pdesc[it] = (ty * g.W + tx) * g.ldp * 2 + ((cpos ^ swzw3(tx)) << 4);
            else pdesc[it] = (ty + 1) | ((tx + 1) << 8) | ((cpos ^ swzw3(tx)) << 24);
        }
#pragma unroll
        for (int it = 0; it < NQI; ++it) {
            const int row = (it * 8 + wave) * 8 + rsub;
            const int hy = row / HW2, hx = row - hy * HW2;
            const int inval = hy >= TH + 2 ? 1 : 0;
            if constexpr (FAST) {
                qdesc[it] = ((hy - 1) * g.W + (hx - 1)) * g.ldq * 2 + ((cpos ^ swzw3(hx)) << 4);
                const unsigned f = (hy == 0 ? 1u : 0u) | (hy == TH + 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == TW + 1 ? 8u : 0u) | (inval ? 16u : 0u);
                qfl |= f << (5 * it);
            } else qdesc[it] = hy | (hx << 8) | (inval << 16) | ((cpos ^ swzw3(hx)) << 24);
        }
    }

    const __bf16* __restrict__ pg = g.p + m0;
    const __bf16* __restrict__ qg = g.q + n0;
    // tile walk: consecutive tiles of a split (wstride 1), or every wstride-th tile: the chip then reads one compact window of
    // wstride tiles at any moment (the 512 x 512 layers are HBM-bound; conv_p.hip's WALK)
    const int first = g.wstride == 1 ? split * g.tiles_per_split : split;
    const int t_begin = 0;
    const int t_end = first >= g.ntiles ? 0 : g.wstride == 1 ? min(g.tiles_per_split, g.ntiles - first) : (g.ntiles - first + g.wstride - 1) / g.wstride;

    if (t_begin < t_end) {
        int tt = first;
        const int tx = tt % g.tiles_x; tt /= g.tiles_x;
        wg2_issue<5, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, tx, tt % g.tiles_y, tt / g.tiles_y, smem);
    }
    if (WG2_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(WG2_PRIO);
    if (grp == 0)      wg4_run<0, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, first, t_begin, t_end, smem, pa, qa, h);
    else if (grp == 1) wg4_run<1, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, first, t_begin, t_end, smem, pa, qa, h);
    else if (grp == 2) wg4_run<2, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, first, t_begin, t_end, smem, pa, qa, h);
    else               wg4_run<3, FAST>(g, pg, qg, pdesc, qdesc, qfl, wave, first, t_begin, t_end, smem, pa, qa, h);
    const float* ot = reinterpret_cast<const float*>(smem);
    float* dst = g.out + (long long)split * g.out_split_stride + ((long long)m0 * g.Nn + n0) * 9;
    for (int i = tid; i < 64 * 144; i += 512) {
        const int row = i / 144, c4 = i - row * 144;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * 576 + c4 * 4);
        *reinterpret_cast<f32x4*>(dst + (long long)row * g.Nn * 9 + c4 * 4) = v;
    }
}

// ---------------------------------------------------------------------------
struct Wgrad2Plan { int tws, tiles_x, tiles_y, ntiles, splits, tiles_per_split, nmt, nnt; };

static Wgrad2Plan wgrad2_plan(int N, int H, int W, int M, int Nn) {
    Wgrad2Plan p;
    p.tws = W >= 32 ? 5 : 4;
    const int TW = 1 << p.tws, TH = 256 / TW;
    p.tiles_x = (W + TW - 1) / TW; p.tiles_y = (H + TH - 1) / TH;
    p.ntiles = N * p.tiles_x * p.tiles_y;
    p.nmt = M / 64; p.nnt = Nn / 64;
    int want = 256 / (p.nmt * p.nnt);                // one workgroup per CU: fewest slabs to write and re-read
    if (want < 1) want = 1;
    want *= fea_wg_split();                          // (x fea_set_persistent_split when CUs may be taken away)
    if (want > p.ntiles) want = p.ntiles;
    p.tiles_per_split = (p.ntiles + want - 1) / want;
    p.splits = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
    return p;
}

size_t fea_wgrad2_ws_bytes(int N, int H, int W, int Cm, int Cn) {
    Wgrad2Plan p = wgrad2_plan(N, H, W, Cm, Cn);
    return p.splits > 1 ? (size_t)p.splits * 9 * Cm * Cn * sizeof(float) : 0;
}

// bf16 conv3x3 wgrad; arguments already validated by the caller (wgrad.hip)
int fea_wgrad2_launch(const void* dy, int lddy, const void* x, int ldx, float* dW, void* ws, size_t ws_bytes,
                      int N, int H, int W, int Cin, int Cout, hipStream_t s) {
    const int M = Cout, Nn = Cin;
    Wgrad2Plan p = wgrad2_plan(N, H, W, M, Nn);
    const size_t need = p.splits > 1 ? (size_t)p.splits * 9 * M * Nn * sizeof(float) : 0;
    FEA_CHECK(ws_bytes >= need, FEA_E_WORKSPACE, "fea_conv3x3_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    FEA_CHECK(((uintptr_t)dW % 16 == 0) && ((uintptr_t)ws % 16 == 0), FEA_E_BADARG, "fea_conv3x3_wgrad: dW/ws must be 16-byte aligned");
    Wgrad2Args a{};
    a.p = (const __bf16*)dy; a.q = (const __bf16*)x;
    a.out = p.splits > 1 ? (float*)ws : dW;
    a.out_split_stride = (long long)9 * M * Nn;
    a.p_img_stride = (long long)H * W * lddy; a.q_img_stride = (long long)H * W * ldx;
    a.ldp = lddy; a.ldq = ldx; a.H = H; a.W = W; a.M = M; a.Nn = Nn;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split;
    a.nmt = p.nmt; a.nnt = p.nnt;
    // wgrad4_kernel's tile walk: consecutive tiles per split.  FEA_AMD_WG_WALK=1 interleaves them (split s: tiles s, s + splits, ...):
    // what gave the HBM-bound STORE streams of conv_p.hip 10 % does nothing for this kernel's reads (72.69 / 73.05 / 73.14 ms per step
    // consecutive, 72.82 / 73.08 / 73.04 interleaved, same box).
    static const bool inter = getenv("FEA_AMD_WG_WALK") && getenv("FEA_AMD_WG_WALK")[0] == '1';
    a.wstride = !inter || p.tiles_per_split == 1 ? 1 : p.splits;
    a.wsx = a.wstride % p.tiles_x; a.wsy = (a.wstride / p.tiles_x) % p.tiles_y; a.wsn = a.wstride / (p.tiles_x * p.tiles_y);
#ifdef WG2_STAMPS
    a.dbg = getenv("WG2_DBG_PTR") ? (unsigned long long*)strtoull(getenv("WG2_DBG_PTR"), nullptr, 0) : nullptr;
#endif
    dim3 grid((unsigned)(p.nmt * p.nnt * p.splits)), block(512);
    // whole tiles and 32-bit in-image byte offsets: the LDS-DMA pieces take the buffer-descriptor form (wg2_issue_piece)
    const int TWl = 1 << p.tws, THl = 256 / TWl;
    const bool fast = H % THl == 0 && W % TWl == 0 && (long long)(H + 2) * W * lddy * 2 < (1ll << 31) && (long long)(H + 2) * W * ldx * 2 < (1ll << 31);
    // FEA_AMD_WG3=1 selects the 16x16x32 form of the 8x32-tile kernel (wgrad3_kernel).  Correct on every weight-gradient test.  Measured
    // round 3, same box: with register-tied waits (a hazard s_nop per MFMA) 4-7 % SLOWER per launch than wgrad2 (the loop holds twice
    // the MFMA issues beside the same number of 8-byte transposing LDS reads: ~4,750 issue cycles per 4,608 matrix-pipe cycles per
    // SIMD and tile); with untied waits and per-tile piece invariants 0-6 % FASTER on random data (1024->1024 @32^2: 0.93 -> 0.88 ms)
    // and equal inside the training step (76.0 vs 76.1 ms).  Off by default.
    static const bool wg3 = getenv("FEA_AMD_WG3") && getenv("FEA_AMD_WG3")[0] == '1';
    // wgrad4_kernel (16x16x32, re-tiled) is the default on whole 8x32 tiles: 6-8 % faster per launch than wgrad2 on random data
    // (1024->1024 @32^2: 0.90 -> 0.84 ms, 1,480 TF/s), step 76.7 -> 75.35 ms same-box.  FEA_AMD_WG4=0 selects wgrad2 (A/B runs).
    static const bool wg4 = !(getenv("FEA_AMD_WG4") && getenv("FEA_AMD_WG4")[0] == '0');
    if (p.tws == 5 && wg4 && fast) hipLaunchKernelGGL((wgrad4_kernel<true>), grid, block, 0, s, a);      // (whole tiles only: the general pieces spill there)
    else if (p.tws == 5 && wg3) { if (fast) hipLaunchKernelGGL((wgrad3_kernel<true>), grid, block, 0, s, a); else hipLaunchKernelGGL((wgrad3_kernel<false>), grid, block, 0, s, a); }
    else if (p.tws == 5) { if (fast) hipLaunchKernelGGL((wgrad2_kernel<5, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((wgrad2_kernel<5, false>), grid, block, 0, s, a); }
    else            { if (fast) hipLaunchKernelGGL((wgrad2_kernel<4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((wgrad2_kernel<4, false>), grid, block, 0, s, a); }
    FEA_LAUNCH_CHECK();
    if (p.splits > 1) {
        const long long per4 = (long long)9 * M * Nn / 4;
        fea_launch_slab_reduce((const float*)ws, dW, p.splits, per4, s);
        FEA_LAUNCH_CHECK();
    }
    return FEA_OK;
}
