// Persistent, software-pipelined GEMM for the transposed-convolution forward (bf16, gfx950).
//
// nn.ConvTranspose2d(ci, co, 2, stride=2) (models.py:15-21) is a plain GEMM per pixel: M = pixels, K = ci, N = 4 co (column
// ab*co + c goes to output pixel (2i+a, 2j+b), channel c).  The implicit-GEMM kernel of igemm.hip runs it as 256 x 128 tiles
// whose every 64-channel chunk is staged between two barriers (LDS-DMA issue + its whole memory latency exposed per chunk), and a
// plain GEMM tile of that size asks the CU's LDS fill path for 47 B/clk at full MFMA rate where ~13 (Infinity Cache) to ~30 (L2)
// are to be had (MI355X_MICROARCH.md, gather into LDS): 350-940 TF/s.  Here (the data gradient, the same GEMM over gathered rows
// with K = 4 co, stays on igemm.hip: its fused BatchNorm-backward epilogue does not fit beside 128 accumulators per wave):
//
//   * ONE workgroup of 8 waves per CU walks a range of 256-pixel tiles for a fixed 256-column group: wave = 128 pixels x
//     64 columns (8 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16, 128 registers), two waves per SIMD so that one wave's
//     LDS-DMA issue stalls are covered by its partner's MFMAs;
//   * operands arrive by LDS-DMA in chunks of 64 channels = whole 128-byte lines of every row (with 32-channel stages every line
//     was fetched twice from L2, one half per k-step: 2,800 cycles per k-step instead of 1,024, the CU's L2 fill rate of ~30 B/clk
//     being the limit of a plain GEMM: 32 KB per k-step and CU); two stages of 64 KB (256 pixel rows + 256 weight rows), the
//     pieces of chunk c + 2 issued behind the barrier in the second k-step of chunk c; ONE barrier per chunk, and the ring runs on
//     across tiles (the next tile's first chunks load during the epilogue);
//   * fragments are read one k-step ahead (pixel fragments re-read right behind their four MFMAs), waits are counted;
//   * the MFMA takes the WEIGHT fragment as its A operand: an accumulator register holds 4 consecutive channels of one pixel,
//     one v_permlane16_swap per dword gives a lane 8 consecutive channels = one 16-byte store, no LDS staging in the epilogue;
//   * the workgroups that share a pixel-tile range (one per column group) are blocks b, b + 8, ...: one XCD, so a pixel tile is
//     read from HBM once and from that XCD's L2 by the others.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

struct UpgArgs {
    const __bf16* a;          // pixel operand x [M][lda]
    const __bf16* w;          // packed weights [K / 64][Ntot][64] (fea_pack_upconv2x2)
    const float* bias;        // [Cup], nullable
    __bf16* y;                // [4 M][ldy], pixel-shuffled
    int lda, ldy;
    unsigned a_bytes, w_bytes;
    int ntiles;               // M / 256
    int NC;                   // 64-channel chunks per tile (even)
    int Ntot, ntn;            // GEMM N, Ntot / 256
    int Cup;                  // channels per output quadrant
    int W;                    // input image width (power of two >= 16)
    int tiles_per_walk, tstride;
    unsigned long long* dbg;  // U_STAMPS builds only: shader cycles of every workgroup
};

constexpr int U_REG = 32768;                   // one stage of one operand: 256 rows x 128 B
constexpr int U_NST = 16;                      // 16-byte stores per wave and tile
#ifndef U_NT_ST
#define U_NT_ST 0
#endif
#ifndef U_PREFETCH
#define U_PREFETCH 0       // 1: one discarded piece per wave and chunk pulls the next tile's first pixel rows into L2 (measured: no gain, see below)
#endif

// XOR of a 128-byte row's 16-byte chunk index (igemm.hip swz: conflict-free ds_read_b128 for every base row)
__device__ __forceinline__ int u_swz(int row) { return ((row >> 1) & 3) << 1; }

// One 1 KB LDS-DMA piece as a buffer load (a free function: see ig_buffer_lds16 in igemm.hip).  nbytes = 0: no lane reads memory, zeros land.
#ifndef U_AUX_A
#define U_AUX_A 0        // cache policy of the pixel pieces (2 = nt) -- A/B builds
#endif
#ifndef U_AUX_W
#define U_AUX_W 0        // ... of the weight pieces
#endif
template <int AUX>
__device__ __forceinline__ void upg_piece(const void* base, unsigned nbytes, char* lds, int voff, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds, 16, voff, soff, 0, AUX);
}

__device__ __forceinline__ uint32_t upg_pack(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__global__ __launch_bounds__(512, 2) void upg_kernel(const UpgArgs g) {
    // [pixel rows stage 0 | stage 1 | weight rows stage 0 | stage 1], 256 rows x 128 B each, then the bias of the column group
    __shared__ __attribute__((aligned(1024))) char smem[4 * U_REG + 2048];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    const int r16 = lane & 15, q16 = lane >> 4;
    // blocks b, b + 8, ... (one XCD) take the column groups of one tile range
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int nt = slot % g.ntn;
    const int walk = (slot / g.ntn) * 8 + xcd;
    // tile walk: tstride == 1: tiles walk * tiles_per_walk ..., consecutive; tstride == number of walkers: tiles walk, walk + tstride,
    // ... (at any moment the chip writes one compact window of output; conv_p.hip, scripts/hbm_tile_probe.hip).  t1 bounds the cursor.
    const int tstride = g.tstride;
    const int t0 = tstride == 1 ? walk * g.tiles_per_walk : walk;
    if (t0 >= g.ntiles) return;
    const int t1 = tstride == 1 ? min(t0 + g.tiles_per_walk, g.ntiles) : t0 + (g.ntiles - t0 + tstride - 1) / tstride * tstride;
    const int NC = g.NC;
#ifdef U_STAMPS
    const unsigned long long tstart = __builtin_amdgcn_s_memtime();
#endif

    // ---- fragment read addresses: pixel row wm*128 + m*16 + r16, weight row wn*64 + n*16 + r16, chunk q16 (+4 in k-step 1) ----
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t fa0 = lds0 + (wm * 128 + r16) * 128 + ((q16 ^ u_swz(r16)) << 4);
    const uint32_t fb0 = lds0 + 2 * U_REG + (wn * 64 + r16) * 128 + ((q16 ^ u_swz(r16)) << 4);
    const uint32_t fa1 = fa0 ^ 64u, fb1 = fb0 ^ 64u;

    // ---- LDS-DMA pieces: piece j fills rows 8 j .. 8 j + 7 (lane -> row 8 j + lane/8, slot lane%8, source chunk slot ^ swz(row));
    //      this wave issues pixel pieces {wave + 8 i} and weight pieces {wave + 8 i}, i = 0..3 (rows + 64 i), of every stage --------
    const int prow = 8 * wave + (lane >> 3);
    const int pchunk = (lane & 7) ^ u_swz(prow);
    int voffA, dA[4];                        // lane offset of the first pixel piece; wave-uniform offsets of the others
    voffA = prow * g.lda * 2 + pchunk * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) dA[i] = 64 * i * g.lda * 2;
    const int voffB = prow * 128 + pchunk * 16;
    const __bf16* const wbase = g.w + (long long)nt * 256 * 64;
    const unsigned w_bytes = g.w_bytes - (unsigned)nt * 256u * 128u;
    char* const dstA = smem + wave * 1024;
    char* const dstB = smem + 2 * U_REG + wave * 1024;

    // The column groups of one tile range start their K loop NC / ntn chunks apart (the K order is free as long as both operands
    // agree; U_ROT 2).  Run in step, the eight workgroups that share a pixel tile ask for the same lines at the same moment, all
    // behind the one HBM miss, and every workgroup of the chip reads the same 128-byte column of its rows: 0.279 / 0.355 / 0.466 ms
    // on the three deep levels in step, 0.261 / 0.324 / 0.432 apart.  (U_ROT 1 staggers the walkers of an XCD instead -- meant to
    // keep the weight lines in L2 -- and measured 0.290 / 0.359 / 0.466; both, U_ROT 3: as 2.)
    const int jwalk = slot / g.ntn, J = max(1, (int)(gridDim.x >> 3) / g.ntn);
#ifndef U_ROT
#define U_ROT 2
#endif
    const int rot = U_ROT == 1 ? (jwalk * NC / J) % NC : U_ROT == 2 ? (nt * NC / g.ntn) % NC : U_ROT == 3 ? (nt * NC / g.ntn + jwalk) % NC : 0;
    // producer cursor: the chunk whose pieces are issued next
    int ptile = t0, pc = 0;
    unsigned pnbA = 0, pnbW = 0;
    int psoA = 0, psoB = 0;
    auto produce_prep = [&]() {
        const bool live = ptile < t1;
        pnbA = live ? g.a_bytes : 0u; pnbW = live ? w_bytes : 0u;
        int pcr = pc + rot; if (pcr >= NC) pcr -= NC;
        psoA = (int)((unsigned)ptile * 256u * (unsigned)g.lda * 2u) + pcr * 128;
        psoB = pcr * g.Ntot * 128;
#ifdef U_EXP_HOT_DMA      // EXPERIMENT (wrong results): every chunk re-reads the first 64 KB = issue + LDS-write cost without the memory system
        psoA = 0; psoB = 0;
#endif
        if (++pc == NC) { pc = 0; ptile += tstride; }
    };
    auto produce_piece = [&](auto stc, auto ic) {
        constexpr int STG = decltype(stc)::value, i = decltype(ic)::value;
#ifdef U_EXP_NO_DMA       // EXPERIMENT (wrong results): no pieces at all = the MFMA + fragment-read rate of the loop
        return;
#endif
        if constexpr (i < 4) upg_piece<U_AUX_A>(g.a, pnbA, dstA + STG * U_REG + i * 8192, voffA, psoA + dA[i]);
        else                 upg_piece<U_AUX_W>(wbase, pnbW, dstB + STG * U_REG + (i - 4) * 8192, voffB, psoB + (i - 4) * 8192);
    };

    // L2 prefetch of the NEXT tile's pixel rows (U_PREFETCH, off: 0.264 / 0.337 / 0.454 ms with it against 0.263 / 0.327 / 0.430
    // without on the three deep levels -- the first-reader misses are not what the ring waits for).  The column groups of a tile range run their K loops NC / ntn chunks apart (rot), so
    // each is the first reader of its first few chunks of every tile and finds the rest in L2; those first reads are HBM misses that
    // the two-stage ring (one chunk = ~1 us of lookahead) cannot cover.  One discarded 1 KB piece per wave and chunk, a whole tile
    // time ahead, lands them in L2 first: piece q = it * 8 + wave of the next tile's chunks rot, rot + 1, ... (32 pieces per chunk).
    char* const dump = smem + 4 * U_REG + 1024;
    auto prefetch = [&](int tile, int it) {
#if U_PREFETCH
        const bool live = tile + tstride < t1 && g.ntn > 1;
        int pcr = rot + (it >> 2); if (pcr >= NC) pcr -= NC;
        int so;
        so = (int)((unsigned)(tile + tstride) * 256u * (unsigned)g.lda * 2u) + pcr * 128;
        upg_piece<0>(g.a, live ? g.a_bytes : 0u, dump, voffA, so + dA[it & 3]);
#endif
    };

    // bias of the 256-column group in LDS (behind the stages): added in the epilogue, where the dead weight fragments' registers
    // hold it (as persistent registers, or as the C operand of a tile's first MFMAs, it spilled the kernel)
    float* const blds = reinterpret_cast<float*>(smem + 4 * U_REG);
    if (tid < 256) blds[tid] = g.bias ? g.bias[(nt * 256 + tid) % g.Cup] : 0.f;
    __syncthreads();           // (with its fence, before the first LDS-DMA piece: the bare s_barrier of the loop orders nothing for the compiler)
    // output: lane = pixel r16 of accumulator tile m, 8 channels from chl of the 32-column pair j
    const int chl = (q16 & 1) * 16 + (q16 >> 1) * 8;
    int lane_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col0 = nt * 256 + wn * 64 + j * 32;
        const int ab = col0 / g.Cup, co0 = col0 - ab * g.Cup;
        lane_off[j] = (2 * r16 + (ab >> 1) * 2 * g.W + (ab & 1)) * g.ldy + co0 + chl;
    }

    f32x4 acc[8][4];
    u32x4 A[8], Bw[2][4];
    auto rd = [&](u32x4& dst, uint32_t addr, auto offc) {
#ifdef U_EXP_NO_READS     // EXPERIMENT (wrong results): no fragment reads = the MFMA issue rate of the loop
        return;
#endif
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(offc)::value) : "memory");
    };
    auto rdA = [&](auto stc, auto ksc, auto mc) {
        constexpr int STG = decltype(stc)::value, m = decltype(mc)::value;
        rd(A[m], decltype(ksc)::value ? fa1 : fa0, std::integral_constant<int, STG * U_REG + m * 2048>{});
    };
    auto rdB = [&](auto stc, auto ksc, auto nc) {
        constexpr int STG = decltype(stc)::value, n = decltype(nc)::value, ks = decltype(ksc)::value;
        rd(Bw[ks][n], ks ? fb1 : fb0, std::integral_constant<int, STG * U_REG + n * 2048>{});
    };

    // One k-step (32 channels) of the chunk in stage STG.  On entry its 12 fragments are in flight, issued in the order
    // B0..B3, A0, A1, A2..A7 during the previous k-step.  Slot m = the four MFMAs of pixel tile m.  In front of slot 2 every read of
    // this k-step has landed; in k-step 1 that means the wave is done reading the stage, and once its pieces of the NEXT chunk have
    // landed too -> barrier -> the other stage may be read and this one overwritten (the 8 pieces of chunk +2, behind slots 2..7).
    // Ring refill behind the barrier: B, A0, A1, then Am behind slot m.
    // VM operations behind the pieces that must have landed: none, or the U_NST stores of the epilogue (`after_epi`: chunk 0 of
    // a tile that follows one).
    auto kstep = [&](auto stc, auto ksc, auto c0c, bool after_epi, int tile, int it) {
        constexpr int STG = decltype(stc)::value, KSI = decltype(ksc)::value;
        constexpr int NSTG = KSI ? STG ^ 1 : STG, NKS = KSI ^ 1;              // where the next k-step's fragments are
        constexpr bool C0 = decltype(c0c)::value, FIRST = C0 && KSI == 0;
        sfor<0, 8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            if constexpr (m == 0) { asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (m == 1) { asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (m == 2) {
                if constexpr (KSI == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else {
                    // (the newest VM operation is the prefetch piece of the previous chunk: it need not have landed)
                    if (C0 && after_epi) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(U_NST + U_PREFETCH) : "memory");
                    else                 asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(U_PREFETCH) : "memory");
#ifndef U_EXP_NO_BARRIER  // EXPERIMENT (racy)
                    __builtin_amdgcn_s_barrier();
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                sfor<0, 4>([&](auto nc) { rdB(std::integral_constant<int, NSTG>{}, std::integral_constant<int, NKS>{}, nc); });
                rdA(std::integral_constant<int, NSTG>{}, std::integral_constant<int, NKS>{}, std::integral_constant<int, 0>{});
                rdA(std::integral_constant<int, NSTG>{}, std::integral_constant<int, NKS>{}, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (KSI == 1) produce_prep();
            }
            // MFMAs as inline asm, accumulating in place: as builtins hipcc renames the accumulators MFMA by MFMA (destination !=
            // C operand) and spills 20-60 of them.  No software hazard applies inside the loop: an accumulator is touched again 32
            // MFMAs later, fragments are only written by LDS reads behind counted waits; the epilogue pads its first accumulator read.
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc[m][n]) : "v"(Bw[KSI][n]), "v"(A[m]));
                else                 asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m][n]) : "v"(Bw[KSI][n]), "v"(A[m]));
            }
            if constexpr (m >= 2) rdA(std::integral_constant<int, NSTG>{}, std::integral_constant<int, NKS>{}, mc);
            if constexpr (KSI == 1 && m >= 2) {
                produce_piece(stc, std::integral_constant<int, m - 2>{});
                if constexpr (m < 4) produce_piece(stc, std::integral_constant<int, m + 4>{});
                if constexpr (m == 7) prefetch(tile, it);
            }
            // (slots are kept apart: with the MFMAs of several slots interleaved hipcc stops accumulating in place and spills accumulators)
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto chunk_pair = [&](auto c0c, bool after_epi, int tile, int it) {
        kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, c0c, after_epi, tile, it);
        kstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, c0c, after_epi, tile, it);
        kstep(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::false_type{}, false, tile, it + 1);
        kstep(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::false_type{}, false, tile, it + 1);
    };

    auto epilogue = [&](int tile) {
        const int pixw = tile * 256 + wm * 128;
        // (the last MFMAs are inline asm: the compiler does not pad the accumulators' first VALU read -- 16 passes at most)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
        f32x4 bv[4];
        {
            // (the address is rebuilt from the lane id here: as a loop invariant it was the one register the kernel spilled, and its
            //  scratch reload is a vmcnt(0) in every tile)
            int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(ln));
            const uint32_t bias_addr = lds0 + 4 * U_REG + (wn * 64 + 4 * (ln >> 4)) * 4;
#pragma unroll
            for (int n = 0; n < 4; ++n)     // (inline asm: a compiler-visible LDS read makes hipcc drain the LDS-DMA pieces in flight, vmcnt(0))
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[n]) : "v"(bias_addr + n * 64), "n"(0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]) :: "memory");
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int pix0 = pixw + m * 16;
            const long long sbase = (long long)(4 * pix0 - 2 * (pix0 & (g.W - 1))) * g.ldy;      // output pixel (2i, 2j) of the tile row
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[m][2 * j] += bv[2 * j]; acc[m][2 * j + 1] += bv[2 * j + 1];
                const uint32_t lo0 = upg_pack(acc[m][2 * j][0], acc[m][2 * j][1]), hi0 = upg_pack(acc[m][2 * j][2], acc[m][2 * j][3]);
                const uint32_t lo1 = upg_pack(acc[m][2 * j + 1][0], acc[m][2 * j + 1][1]), hi1 = upg_pack(acc[m][2 * j + 1][2], acc[m][2 * j + 1][3]);
                // odd 16-lane rows of the first operand <-> even rows of the second: every lane ends up with 8 consecutive channels
                const auto r0 = __builtin_amdgcn_permlane16_swap(lo0, lo1, false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(hi0, hi1, false, false);
                const u32x4 out = {r0[0], r1[0], r0[1], r1[1]};
#if U_NT_ST      // A/B: nontemporal output stores
                __builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(g.y + sbase + lane_off[j]));
#else
                *reinterpret_cast<u32x4*>(g.y + sbase + lane_off[j]) = out;
#endif
            }
        }
    };

    // ---- prologue: the first two chunks' pieces, then the fragments of k-step 0 ---------------------------------------
    sfor<0, 2>([&](auto stc) {
        produce_prep();
        sfor<0, 8>([&](auto ic) { produce_piece(stc, ic); });
    });
    prefetch(t1, 0);                  // (a dead piece: keeps the barrier waits' "one prefetch piece behind the pieces" true in chunk 0)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + U_PREFETCH) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    sfor<0, 4>([&](auto nc) { rdB(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, nc); });
    sfor<0, 8>([&](auto mc) { rdA(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, mc); });

    const int npairs = NC >> 1;
    for (int tile = t0; tile < t1; tile += tstride) {
        chunk_pair(std::true_type{}, tile != t0, tile, 0);
        for (int gi = 1; gi < npairs; ++gi) chunk_pair(std::false_type{}, false, tile, 2 * gi);
        epilogue(tile);
    }
    // the trailing (dead) pieces write zeros into LDS: they must have landed before the workgroup's LDS is handed on
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef U_STAMPS
    if (tid == 0 && g.dbg) g.dbg[blockIdx.x] = __builtin_amdgcn_s_memtime() - tstart;
#endif
}

// ---------------------------------------------------------------------------
static int upg_walkers(int ntn) {
    int want = 256 / ntn * fea_wg_split();
    if (const char* e = getenv("FEA_AMD_UPG_WALKERS")) { const int v = atoi(e); if (v > 0) want = (v + 7) / 8 * 8; }   // tests: many tiles per workgroup on small shapes
    return want;
}

static bool upg_enabled() {
    const char* e = getenv("FEA_AMD_UPGEMM");
    return !(e && e[0] == '0');
}

// ConvTranspose2d forward on the persistent GEMM; false = shape not served (the caller falls back to igemm.hip)
bool fea_upg_fwd(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy, int N, int H, int W, int Cin, int Cout,
                 hipStream_t stream, int* rc) {
    const long long M = (long long)N * H * W;
    if (!upg_enabled()) return false;
    if (W < 16 || (W & (W - 1)) || M % 256 || Cin % 128 || (4 * Cout) % 256 || Cout % 32 || ldx % 8 || ldy % 8) return false;
    if (M * ldx * 2 >= (1ll << 32) || 4 * M * ldy * 2 >= (1ll << 40) || M >= (1ll << 28)) return false;
    UpgArgs a{};
    a.a = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y;
    a.lda = ldx; a.ldy = ldy;
    a.a_bytes = (unsigned)(M * ldx * 2); a.w_bytes = (unsigned)((long long)Cin * 4 * Cout * 2);
    a.ntiles = (int)(M / 256); a.NC = Cin / 64; a.Ntot = 4 * Cout; a.ntn = a.Ntot / 256; a.Cup = Cout; a.W = W;
    int nwalk = upg_walkers(a.ntn);
    if (nwalk < 8) nwalk = 8;
    a.tiles_per_walk = (a.ntiles + nwalk - 1) / nwalk;
    nwalk = ((a.ntiles + a.tiles_per_walk - 1) / a.tiles_per_walk + 7) / 8 * 8;
    // interleaved walk (FEA_AMD_UPG_WALK=0: consecutive tiles, A/B runs)
    static const bool consecutive = getenv("FEA_AMD_UPG_WALK") && getenv("FEA_AMD_UPG_WALK")[0] == '0';
    a.tstride = consecutive || a.tiles_per_walk == 1 ? 1 : nwalk;
#ifdef U_STAMPS
    a.dbg = getenv("UPG_DBG_PTR") ? (unsigned long long*)strtoull(getenv("UPG_DBG_PTR"), nullptr, 0) : nullptr;
#endif
    hipLaunchKernelGGL(upg_kernel, dim3((unsigned)(nwalk * a.ntn)), dim3(512), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fea_set_error("fea_upconv2x2_fwd (persistent GEMM): launch failed: %s", hipGetErrorString(e)); *rc = FEA_E_LAUNCH; }
    else *rc = FEA_OK;
    return true;
}
