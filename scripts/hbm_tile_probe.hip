// Write-only HBM rate of TILE-shaped store streams (the shape of the conv kernels' outputs at level 0: NHWC bf16, 64 channels =
// 128 B per pixel, 512 x 512 images): R rows x S contiguous bytes per tile, row pitch 64 KB, 64 images = 2 GiB; tile walk
// interleaved (tile = wg + k * nwg) or chunked (wg owns consecutive tiles); wave-instruction footprint 1 KB contiguous or
// 16 pixels x 64-byte halves (an accumulator's direct stores).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/hbm_tile_probe.bin scripts/hbm_tile_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct TArgs { char* dst; int R, S, tiles_x, tiles_y, ntiles, chunked, half, xcd; long long pitch, img, bytes; int* bad; };

__global__ __launch_bounds__(256) void tile_fill_k(const TArgs g) {
    const u32x4 val = {1u, 2u, 3u, 4u};
    int wg = blockIdx.x;
    const int nwg = gridDim.x;
    if (g.xcd) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);          // consecutive logical workgroups on one XCD
    const int per = (g.ntiles + nwg - 1) / nwg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // every store is bounds-checked against the buffer (a wrong index raises the flag instead of faulting)
    auto put = [&](char* p) { if (p >= g.dst && p + 16 <= g.dst + g.bytes) *reinterpret_cast<u32x4*>(p) = val; else *g.bad = 1; };
    for (int k = 0; k < per; ++k) {
        const int t = g.chunked ? wg * per + k : wg + k * nwg;
        if (t >= g.ntiles) break;
        const int tx = t % g.tiles_x, ty = (t / g.tiles_x) % g.tiles_y, n = t / (g.tiles_x * g.tiles_y);
        char* base = g.dst + n * g.img + (long long)ty * g.R * g.pitch + (long long)tx * g.S;
        if (!g.half) {
            for (int r = 0; r < g.R; ++r)
                for (int off = tid * 16; off < g.S; off += 4096) put(base + r * g.pitch + off);
        } else {
            // a unit = 16 pixels = 2 KB of one row, written by one wave as two instructions of 64-byte halves
            const int upr = g.S / 2048, units = g.R * upr;
            for (int u = wave; u < units; u += 4) {
                // half 1: four consecutive lanes write one pixel's 64 bytes; half 2: the direct-store epilogue's lanes (conv_p.hip,
                // first_conv.hip): consecutive lanes are consecutive PIXELS, the four 16-byte pieces of a half sit 16 lanes apart
                char* p = base + (u / upr) * g.pitch + (u % upr) * 2048;
                if (g.half == 1)      { p += (lane >> 2) * 128 + (lane & 3) * 16; put(p); put(p + 64); }
                else if (g.half == 2) { p += (lane & 15) * 128 + (lane >> 4) * 16; put(p); put(p + 64); }
                else {
                    // half 3: each instruction covers 8 whole pixels (1 KB contiguous), lanes permuted as a row_ror:8 exchange of the
                    // direct-store epilogue's two halves would leave them: lane (r16, q) -> pixel r16 & 7, half r16 >> 3, chunk q
                    p += (lane & 7) * 128 + ((lane >> 3) & 1) * 64 + (lane >> 4) * 16; put(p); put(p + 1024);
                }
            }
        }
    }
}

int main() {
    const long long pitch = 512 * 128, img = 512 * pitch, bytes = 64 * img;
    char* a; CK(hipMalloc(&a, bytes)); CK(hipMemset(a, 1, bytes));
    int* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int shapes[][2] = {{8, 4096}};
    for (auto& sh : shapes)
        for (int nwg : {768})
            for (int chunked = 0; chunked < 2; ++chunked)
                for (int half = 0; half < 4; ++half)
                    for (int xcd = 0; xcd < 2; ++xcd) {
                        if (xcd && !chunked) continue;
                        TArgs g{a, sh[0], sh[1], (int)(pitch / sh[1]), 512 / sh[0], 0, chunked, half, xcd, pitch, img, bytes, bad};
                        if (sh[1] % 2048 || pitch % sh[1] || 512 % sh[0]) { printf("bad shape\n"); return 1; }
                        g.ntiles = 64 * g.tiles_x * g.tiles_y;
                        hipLaunchKernelGGL(tile_fill_k, dim3(nwg), dim3(256), 0, 0, g);
                        CK(hipDeviceSynchronize());
                        CK(hipEventRecord(e0));
                        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(tile_fill_k, dim3(nwg), dim3(256), 0, 0, g);
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
                        printf("tile %2d x %5d B  nwg %4d  %-11s %-12s %-4s %7.3f ms  %5.2f TB/s\n", sh[0], sh[1], nwg,
                               chunked ? "chunked" : "interleaved", half == 3 ? "1KB-permuted" : half == 2 ? "16B-lanes" : half ? "64B-halves" : "1KB", xcd ? "xcd" : "", ms, bytes / ms / 1e9);
                    }
    int hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("out-of-range stores suppressed: %s\n", hb ? "YES (index bug)" : "none");
    return 0;
}
