// Store-pattern probe: how much does the conv epilogue's "lane = pixel, 16 bytes per lane, 8 instructions per 128-byte pixel row" cost
// against whole-row stores?  Each wave writes 4 KiB blocks (32 pixels x 128 B) of a 128 MiB buffer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_pattern_probe tools/store_pattern_probe.hip && tools/store_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(256) void k_store(u32x4* out, long nblocks, int pitch16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long b = (long)blockIdx.x * 4 + wave; b < nblocks; b += (long)gridDim.x * 4) {
        u32x4* blk = out + b * 256;                                   // 4 KiB = 256 x 16 B
        const u32x4 v = {(unsigned)b, (unsigned)lane, 1u, 2u};
        if (PAT == 0) {            // epilogue pattern: pixel = lane & 31, piece = 2 p + (lane >> 5); pixel pitch = pitch16 * 16 B
#pragma unroll
            for (int p = 0; p < 4; ++p) blk[(lane & 31) * pitch16 + 2 * p + (lane >> 5)] = v;
        } else if (PAT == 1) {     // whole rows: 8 lanes per pixel row
#pragma unroll
            for (int p = 0; p < 4; ++p) blk[(p * 8 + (lane >> 3)) * pitch16 + (lane & 7)] = v;
        }
    }
}
int main() {
    const long bytes = 128l << 20, nblocks = bytes / 4096;
    u32x4* d;
    hipMalloc(&d, bytes * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pitch16 : {8, 16}) {
        for (int pat = 0; pat < 2; ++pat) {
            for (int grid : {1024, 4096}) {
                float best = 1e9;
                for (int r = 0; r < 6; ++r) {
                    hipEventRecord(e0);
                    const long nb = (pitch16 == 8) ? nblocks : nblocks;   // pitch 16: the rows of two classes interleave (256-B pitch, 128 B written)
                    if (pat == 0) hipLaunchKernelGGL(k_store<0>, dim3(grid), dim3(256), 0, 0, d, nb, pitch16);
                    else hipLaunchKernelGGL(k_store<1>, dim3(grid), dim3(256), 0, 0, d, nb, pitch16);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (r && ms < best) best = ms;
                }
                printf("pitch %3d B pattern %d grid %5d: %7.1f us  %6.2f TB/s\n", pitch16 * 16, pat, grid, best * 1e3, bytes / best / 1e9);
            }
        }
    }
    return 0;
}
