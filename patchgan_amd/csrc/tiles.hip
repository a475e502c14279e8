// tiles.hip -- tiled inference around the generator forward (reference patchgan/infer.py:14-68): cut an image into
// overlapping size x size tiles straight into the NHWC batch the generator kernels read (n_crop), and overlap-average the
// predicted tiles back into the image frame with the optional threshold and the class argmax (build_mask).  HBM-bound:
// 8 B per gathered element; the blend reads 4 B per tile element and writes 8 B per mask element.
//
// Tile k along an axis of `extent` pixels starts at  s = k*eff - max(k*eff + size - extent, 0),  eff = int(overlap*size),
// k < ceil(extent / eff)  (infer.py:20-31).  Tiles are numbered row-major, j * nx + i (the reference's j*ncropsy + i is
// the same number for the square images it supports).  The blend adds the tiles covering a pixel in that order into a
// double, divides by the cover count, compares >= threshold in double (threshold > 0) -- the reference's arithmetic and
// order, so the result is bit-identical to it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "patchgan_hip.h"
#include "pg_common.h"

namespace {

__device__ __forceinline__ int tile_start(int k, int eff, int size, int extent) {
    const int s = k * eff, over = s + size - extent;
    return over > 0 ? s - over : s;
}

__global__ void k_tiles_gather(const float* __restrict__ img, int C, int H, int W, int size, int eff, int ny, int nx,
                               float* __restrict__ tiles, int ld) {
    const long total = (long)ny * nx * size * size * C;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        long r = idx / C;
        const int x = (int)(r % size);
        r /= size;
        const int y = (int)(r % size);
        const int t = (int)(r / size);
        const int j = t / nx, i = t - j * nx;
        const int sy = tile_start(j, eff, size, H), sx = tile_start(i, eff, size, W);
        tiles[(((long)t * size + y) * size + x) * ld + c] = img[((long)c * H + sy + y) * W + sx + x];
    }
}

__global__ void k_tiles_blend(const float* __restrict__ tiles, int ld, int C, int size, int eff, int ny, int nx, int H,
                              int W, double thr, double* __restrict__ mask, long long* __restrict__ amax) {
    const long total = (long)H * W;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int h = (int)(idx / W), w = (int)(idx - (long)h * W);
        double best = 0.0;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            double acc = 0.0, cnt = 0.0;
            for (int j = 0; j < ny; ++j) {
                const int sy = tile_start(j, eff, size, H);
                if (h < sy || h >= sy + size) continue;
                for (int i = 0; i < nx; ++i) {
                    const int sx = tile_start(i, eff, size, W);
                    if (w < sx || w >= sx + size) continue;
                    acc += (double)tiles[(((long)(j * nx + i) * size + (h - sy)) * size + (w - sx)) * ld + c];
                    cnt += 1.0;
                }
            }
            double v = acc / cnt;
            if (thr > 0.0) v = (v >= thr) ? 1.0 : 0.0;
            if (mask) mask[((long)c * H + h) * W + w] = v;
            if (c == 0 || v > best) {      // first maximum, like numpy.argmax
                best = v;
                bi = c;
            }
        }
        if (amax) amax[idx] = bi;
    }
}

int grid_for(long total) {
    long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int pg_tiles_count(int extent, int size, int eff) {
    if (extent < size || size <= 0 || eff <= 0) return 0;
    return (extent + eff - 1) / eff;
}

int pg_tiles_gather(const float* image, int C, int H, int W, int size, int eff, float* tiles, int ld, void* stream) {
    const int ny = pg_tiles_count(H, size, eff), nx = pg_tiles_count(W, size, eff);
    if (!image || !tiles || C <= 0 || ld < C || ny <= 0 || nx <= 0) return PG_EINVAL;
    const long total = (long)ny * nx * size * size * C;
    hipLaunchKernelGGL(k_tiles_gather, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, image, C, H, W, size, eff, ny,
                       nx, tiles, ld);
    return pg_launch_status();
}

int pg_tiles_blend(const float* tiles, int ld, int C, int size, int eff, int H, int W, double threshold, double* mask,
                   long long* argmax, void* stream) {
    const int ny = pg_tiles_count(H, size, eff), nx = pg_tiles_count(W, size, eff);
    if (!tiles || (!mask && !argmax) || C <= 0 || ld < C || ny <= 0 || nx <= 0) return PG_EINVAL;
    hipLaunchKernelGGL(k_tiles_blend, dim3(grid_for((long)H * W)), dim3(256), 0, (hipStream_t)stream, tiles, ld, C, size, eff,
                       ny, nx, H, W, threshold, mask, argmax);
    return pg_launch_status();
}

}  // extern "C"
