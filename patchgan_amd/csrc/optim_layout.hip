// optim_layout.hip -- fused Adam over a flat parameter buffer, NCHW <-> NHWC at the API edge, fills/copies.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "patchgan_hip.h"
#include "pg_common.h"

namespace {

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_over_bc1, float beta1,
                                         float beta2, float eps, float sqrt_bc2) {
    // torch single-tensor Adam: exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, value=1-b2);
    // denom = sqrt(v)/sqrt(bc2) + eps; p.addcdiv_(m, denom, value=-lr/bc1)
    m = m + (g - m) * (1.f - beta1);
    v = v * beta2 + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) / sqrt_bc2 + eps;
    p = p - lr_over_bc1 * (m / denom);
}

// DEV: the two step-dependent scalars (lr / bc1, sqrt(bc2)) come from device memory -- the form a captured hipGraph replays with a
// new Adam step count (pg_adam_step_dev); otherwise they are kernel arguments.  Same arithmetic either way.
template <bool DEV>
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       long n, float lr_over_bc1, float beta1, float beta2, float eps, float sqrt_bc2, const float* __restrict__ scal) {
    if constexpr (DEV) {
        lr_over_bc1 = scal[0];
        sqrt_bc2 = scal[1];
    }
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, lr_over_bc1, beta1, beta2, eps, sqrt_bc2);
        adam_one(pp.y, gg.y, mm.y, vv.y, lr_over_bc1, beta1, beta2, eps, sqrt_bc2);
        adam_one(pp.z, gg.z, mm.z, vv.z, lr_over_bc1, beta1, beta2, eps, sqrt_bc2);
        adam_one(pp.w, gg.w, mm.w, vv.w, lr_over_bc1, beta1, beta2, eps, sqrt_bc2);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride)
        adam_one(p[i], g[i], m[i], v[i], lr_over_bc1, beta1, beta2, eps, sqrt_bc2);
}

// dst NHWC (pixel stride ld_dst) <- src NCHW; one thread per PIXEL: the C plane reads of a wave are C coalesced rows (one thread per
// (pixel, channel) read a different cache line per lane and paid three 64-bit divisions per element), the C floats of a pixel are
// written together
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, int ld_dst, int N, int C,
                               long HW) {
    const long npix = (long)N * HW;
    if (npix < 0x7fffffffL && C <= 8) {
        const unsigned np = (unsigned)npix, hwu = (unsigned)HW, stride = gridDim.x * blockDim.x;
        for (unsigned pix = blockIdx.x * blockDim.x + threadIdx.x; pix < np; pix += stride) {
            const unsigned n = pix / hwu, hw = pix - n * hwu;
            const float* sp = src + (size_t)n * C * HW + hw;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (c < C) ? sp[(size_t)c * HW] : 0.f;
            float* dp = dst + (size_t)pix * ld_dst;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < C) dp[c] = v[c];
        }
        return;
    }
    const long total = npix * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const long n = pix / HW, hw = pix - n * HW;
        dst[pix * ld_dst + c] = src[(n * C + c) * HW + hw];
    }
}

// The discriminator's input buffer din[2N][HW][ld] in ONE pass (trainer.py:65,96,98: cat((x, y), 1) and cat((x, G(x)), 1)): sample n of
// the real half = x[n] | y[n], of the fake half = x[n] | 0 (the generator's head writes its output there later in the step); channels
// Cx + Cy .. ld - 1 (the pad of an 8-float pixel) = 0.  One thread per pixel: Cx + Cy coalesced plane reads, then the two pixels as
// 16-byte stores where ld == 8 (the three separate pg_nchw_to_nhwc calls wrote 3 + 3 + 4 scalars per pixel at a 28-byte stride and ran
// at 0.6 TB/s on 512 x 512 inputs).
template <bool V8>
__global__ __launch_bounds__(256) void k_din_fill(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ real,
                                                  float* __restrict__ fake, int ld, int N, int Cx, int Cy, unsigned HW) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const unsigned np = (unsigned)N * HW, stride = gridDim.x * blockDim.x;
    for (unsigned pix = blockIdx.x * blockDim.x + threadIdx.x; pix < np; pix += stride) {
        const unsigned n = pix / HW, hw = pix - n * HW;
        const float* xp = x + (size_t)n * Cx * HW + hw;
        const float* yp = y + (size_t)n * Cy * HW + hw;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = c < Cx ? xp[(size_t)c * HW] : (c < Cx + Cy ? yp[(size_t)(c - Cx) * HW] : 0.f);
        float* rp = real + (size_t)pix * ld;
        float* fp = fake + (size_t)pix * ld;
        if (V8) {
            *reinterpret_cast<f32x4*>(rp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(rp + 4) = f32x4{v[4], v[5], v[6], v[7]};
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = c < Cx ? v[c] : 0.f;
            *reinterpret_cast<f32x4*>(fp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(fp + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < ld) {
                    rp[c] = v[c];
                    fp[c] = c < Cx ? v[c] : 0.f;
                }
        }
    }
}

// dst NCHW <- src NHWC; one thread per NCHW element (pixel fastest)
__global__ void k_nhwc_to_nchw(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int N, int C,
                               long HW) {
    const long total = (long)N * HW * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % HW;
        const long nc = i / HW;
        const int c = (int)(nc % C);
        const long n = nc / C;
        dst[i] = src[(n * HW + hw) * ld_src + c];
    }
}

// decoded image bytes [npix][C] (HWC, what a JPEG decoder hands over) -> float NHWC slice: value / 255.f, the reference's
// `read_image(...) / 255.` (io.py:42) computed on the device
__global__ void k_u8_to_f32(const uint8_t* __restrict__ src, float* __restrict__ dst, int ld_dst, long npix, int C, float div) {
    const long total = npix * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long pix = i / C;
        dst[pix * ld_dst + c] = (float)src[i] / div;
    }
}

// label map bytes [npix] -> one-hot float NHWC slice of nl channels: dst[pix][i] = ((uint8)(src + add) == labels[i])
// (io.py:43,53-56: `read_image(mask, GRAY) + 1` is uint8 arithmetic -- 255 wraps to 0 -- then `mask[i, labels == label] = 1`)
struct LabelSet {
    int v[PG_MAX_LABELS];
};
__global__ void k_labels_onehot(const uint8_t* __restrict__ src, float* __restrict__ dst, int ld_dst, long npix, LabelSet ls,
                                int nl, int add) {
    for (long pix = blockIdx.x * (long)blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
        const int v = (uint8_t)(src[pix] + add);
        for (int i = 0; i < nl; ++i) dst[pix * ld_dst + i] = (v == ls.v[i]) ? 1.f : 0.f;
    }
}

__global__ void k_copy_channels(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst,
                                long npix, int C) {
    const long total = npix * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long pix = i / C;
        dst[pix * ld_dst + c] = src[pix * ld_src + c];
    }
}

__global__ void k_fill(float* __restrict__ dst, long n, float value) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = value;
}

// fp32 NHWC channel slice (C <= 8 channels, pixel stride ld_src) -> bf16 8-channel pixels: one 16-byte store per pixel, pads zero
template <bool V8>      // V8: 8-float source pixels, 16-byte aligned (two 16-byte loads per pixel instead of C scalar ones at a ragged stride)
__global__ void k_pad8_bf16(const float* __restrict__ src, int ld_src, unsigned* __restrict__ dst, long npix, int C) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        float v[8];
        if (V8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8), b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = c < C ? a[c] : 0.f;
                v[4 + c] = 4 + c < C ? b[c] : 0.f;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (c < C) ? src[i * ld_src + c] : 0.f;
        }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bf16x2 h;
            h[0] = (__bf16)v[2 * k];
            h[1] = (__bf16)v[2 * k + 1];
            o[k] = __builtin_bit_cast(unsigned, h);
        }
        *reinterpret_cast<u32x4*>(dst + 4 * i) = o;
    }
}

int blocks_for(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" {

int pg_version(void) { return 2; }

int pg_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 float bc1, float sqrt_bc2, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || bc1 <= 0.f || sqrt_bc2 <= 0.f) return PG_EINVAL;
    const uintptr_t al = (uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v;
    if (al & 15) return PG_EINVAL;
    hipLaunchKernelGGL(k_adam<false>, dim3(blocks_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr / bc1,
                       beta1, beta2, eps, sqrt_bc2, (const float*)nullptr);
    return pg_launch_status();
}

int pg_adam_step_dev(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2, float eps, const float* scalars,
                     void* stream) {
    if (!p || !g || !m || !v || !scalars || n <= 0) return PG_EINVAL;
    const uintptr_t al = (uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v;
    if (al & 15) return PG_EINVAL;
    hipLaunchKernelGGL(k_adam<true>, dim3(blocks_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, 0.f, beta1, beta2, eps,
                       0.f, scalars);
    return pg_launch_status();
}

int pg_nchw_to_nhwc(const float* src, float* dst, int ld_dst, int N, int C, int H, int W, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ld_dst < C) return PG_EINVAL;
    const long HW = (long)H * W;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(blocks_for(C <= 8 ? (long)N * HW : (long)N * HW * C)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       ld_dst, N, C, HW);
    return pg_launch_status();
}

int pg_pad8_bf16(const float* src, int ld_src, void* dst, long npix, int C, void* stream) {
    if (!src || !dst || npix <= 0 || C <= 0 || C > 8 || ld_src < C || ((uintptr_t)dst & 15)) return PG_EINVAL;
    if (ld_src == 8 && ((uintptr_t)src & 15) == 0)
        hipLaunchKernelGGL(k_pad8_bf16<true>, dim3(blocks_for(npix)), dim3(256), 0, (hipStream_t)stream, src, ld_src, (unsigned*)dst, npix, C);
    else
        hipLaunchKernelGGL(k_pad8_bf16<false>, dim3(blocks_for(npix)), dim3(256), 0, (hipStream_t)stream, src, ld_src, (unsigned*)dst, npix, C);
    return pg_launch_status();
}

int pg_din_fill(const float* x, const float* y, float* real, float* fake, int ld, int N, int Cx, int Cy, int H, int W, void* stream) {
    if (!x || !y || !real || !fake || N <= 0 || Cx <= 0 || Cy <= 0 || Cx + Cy > 8 || ld < Cx + Cy || ld > 8 || H <= 0 || W <= 0) return PG_EINVAL;
    const long HW = (long)H * W;
    if ((long)N * HW >= 0x7fffffffL) return PG_EINVAL;
    const bool v8 = ld == 8 && (((uintptr_t)real | (uintptr_t)fake) & 15) == 0;
    if (v8)
        hipLaunchKernelGGL(k_din_fill<true>, dim3(blocks_for((long)N * HW)), dim3(256), 0, (hipStream_t)stream, x, y, real, fake, ld, N, Cx, Cy,
                           (unsigned)HW);
    else
        hipLaunchKernelGGL(k_din_fill<false>, dim3(blocks_for((long)N * HW)), dim3(256), 0, (hipStream_t)stream, x, y, real, fake, ld, N, Cx, Cy,
                           (unsigned)HW);
    return pg_launch_status();
}

int pg_nhwc_to_nchw(const float* src, int ld_src, float* dst, int N, int C, int H, int W, void* stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ld_src < C) return PG_EINVAL;
    const long HW = (long)H * W;
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(blocks_for((long)N * HW * C)), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                       dst, N, C, HW);
    return pg_launch_status();
}

int pg_copy_channels(const float* src, int ld_src, float* dst, int ld_dst, long npix, int C, void* stream) {
    if (!src || !dst || npix <= 0 || C <= 0 || ld_src < C || ld_dst < C) return PG_EINVAL;
    hipLaunchKernelGGL(k_copy_channels, dim3(blocks_for(npix * C)), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst,
                       ld_dst, npix, C);
    return pg_launch_status();
}

int pg_u8_to_f32(const unsigned char* src, float* dst, int ld_dst, long npix, int C, float div, void* stream) {
    if (!src || !dst || npix <= 0 || C <= 0 || ld_dst < C || div == 0.f) return PG_EINVAL;
    hipLaunchKernelGGL(k_u8_to_f32, dim3(blocks_for(npix * C)), dim3(256), 0, (hipStream_t)stream, src, dst, ld_dst, npix, C,
                       div);
    return pg_launch_status();
}

int pg_labels_to_onehot(const unsigned char* src, float* dst, int ld_dst, long npix, const int* labels, int nlabels, int add,
                        void* stream) {
    if (!src || !dst || !labels || npix <= 0 || nlabels <= 0 || nlabels > PG_MAX_LABELS || ld_dst < nlabels) return PG_EINVAL;
    LabelSet ls;
    for (int i = 0; i < PG_MAX_LABELS; ++i) ls.v[i] = i < nlabels ? labels[i] : -1;
    hipLaunchKernelGGL(k_labels_onehot, dim3(blocks_for(npix)), dim3(256), 0, (hipStream_t)stream, src, dst, ld_dst, npix, ls,
                       nlabels, add);
    return pg_launch_status();
}

int pg_fill(float* dst, long n, float value, void* stream) {
    if (!dst || n <= 0) return PG_EINVAL;
    hipLaunchKernelGGL(k_fill, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, dst, n, value);
    return pg_launch_status();
}

}  // extern "C"
