// norm_act.hip -- InstanceNorm2d + activation + dropout (forward / backward), plain activation
// forward / backward, channel softmax.  HBM-bound elementwise work over NHWC tensors: 16-byte accesses
// along the channel dim when alignment allows (VEC = 4), scalar otherwise (VEC = 1).
//
// InstanceNorm statistics are per (sample, channel) over HW.  One 256-thread workgroup owns one sample
// and a group of G channel-units (G*VEC channels); its threads stride over the pixels.  Sums are
// accumulated in fp64 (torch's CPU batch_norm uses a double accumulator for float input) with a two-pass
// variance, so the tiny-spatial layers (enc6: 2x2) are not at the mercy of E[x^2]-E[x]^2 cancellation.
// Reductions are fixed-order LDS trees: bit-reproducible run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "patchgan_hip.h"
#include "pg_common.h"

namespace {

template <int VEC>
struct Vec {
    float v[VEC];
};

// Pointer to an fp32 or a bf16 activation tensor (bf16 storage mode); arithmetic in ELEMENTS.  Statistics, partial
// sums and all arithmetic stay fp32 / fp64 either way: a bf16 tensor is widened on load and rounded (RNE) on store.
struct TPtr {
    const char* p;
    int bf;
    __host__ __device__ TPtr operator+(long n) const { return TPtr{p + n * (bf ? 2 : 4), bf}; }
    __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
inline TPtr tp(const void* p, bool bf) { return TPtr{(const char*)p, bf ? 1 : 0}; }

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

// KD: the storage type where the instantiation knows it (0 fp32, 1 bf16; -1 = read t.bf at run time).  VEC = 8 is bf16 by construction
// (v8_ok below).  A known type leaves no branch around the access, so the loads of an unrolled loop are issued back to back.
template <int VEC, int KD = -1>
__device__ __forceinline__ Vec<VEC> vload(TPtr t) {
    Vec<VEC> r;
    const bool is_bf = (VEC == 8 || KD == 1) ? true : (KD == 0 ? false : (bool)t.bf);
    if (is_bf) {
        if constexpr (VEC == 8) {          // 16 bytes per lane on a bf16 tensor
            const uint4 u = *reinterpret_cast<const uint4*>(t.p);
            r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
            r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
            r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
            r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
        } else if constexpr (VEC == 4) {
            const uint2 u = *reinterpret_cast<const uint2*>(t.p);
            r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
            r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
        } else {
            r.v[0] = bf2f(*reinterpret_cast<const unsigned short*>(t.p));
        }
    } else {
        if constexpr (VEC == 8) {
            const float4 f = *reinterpret_cast<const float4*>(t.p), h = *reinterpret_cast<const float4*>(t.p + 16);
            r.v[0] = f.x; r.v[1] = f.y; r.v[2] = f.z; r.v[3] = f.w;
            r.v[4] = h.x; r.v[5] = h.y; r.v[6] = h.z; r.v[7] = h.w;
        } else if constexpr (VEC == 4) {
            float4 f = *reinterpret_cast<const float4*>(t.p);
            r.v[0] = f.x; r.v[1] = f.y; r.v[2] = f.z; r.v[3] = f.w;
        } else {
            r.v[0] = *reinterpret_cast<const float*>(t.p);
        }
    }
    return r;
}
template <int VEC, int KD = -1>
__device__ __forceinline__ void vstore(TPtr t, const Vec<VEC>& r) {
    char* q = const_cast<char*>(t.p);
    const bool is_bf = (VEC == 8 || KD == 1) ? true : (KD == 0 ? false : (bool)t.bf);
    if (is_bf) {
        if constexpr (VEC == 8) {
            uint4 u;
            u.x = (unsigned)f2bf(r.v[0]) | ((unsigned)f2bf(r.v[1]) << 16);
            u.y = (unsigned)f2bf(r.v[2]) | ((unsigned)f2bf(r.v[3]) << 16);
            u.z = (unsigned)f2bf(r.v[4]) | ((unsigned)f2bf(r.v[5]) << 16);
            u.w = (unsigned)f2bf(r.v[6]) | ((unsigned)f2bf(r.v[7]) << 16);
            *reinterpret_cast<uint4*>(q) = u;
        } else if constexpr (VEC == 4) {
            uint2 u;
            u.x = (unsigned)f2bf(r.v[0]) | ((unsigned)f2bf(r.v[1]) << 16);
            u.y = (unsigned)f2bf(r.v[2]) | ((unsigned)f2bf(r.v[3]) << 16);
            *reinterpret_cast<uint2*>(q) = u;
        } else {
            *reinterpret_cast<unsigned short*>(q) = f2bf(r.v[0]);
        }
    } else {
        if constexpr (VEC == 8) {
            *reinterpret_cast<float4*>(q) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
            *reinterpret_cast<float4*>(q + 16) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
        } else if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(q) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
        } else {
            *reinterpret_cast<float*>(q) = r.v[0];
        }
    }
}

// derivative of the post-norm activation evaluated at the normalised value z
__device__ __forceinline__ float pg_norm_act_grad(float z, int act) {
    switch (act) {
        case PG_ACT_LEAKY: return z > 0.f ? 1.f : 0.2f;
        case PG_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case PG_ACT_TANH: { float t = tanhf(z); return 1.f - t * t; }
        case PG_ACT_SIGMOID: { float t = 1.f / (1.f + expf(-z)); return t * (1.f - t); }
        default: return 1.f;
    }
}

// tree-reduce red[k][tid] over the pixel-lane dim (tid = pl*G + cu); result in red[k][cu]
template <int NK>
__device__ __forceinline__ void lane_tree(double (*red)[256], int tid, int G) {
    const int PL = 256 / G;
    for (int off = PL >> 1; off > 0; off >>= 1) {
        __syncthreads();
        if (tid < off * G) {
#pragma unroll
            for (int k = 0; k < NK; ++k) red[k][tid] += red[k][tid + off * G];
        }
    }
    __syncthreads();
}

template <int VEC>
__global__ __launch_bounds__(256) void k_instnorm_fwd(TPtr y, int ld_y, TPtr out,
                                                      int ld_out, float* __restrict__ stats, int HW, int C, int G,
                                                      int act, float eps, float drop_p, uint64_t seed) {
    __shared__ double red[VEC][256];
    const int tid = threadIdx.x, cu = tid % G, pl = tid / G, PL = 256 / G;
    const int c0 = (blockIdx.x * G + cu) * VEC;
    const int n = blockIdx.y;
    const bool on = c0 < C;
    const TPtr yb = y + ((long)n * HW * ld_y + c0);
    const TPtr ob = out + ((long)n * HW * ld_out + c0);

    double s[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = 0.0;
    if (on)
        for (int pix = pl; pix < HW; pix += PL) {
            Vec<VEC> v = vload<VEC>(yb + (long)((long)pix * ld_y));
#pragma unroll
            for (int k = 0; k < VEC; ++k) s[k] += (double)v.v[k];
        }
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[k][tid] = s[k];
    lane_tree<VEC>(red, tid, G);
    double mean[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) mean[k] = red[k][cu] / (double)HW;
    __syncthreads();

#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = 0.0;
    if (on)
        for (int pix = pl; pix < HW; pix += PL) {
            Vec<VEC> v = vload<VEC>(yb + (long)((long)pix * ld_y));
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                double d = (double)v.v[k] - mean[k];
                s[k] += d * d;
            }
        }
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[k][tid] = s[k];
    lane_tree<VEC>(red, tid, G);
    float alpha[VEC], beta[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const double var = red[k][cu] / (double)HW;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float mf = (float)mean[k];
        alpha[k] = rstd;
        beta[k] = -mf * rstd;
        if (on && pl == 0) {
            stats[((long)n * C + c0 + k) * 2 + 0] = mf;
            stats[((long)n * C + c0 + k) * 2 + 1] = rstd;
        }
    }
    if (!on) return;
    const float keep_scale = 1.f / (1.f - drop_p);
    for (int pix = pl; pix < HW; pix += PL) {
        Vec<VEC> v = vload<VEC>(yb + (long)((long)pix * ld_y));
        Vec<VEC> o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            // x*alpha + beta, two roundings, as torch's CPU batch_norm transform does
            float z = __fadd_rn(__fmul_rn(v.v[k], alpha[k]), beta[k]);
            float a = pg_act(z, act);
            if (drop_p > 0.f) {
                const uint64_t e = ((uint64_t)n * HW + pix) * C + c0 + k;
                a = pg_dropout_keep(seed, e, drop_p) ? a * keep_scale : 0.f;
            }
            o.v[k] = a;
        }
        vstore<VEC>(ob + (long)((long)pix * ld_out), o);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_instnorm_bwd(TPtr g1, int ld_g1,
                                                      TPtr g2, int ld_g2,
                                                      TPtr y, int ld_y,
                                                      const float* __restrict__ stats, TPtr dy,
                                                      int ld_dy, int HW, int C, int G, int act, float drop_p,
                                                      uint64_t seed) {
    __shared__ double red[2 * VEC][256];
    const int tid = threadIdx.x, cu = tid % G, pl = tid / G, PL = 256 / G;
    const int c0 = (blockIdx.x * G + cu) * VEC;
    const int n = blockIdx.y;
    const bool on = c0 < C;
    const long nb = (long)n * HW;
    float mean[VEC], rstd[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        mean[k] = on ? stats[((long)n * C + c0 + k) * 2 + 0] : 0.f;
        rstd[k] = on ? stats[((long)n * C + c0 + k) * 2 + 1] : 0.f;
    }
    const float keep_scale = 1.f / (1.f - drop_p);

    auto dz_of = [&](int pix, Vec<VEC>& xm, Vec<VEC>& dz) {
        Vec<VEC> v = vload<VEC>(y + (long)((nb + pix) * ld_y + c0));
        Vec<VEC> g = vload<VEC>(g1 + (long)((nb + pix) * ld_g1 + c0));
        if (g2) {
            Vec<VEC> h = vload<VEC>(g2 + (long)((nb + pix) * ld_g2 + c0));
#pragma unroll
            for (int k = 0; k < VEC; ++k) g.v[k] += h.v[k];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float gg = g.v[k];
            if (drop_p > 0.f) {
                const uint64_t e = ((uint64_t)n * HW + pix) * C + c0 + k;
                gg = pg_dropout_keep(seed, e, drop_p) ? gg * keep_scale : 0.f;
            }
            const float z = __fadd_rn(__fmul_rn(v.v[k], rstd[k]), -mean[k] * rstd[k]);
            float d;
            switch (act) {
                case PG_ACT_LEAKY: d = z > 0.f ? 1.f : 0.2f; break;
                case PG_ACT_RELU: d = z > 0.f ? 1.f : 0.f; break;
                case PG_ACT_TANH: { float t = tanhf(z); d = 1.f - t * t; } break;
                case PG_ACT_SIGMOID: { float t = 1.f / (1.f + expf(-z)); d = t * (1.f - t); } break;
                default: d = 1.f;
            }
            xm.v[k] = v.v[k] - mean[k];
            dz.v[k] = gg * d;
        }
    };

    double s1[VEC], s2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s1[k] = s2[k] = 0.0;
    if (on)
        for (int pix = pl; pix < HW; pix += PL) {
            Vec<VEC> xm, dz;
            dz_of(pix, xm, dz);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                s1[k] += (double)dz.v[k];
                s2[k] += (double)dz.v[k] * (double)xm.v[k];
            }
        }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        red[k][tid] = s1[k];
        red[VEC + k][tid] = s2[k];
    }
    lane_tree<2 * VEC>(red, tid, G);
    float gmean[VEC], kk[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        gmean[k] = (float)(red[k][cu] / (double)HW);
        kk[k] = (float)(red[VEC + k][cu] * (double)rstd[k] * (double)rstd[k] / (double)HW);
    }
    if (!on) return;
    for (int pix = pl; pix < HW; pix += PL) {
        Vec<VEC> xm, dz, o;
        dz_of(pix, xm, dz);
#pragma unroll
        for (int k = 0; k < VEC; ++k) o.v[k] = (dz.v[k] - gmean[k] - xm.v[k] * kk[k]) * rstd[k];
        vstore<VEC>(dy + (long)((nb + pix) * ld_dy + c0), o);
    }
}

// ------------------------------------------------------------------------------------------------
// Chunked path for large planes (HW >= 512, PATCHGAN_IN_CHUNK_MIN): the single-workgroup-per-(n, channel group) kernels above either
// run too few workgroups or (G = 1) touch 16 B of every 128-B line per lane.  Here a workgroup owns (n, pixel
// chunk, channel group) with G as wide as the channel count allows (full-line reads), partial sums go to a
// workspace in fp64, a tiny merge kernel finishes the statistics in fixed order, and the apply pass is a flat
// grid-stride elementwise kernel.  fp64 sum / sum-of-squares (squares of fp32 are exact in fp64) replaces the
// two-pass variance: relative error ~1e-16 * mean^2/var.
// ------------------------------------------------------------------------------------------------
template <int VEC, bool BWD, int KD = -1, bool HG2 = false>      // HG2: a second gradient source g2 (the skip connection's) is summed in
__global__ __launch_bounds__(256) void k_in_partial(TPtr y, int ld_y,
                                                    TPtr g1, int ld_g1,
                                                    TPtr g2, int ld_g2,
                                                    const float* __restrict__ stats, double* __restrict__ part, int HW,
                                                    int C, int G, int pix_per_chunk, int act, float drop_p,
                                                    uint64_t seed) {
    __shared__ double red[2 * VEC][256];
    const int tid = threadIdx.x, cu = tid % G, pl = tid / G, PL = 256 / G;
    const int c0 = (blockIdx.x * G + cu) * VEC;
    const int chunk = blockIdx.y, nchunk = gridDim.y, n = blockIdx.z;
    const bool on = c0 < C;
    const long nb = (long)n * HW;
    const int p_begin = chunk * pix_per_chunk, p_end = min(HW, p_begin + pix_per_chunk);
    float mean[VEC], rstd[VEC];
    if (BWD) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            mean[k] = on ? stats[((long)n * C + c0 + k) * 2 + 0] : 0.f;
            rstd[k] = on ? stats[((long)n * C + c0 + k) * 2 + 1] : 0.f;
        }
    }
    const float keep_scale = 1.f / (1.f - drop_p);
    double s1[VEC], s2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s1[k] = s2[k] = 0.0;
    // one pixel's contribution, added in pixel order (the unrolled loop below only ISSUES the loads of UNR pixels together: a thread with one
    // load pair in flight per iteration ran this pass at the latency of a load, 2.5-3.5 TB/s; the sums and their order are unchanged)
    auto accumulate = [&](int pix, const Vec<VEC>& v, Vec<VEC> g, const Vec<VEC>& h) {
        if (!BWD) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const double d = (double)v.v[k];
                s1[k] += d;
                s2[k] += d * d;
            }
        } else {
            if (HG2) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) g.v[k] += h.v[k];
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float gg = g.v[k];
                if (drop_p > 0.f) {
                    const uint64_t e = ((uint64_t)n * HW + pix) * C + c0 + k;
                    gg = pg_dropout_keep(seed, e, drop_p) ? gg * keep_scale : 0.f;
                }
                const float z = __fadd_rn(__fmul_rn(v.v[k], rstd[k]), -mean[k] * rstd[k]);
                const float dz = gg * pg_norm_act_grad(z, act);
                s1[k] += (double)dz;
                s2[k] += (double)dz * (double)(v.v[k] - mean[k]);
            }
        }
    };
    if (on) {
        constexpr int UNR = 4;
        int pix = p_begin + pl;
        for (; pix + (UNR - 1) * PL < p_end; pix += UNR * PL) {
            Vec<VEC> v[UNR], g[UNR], h[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                v[u] = vload<VEC, KD>(y + (long)((nb + pix + u * PL) * ld_y + c0));
                if (BWD) {
                    g[u] = vload<VEC, KD>(g1 + (long)((nb + pix + u * PL) * ld_g1 + c0));
                    if (HG2) h[u] = vload<VEC, KD>(g2 + (long)((nb + pix + u * PL) * ld_g2 + c0));
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) accumulate(pix + u * PL, v[u], g[u], h[u]);
        }
        for (; pix < p_end; pix += PL) {
            Vec<VEC> v = vload<VEC, KD>(y + (long)((nb + pix) * ld_y + c0)), g, h;
            if (BWD) {
                g = vload<VEC, KD>(g1 + (long)((nb + pix) * ld_g1 + c0));
                if (HG2) h = vload<VEC, KD>(g2 + (long)((nb + pix) * ld_g2 + c0));
            }
            accumulate(pix, v, g, h);
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        red[k][tid] = s1[k];
        red[VEC + k][tid] = s2[k];
    }
    lane_tree<2 * VEC>(red, tid, G);
    if (on && pl == 0) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            double* o = part + (((long)n * nchunk + chunk) * C + c0 + k) * 2;
            o[0] = red[k][cu];
            o[1] = red[VEC + k][cu];
        }
    }
}

// merge the chunk partials of one (n, c) in chunk order.  FWD: stats = (mean, rstd).  BWD: coef = (gmean, kk).
template <bool BWD>
__global__ void k_in_merge(const double* __restrict__ part, int nchunk, int NC, int C, int HW, float eps,
                           float* __restrict__ stats, float* __restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    const int n = i / C, c = i - n * C;
    double s1 = 0.0, s2 = 0.0;
    const double2* p2 = reinterpret_cast<const double2*>(part) + ((long)n * nchunk) * C + c;
    // chunk order, 32 loads in flight (the adds stay sequential: same sums; with 8 in flight a merge over 128 chunks took 8.5 us)
    int k = 0;
    for (; k + 32 <= nchunk; k += 32) {
        double2 v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = p2[(long)(k + u) * C];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            s1 += v[u].x;
            s2 += v[u].y;
        }
    }
#pragma unroll 8
    for (; k < nchunk; ++k) {
        const double2 v = p2[(long)k * C];
        s1 += v.x;
        s2 += v.y;
    }
    if (!BWD) {
        const double mean = s1 / (double)HW;
        double var = s2 / (double)HW - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        stats[(long)i * 2 + 0] = (float)mean;
        stats[(long)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
        const double rstd = (double)stats[(long)i * 2 + 1];
        coef[(long)i * 2 + 0] = (float)(s1 / (double)HW);
        coef[(long)i * 2 + 1] = (float)(s2 * rstd * rstd / (double)HW);
    }
}

// grid (blocks per sample, N).  Where the channel units divide 256 (every power-of-two channel count) a thread keeps ONE channel unit
// for the whole launch -- its statistics / coefficients live in registers and the pixel index advances by a constant: no division and
// no per-element reload of the statistics in the loop (the flat form spent two 64-bit divisions and 16 scalar-indexed loads per
// 16 bytes of payload and ran at 3 TB/s).  Other channel counts: the flat form within the sample, 32-bit.
template <int VEC, bool BWD, int KD = -1, bool HG2 = false>
__global__ __launch_bounds__(256) void k_in_apply(TPtr y, int ld_y, TPtr g1, int ld_g1,
                           TPtr g2, int ld_g2, const float* __restrict__ stats,
                           const float* __restrict__ coef, TPtr out, int ld_out, int N, int HW, int C,
                           int act, float drop_p, uint64_t seed) {
    const int cq = C / VEC;
    const int n = blockIdx.y;
    const long nb = (long)n * HW;
    const float keep_scale = 1.f / (1.f - drop_p);
    const bool fixed = cq <= 256 && 256 % cq == 0;
    const int PL = fixed ? 256 / cq : 1;
    int c0 = fixed ? (int)(threadIdx.x % cq) * VEC : 0;
    float mf[VEC], rs[VEC], cf0[VEC], cf1[VEC];
    auto load_coef = [&]() {
        const float* st = stats + ((long)n * C + c0) * 2;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            mf[k] = st[2 * k];
            rs[k] = st[2 * k + 1];
        }
        if (BWD) {
            const float* cf = coef + ((long)n * C + c0) * 2;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                cf0[k] = cf[2 * k];
                cf1[k] = cf[2 * k + 1];
            }
        }
    };
    if (fixed) load_coef();
    const int first = fixed ? blockIdx.x * PL + (int)(threadIdx.x / cq) : blockIdx.x * 256 + (int)threadIdx.x;
    const int step = fixed ? gridDim.x * PL : gridDim.x * 256;
    const int limit = fixed ? HW : HW * cq;
    auto one = [&](long pix, const Vec<VEC>& v, Vec<VEC> g, const Vec<VEC>& h) {
        Vec<VEC> o;
        if (!BWD) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float a = pg_act(__fadd_rn(__fmul_rn(v.v[k], rs[k]), -mf[k] * rs[k]), act);
                if (drop_p > 0.f) a = pg_dropout_keep(seed, (uint64_t)pix * C + c0 + k, drop_p) ? a * keep_scale : 0.f;
                o.v[k] = a;
            }
        } else {
            if (HG2) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) g.v[k] += h.v[k];
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float gg = g.v[k];
                if (drop_p > 0.f) gg = pg_dropout_keep(seed, (uint64_t)pix * C + c0 + k, drop_p) ? gg * keep_scale : 0.f;
                const float z = __fadd_rn(__fmul_rn(v.v[k], rs[k]), -mf[k] * rs[k]);
                const float dz = gg * pg_norm_act_grad(z, act);
                o.v[k] = (dz - cf0[k] - (v.v[k] - mf[k]) * cf1[k]) * rs[k];
            }
        }
        vstore<VEC, KD>(out + (long)(pix * ld_out + c0), o);
    };
    int i = first;
    if (fixed) {
        // the loads of UNR pixels are issued together (one load set in flight per thread left this pass at the latency of a load)
        constexpr int UNR = 4;
        for (; i + (UNR - 1) * step < limit; i += UNR * step) {
            Vec<VEC> v[UNR], g[UNR], h[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long pix = nb + i + u * step;
                v[u] = vload<VEC, KD>(y + (long)(pix * ld_y + c0));
                if (BWD) {
                    g[u] = vload<VEC, KD>(g1 + (long)(pix * ld_g1 + c0));
                    if (HG2) h[u] = vload<VEC, KD>(g2 + (long)(pix * ld_g2 + c0));
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) one(nb + i + u * step, v[u], g[u], h[u]);
        }
    }
    for (; i < limit; i += step) {
        int p = i;
        if (!fixed) {
            p = i / cq;
            c0 = (i - p * cq) * VEC;
            load_coef();
        }
        const long pix = nb + p;
        Vec<VEC> v = vload<VEC, KD>(y + (long)(pix * ld_y + c0)), g, h;
        if (BWD) {
            g = vload<VEC, KD>(g1 + (long)(pix * ld_g1 + c0));
            if (HG2) h = vload<VEC, KD>(g2 + (long)(pix * ld_g2 + c0));
        }
        one(pix, v, g, h);
    }
}

// blocks per sample of k_in_apply: ~4096 workgroups in all, at least 4 (fixed form: pixel rows of 256 / cq) iterations per thread where
// the plane allows
inline dim3 in_apply_grid(int N, int HW, int units) {
    const long per_sample = ((long)HW * units + 255) / 256;
    long gx = (4096 + N - 1) / N;
    if (gx > (per_sample + 3) / 4) gx = (per_sample + 3) / 4;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)N);
}

template <int VEC>
__global__ void k_act_fwd(TPtr y, int ld_y, TPtr out, int ld_out, long npix, int C,
                          int act, float drop_p, uint64_t seed) {
    const int cq = C / VEC;
    const long total = npix * cq;
    const float keep_scale = 1.f / (1.f - drop_p);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cq;
        const int c0 = (int)(i - pix * cq) * VEC;
        Vec<VEC> v = vload<VEC>(y + (long)(pix * ld_y + c0)), o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float a = pg_act(v.v[k], act);
            if (drop_p > 0.f) a = pg_dropout_keep(seed, (uint64_t)pix * C + c0 + k, drop_p) ? a * keep_scale : 0.f;
            o.v[k] = a;
        }
        vstore<VEC>(out + (long)(pix * ld_out + c0), o);
    }
}

template <int VEC>
__global__ void k_act_bwd(TPtr g1, int ld_g1, TPtr g2, int ld_g2,
                          TPtr a, int ld_a, TPtr dy, int ld_dy, long npix, int C,
                          int act, float drop_p, uint64_t seed) {
    const int cq = C / VEC;
    const long total = npix * cq;
    const float keep_scale = 1.f / (1.f - drop_p);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cq;
        const int c0 = (int)(i - pix * cq) * VEC;
        Vec<VEC> g = vload<VEC>(g1 + (long)(pix * ld_g1 + c0)), o;
        if (g2) {
            Vec<VEC> h = vload<VEC>(g2 + (long)(pix * ld_g2 + c0));
#pragma unroll
            for (int k = 0; k < VEC; ++k) g.v[k] += h.v[k];
        }
        Vec<VEC> av;
        if (a) av = vload<VEC>(a + (long)(pix * ld_a + c0));
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float gg = g.v[k];
            float ao = a ? av.v[k] : 0.f;
            if (drop_p > 0.f) {
                // the saved tensor is the post-dropout output: undo the scale to recover the activation output
                const bool keep = pg_dropout_keep(seed, (uint64_t)pix * C + c0 + k, drop_p);
                gg = keep ? gg * keep_scale : 0.f;
                ao = ao * (1.f - drop_p);
            }
            o.v[k] = gg * pg_act_grad_from_out(ao, act);
        }
        vstore<VEC>(dy + (long)(pix * ld_dy + c0), o);
    }
}

// Softmax over the C channels of a pixel (unet.py:60, nn.Softmax(dim=1)) and its backward (two gradient sources summed on the fly).
// One thread per pixel; C == 4 on 16-byte-aligned views (the multi-class head of BASELINE config 3/4): the pixel is one 16-byte load /
// store and every exponential is evaluated once (the scalar form below: C scalar accesses, each exponential twice).  Same expression
// per element in both forms (bit-identical).
typedef float sm_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool sm_al(const void* p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; }
__device__ __forceinline__ sm_f4 sm_ld(const float* p, bool vec) {
    if (vec) return *reinterpret_cast<const sm_f4*>(p);
    return sm_f4{p[0], p[1], p[2], p[3]};
}
__device__ __forceinline__ void sm_st(float* p, sm_f4 v, bool vec) {
    if (vec) {
        *reinterpret_cast<sm_f4*>(p) = v;
    } else {
        p[0] = v[0];
        p[1] = v[1];
        p[2] = v[2];
        p[3] = v[3];
    }
}
__global__ void k_softmax_fwd(const float* __restrict__ y, int ld_y, float* __restrict__ out, int ld_out, long npix,
                              int C) {
    if (C == 4) {       // (a slice of a wider pixel -- the mask channels of the discriminator-input buffer -- is accessed by element)
        const bool vy = sm_al(y, ld_y), vo = sm_al(out, ld_out);
        for (long pix = blockIdx.x * (long)blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
            const sm_f4 v = sm_ld(y + pix * ld_y, vy);
            float m = v[0];
#pragma unroll
            for (int c = 1; c < 4; ++c) m = fmaxf(m, v[c]);
            sm_f4 e;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                e[c] = expf(v[c] - m);
                s += e[c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = e[c] / s;
            sm_st(out + pix * ld_out, e, vo);
        }
        return;
    }
    for (long pix = blockIdx.x * (long)blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
        const float* p = y + pix * ld_y;
        float m = p[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, p[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(p[c] - m);
        float* o = out + pix * ld_out;
        for (int c = 0; c < C; ++c) o[c] = expf(p[c] - m) / s;
    }
}

__global__ void k_softmax_bwd(const float* __restrict__ g1, int ld_g1, const float* __restrict__ g2, int ld_g2,
                              const float* __restrict__ out, int ld_out, float* __restrict__ dy, int ld_dy, long npix,
                              int C) {
    if (C == 4) {
        const bool v1 = sm_al(g1, ld_g1), v2 = g2 && sm_al(g2, ld_g2), vo = sm_al(out, ld_out), vd = sm_al(dy, ld_dy);
        for (long pix = blockIdx.x * (long)blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
            const sm_f4 o = sm_ld(out + pix * ld_out, vo);
            sm_f4 a = sm_ld(g1 + pix * ld_g1, v1);
            const sm_f4 b = g2 ? sm_ld(g2 + pix * ld_g2, v2) : sm_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] = a[c] + b[c];
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) dot += a[c] * o[c];
            sm_f4 d;
#pragma unroll
            for (int c = 0; c < 4; ++c) d[c] = o[c] * (a[c] - dot);
            sm_st(dy + pix * ld_dy, d, vd);
        }
        return;
    }
    for (long pix = blockIdx.x * (long)blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
        const float* o = out + pix * ld_out;
        const float* a = g1 + pix * ld_g1;
        const float* b = g2 ? g2 + pix * ld_g2 : nullptr;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot += (a[c] + (b ? b[c] : 0.f)) * o[c];
        float* d = dy + pix * ld_dy;
        for (int c = 0; c < C; ++c) d[c] = o[c] * ((a[c] + (b ? b[c] : 0.f)) - dot);
    }
}

__global__ void k_dropout_mask(float* __restrict__ mask, long n, float p, uint64_t seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        mask[i] = pg_dropout_keep(seed, (uint64_t)i, p) ? 1.f : 0.f;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// channel-units per workgroup: as many as keeps >= ~512 workgroups, at most 32 (128 B per pixel row segment)
int pick_group(int N, int units) {
    int G = 1;
    while (G < 32 && G * 2 <= units && (long)N * ((units + G * 2 - 1) / (G * 2)) >= 512) G *= 2;
    return G;
}

// chunked InstanceNorm plan: channel-group width G (as wide as the channels allow, <= 64 units), pixel chunks so
// that the grid has >= ~2048 workgroups; planes under 512 pixels keep the single-workgroup kernels
struct ChunkPlan {
    int G, groups, nchunk, ppc;
    size_t part_bytes, coef_bytes;
};
ChunkPlan chunk_plan(int N, int HW, int C, int vecw) {
    ChunkPlan p;
    const int units = C / vecw;
    int G = 1;
    while (G < 64 && G * 2 <= units) G *= 2;
    p.G = G;
    p.groups = (units + G - 1) / G;
    int nchunk = 1;
    static const int chunk_min = [] {
        const char* e = pg_exp_env("PATCHGAN_IN_CHUNK_MIN");
        return e ? atoi(e) : 512;
    }();
    if (HW >= chunk_min) {
        const int PL = 256 / G;
        static const int wgs = [] {
            const char* e = pg_exp_env("PATCHGAN_IN_CHUNK_WGS");
            return e ? atoi(e) : 1024;
        }();
        long want = (wgs + (long)N * p.groups - 1) / ((long)N * p.groups);
        long maxc = HW / (PL * 2);            // at least two pixels per lane per chunk
        if (want > maxc) want = maxc;
        if (want > 1024) want = 1024;
        nchunk = want < 1 ? 1 : (int)want;
    }
    p.ppc = (HW + nchunk - 1) / nchunk;
    p.nchunk = (HW + p.ppc - 1) / p.ppc;
    p.part_bytes = ((size_t)N * p.nchunk * C * 2 * sizeof(double) + 255) & ~(size_t)255;
    p.coef_bytes = ((size_t)N * C * 2 * sizeof(float) + 255) & ~(size_t)255;
    return p;
}

int ew_blocks(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" {

size_t pg_instnorm_workspace_bytes(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    ChunkPlan a = chunk_plan(N, HW, C, 4), b = chunk_plan(N, HW, C, 1);
    size_t x = a.part_bytes + a.coef_bytes;
    const size_t y = b.part_bytes + b.coef_bytes;
    if (C % 8 == 0) {
        ChunkPlan c = chunk_plan(N, HW, C, 8);
        x = std::max(x, c.part_bytes + c.coef_bytes);
    }
    return x > y ? x : y;
}

// ---- dtype masks of the *_t entry points: bit i set = tensor i (in the order of the signature's tensor arguments) is bf16.
// An 8-byte-per-4-elements access needs 8-byte alignment for bf16, 16 for fp32.
static inline bool al_ok(const void* p, bool bf) { return (reinterpret_cast<uintptr_t>(p) & (bf ? 7 : 15)) == 0; }
// VEC = 8 (16 bytes per lane on bf16 tensors; 8-byte accesses run at 0.54-0.70 of the 16-byte rate): only where EVERY tensor of the call
// is bf16 -- the fp32 paths keep their summation order bit for bit -- with 8-channel granularity and 16-byte alignment throughout.
// Measured in the cfg4 bf16 step (PATCHGAN_NO_VEC8 A/B, one box): 6.60 -> 6.57 ms; the chunked statistics pass is not bound by its
// access width.
static inline bool v8_ok(const void* p, int ld, bool bf) {
    static const bool off = pg_exp_env("PATCHGAN_NO_VEC8") != nullptr;
    if (!p) return !off;
    return !off && bf && (ld % 8 == 0) && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

// launch dispatch of the chunked kernels on what the call knows: all tensors fp32 (dt == 0: loads without a storage-type branch), a second
// gradient source present (backward)
#define IN_GO_G2(K, V, KD, ...)                                                       \
    do {                                                                              \
        if (g2) hipLaunchKernelGGL((K<V, true, KD, true>), __VA_ARGS__);              \
        else hipLaunchKernelGGL((K<V, true, KD, false>), __VA_ARGS__);                \
    } while (0)
#define IN_GO_DT_G2(K, V, ...)                                                        \
    do {                                                                              \
        if (dt == 0) IN_GO_G2(K, V, 0, __VA_ARGS__);                                  \
        else IN_GO_G2(K, V, -1, __VA_ARGS__);                                         \
    } while (0)
#define IN_GO_DT(K, V, ...)                                                           \
    do {                                                                              \
        if (dt == 0) hipLaunchKernelGGL((K<V, false, 0>), __VA_ARGS__);               \
        else hipLaunchKernelGGL((K<V, false>), __VA_ARGS__);                          \
    } while (0)

int pg_instnorm_act_fwd_t(const void* y, int ld_y, void* out, int ld_out, float* stats, int N, int HW, int C, int act, float eps,
                          float drop_p, uint64_t seed, void* ws, size_t ws_bytes, void* stream, int dt) {
    if (!y || !out || !stats || N <= 0 || HW <= 0 || C <= 0 || ld_y < C || ld_out < C) return PG_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || act < 0 || act > PG_ACT_SIGMOID) return PG_EINVAL;
    if (N > 65535 || (dt & ~3)) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const TPtr ty = tp(y, dt & 1), to = tp(out, dt & 2), none = tp(nullptr, false);
    const bool vec = (C % 4 == 0) && (ld_y % 4 == 0) && (ld_out % 4 == 0) && al_ok(y, dt & 1) && al_ok(out, dt & 2);
    const bool vec8 = vec && (C % 8 == 0) && v8_ok(y, ld_y, dt & 1) && v8_ok(out, ld_out, dt & 2);
    ChunkPlan cp = chunk_plan(N, HW, C, vec8 ? 8 : vec ? 4 : 1);
    if (cp.nchunk > 1 && ws && ws_bytes >= cp.part_bytes) {
        double* part = (double*)ws;
        dim3 grid(cp.groups, cp.nchunk, N);
        if (vec8)
            hipLaunchKernelGGL((k_in_partial<8, false>), grid, dim3(256), 0, st, ty, ld_y, none, 0, none, 0, (const float*)nullptr, part,
                               HW, C, cp.G, cp.ppc, act, drop_p, seed);
        else if (vec)
            IN_GO_DT(k_in_partial, 4, grid, dim3(256), 0, st, ty, ld_y, none, 0, none, 0, (const float*)nullptr, part, HW, C, cp.G, cp.ppc, act, drop_p, seed);
        else
            IN_GO_DT(k_in_partial, 1, grid, dim3(256), 0, st, ty, ld_y, none, 0, none, 0, (const float*)nullptr, part, HW, C, cp.G, cp.ppc, act, drop_p, seed);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        hipLaunchKernelGGL((k_in_merge<false>), dim3((N * C + 255) / 256), dim3(256), 0, st, part, cp.nchunk, N * C, C, HW, eps,
                           stats, (float*)nullptr);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        if (vec8)
            hipLaunchKernelGGL((k_in_apply<8, false>), in_apply_grid(N, HW, C / 8), dim3(256), 0, st, ty, ld_y, none, 0,
                               none, 0, stats, (const float*)nullptr, to, ld_out, N, HW, C, act, drop_p, seed);
        else if (vec)
            IN_GO_DT(k_in_apply, 4, in_apply_grid(N, HW, C / 4), dim3(256), 0, st, ty, ld_y, none, 0, none, 0, stats, (const float*)nullptr, to, ld_out, N, HW, C, act, drop_p, seed);
        else
            IN_GO_DT(k_in_apply, 1, in_apply_grid(N, HW, C), dim3(256), 0, st, ty, ld_y, none, 0, none, 0, stats, (const float*)nullptr, to, ld_out, N, HW, C, act, drop_p, seed);
        return pg_launch_status();
    }
    if (vec) {
        const int units = C / 4, G = pick_group(N, units);
        hipLaunchKernelGGL(k_instnorm_fwd<4>, dim3((units + G - 1) / G, N), dim3(256), 0, st, ty, ld_y, to, ld_out, stats,
                           HW, C, G, act, eps, drop_p, seed);
    } else {
        const int G = pick_group(N, C);
        hipLaunchKernelGGL(k_instnorm_fwd<1>, dim3((C + G - 1) / G, N), dim3(256), 0, st, ty, ld_y, to, ld_out, stats, HW,
                           C, G, act, eps, drop_p, seed);
    }
    return pg_launch_status();
}

int pg_instnorm_act_fwd(const float* y, int ld_y, float* out, int ld_out, float* stats, int N, int HW, int C,
                        int act, float eps, float drop_p, uint64_t seed, void* ws, size_t ws_bytes, void* stream) {
    return pg_instnorm_act_fwd_t(y, ld_y, out, ld_out, stats, N, HW, C, act, eps, drop_p, seed, ws, ws_bytes, stream, 0);
}

int pg_instnorm_act_fwd_parts_t(const void* y, int ld_y, void* out, int ld_out, float* stats, const double* part, int chunks,
                                int N, int HW, int C, int act, float eps, float drop_p, uint64_t seed, void* stream, int dt) {
    if (!y || !out || !stats || !part || chunks <= 0 || N <= 0 || HW <= 0 || C <= 0 || ld_y < C || ld_out < C) return PG_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || act < 0 || act > PG_ACT_SIGMOID || (dt & ~3)) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const TPtr ty = tp(y, dt & 1), to = tp(out, dt & 2), none = tp(nullptr, false);
    const bool vec = (C % 4 == 0) && (ld_y % 4 == 0) && (ld_out % 4 == 0) && al_ok(y, dt & 1) && al_ok(out, dt & 2);
    hipLaunchKernelGGL((k_in_merge<false>), dim3((N * C + 255) / 256), dim3(256), 0, st, part, chunks, N * C, C, HW, eps, stats,
                       (float*)nullptr);
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    if (vec && (C % 8 == 0) && v8_ok(y, ld_y, dt & 1) && v8_ok(out, ld_out, dt & 2))
        hipLaunchKernelGGL((k_in_apply<8, false>), in_apply_grid(N, HW, C / 8), dim3(256), 0, st, ty, ld_y, none, 0, none,
                           0, stats, (const float*)nullptr, to, ld_out, N, HW, C, act, drop_p, seed);
    else if (vec)
        IN_GO_DT(k_in_apply, 4, in_apply_grid(N, HW, C / 4), dim3(256), 0, st, ty, ld_y, none, 0, none, 0, stats, (const float*)nullptr, to, ld_out, N, HW, C, act, drop_p, seed);
    else
        IN_GO_DT(k_in_apply, 1, in_apply_grid(N, HW, C), dim3(256), 0, st, ty, ld_y, none, 0, none, 0, stats, (const float*)nullptr, to, ld_out, N, HW, C, act, drop_p, seed);
    return pg_launch_status();
}

int pg_instnorm_act_fwd_parts(const float* y, int ld_y, float* out, int ld_out, float* stats, const double* part, int chunks,
                              int N, int HW, int C, int act, float eps, float drop_p, uint64_t seed, void* stream) {
    return pg_instnorm_act_fwd_parts_t(y, ld_y, out, ld_out, stats, part, chunks, N, HW, C, act, eps, drop_p, seed, stream, 0);
}

int pg_instnorm_act_bwd_t(const void* g1, int ld_g1, const void* g2, int ld_g2, const void* y, int ld_y, const float* stats,
                          void* dy, int ld_dy, int N, int HW, int C, int act, float drop_p, uint64_t seed, void* ws,
                          size_t ws_bytes, void* stream, int dt) {
    if (!g1 || !y || !stats || !dy || N <= 0 || HW <= 0 || C <= 0) return PG_EINVAL;
    if (ld_g1 < C || ld_y < C || ld_dy < C || (g2 && ld_g2 < C)) return PG_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || act < 0 || act > PG_ACT_SIGMOID || N > 65535 || (dt & ~15)) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const TPtr tg1 = tp(g1, dt & 1), tg2 = tp(g2, dt & 2), ty = tp(y, dt & 4), tdy = tp(dy, dt & 8);
    const bool vec = (C % 4 == 0) && (ld_g1 % 4 == 0) && (ld_y % 4 == 0) && (ld_dy % 4 == 0) && al_ok(g1, dt & 1) &&
                     al_ok(y, dt & 4) && al_ok(dy, dt & 8) && (!g2 || ((ld_g2 % 4 == 0) && al_ok(g2, dt & 2)));
    const bool vec8 = vec && (C % 8 == 0) && v8_ok(g1, ld_g1, dt & 1) && v8_ok(g2, ld_g2, dt & 2) && v8_ok(y, ld_y, dt & 4) &&
                      v8_ok(dy, ld_dy, dt & 8);
    ChunkPlan cp = chunk_plan(N, HW, C, vec8 ? 8 : vec ? 4 : 1);
    if (cp.nchunk > 1 && ws && ws_bytes >= cp.part_bytes + cp.coef_bytes) {
        double* part = (double*)ws;
        float* coef = (float*)((char*)ws + cp.part_bytes);
        dim3 grid(cp.groups, cp.nchunk, N);
        if (vec8)
            IN_GO_G2(k_in_partial, 8, -1, grid, dim3(256), 0, st, ty, ld_y, tg1, ld_g1, tg2, ld_g2, stats, part, HW, C, cp.G, cp.ppc, act, drop_p, seed);
        else if (vec)
            IN_GO_DT_G2(k_in_partial, 4, grid, dim3(256), 0, st, ty, ld_y, tg1, ld_g1, tg2, ld_g2, stats, part, HW, C, cp.G, cp.ppc, act, drop_p, seed);
        else
            IN_GO_DT_G2(k_in_partial, 1, grid, dim3(256), 0, st, ty, ld_y, tg1, ld_g1, tg2, ld_g2, stats, part, HW, C, cp.G, cp.ppc, act, drop_p, seed);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        hipLaunchKernelGGL((k_in_merge<true>), dim3((N * C + 255) / 256), dim3(256), 0, st, part, cp.nchunk, N * C, C, HW, 0.f,
                           const_cast<float*>(stats), coef);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        if (vec8)
            IN_GO_G2(k_in_apply, 8, -1, in_apply_grid(N, HW, C / 8), dim3(256), 0, st, ty, ld_y, tg1, ld_g1, tg2, ld_g2, stats, coef, tdy, ld_dy, N, HW, C, act, drop_p, seed);
        else if (vec)
            IN_GO_DT_G2(k_in_apply, 4, in_apply_grid(N, HW, C / 4), dim3(256), 0, st, ty, ld_y, tg1, ld_g1, tg2, ld_g2, stats, coef, tdy, ld_dy, N, HW, C, act, drop_p, seed);
        else
            IN_GO_DT_G2(k_in_apply, 1, in_apply_grid(N, HW, C), dim3(256), 0, st, ty, ld_y, tg1, ld_g1, tg2, ld_g2, stats, coef, tdy, ld_dy, N, HW, C, act, drop_p, seed);
        return pg_launch_status();
    }
    if (vec) {
        const int units = C / 4, G = pick_group(N, units);
        hipLaunchKernelGGL(k_instnorm_bwd<4>, dim3((units + G - 1) / G, N), dim3(256), 0, st, tg1, ld_g1, tg2, ld_g2, ty,
                           ld_y, stats, tdy, ld_dy, HW, C, G, act, drop_p, seed);
    } else {
        const int G = pick_group(N, C);
        hipLaunchKernelGGL(k_instnorm_bwd<1>, dim3((C + G - 1) / G, N), dim3(256), 0, st, tg1, ld_g1, tg2, ld_g2, ty, ld_y,
                           stats, tdy, ld_dy, HW, C, G, act, drop_p, seed);
    }
    return pg_launch_status();
}

int pg_instnorm_act_bwd(const float* g1, int ld_g1, const float* g2, int ld_g2, const float* y, int ld_y,
                        const float* stats, float* dy, int ld_dy, int N, int HW, int C, int act, float drop_p,
                        uint64_t seed, void* ws, size_t ws_bytes, void* stream) {
    return pg_instnorm_act_bwd_t(g1, ld_g1, g2, ld_g2, y, ld_y, stats, dy, ld_dy, N, HW, C, act, drop_p, seed, ws, ws_bytes, stream, 0);
}

int pg_act_fwd_t(const void* y, int ld_y, void* out, int ld_out, long npix, int C, int act, float drop_p, uint64_t seed,
                 void* stream, int dt) {
    if (!y || !out || npix <= 0 || C <= 0 || ld_y < C || ld_out < C) return PG_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || act < 0 || act > PG_ACT_SIGMOID || (dt & ~3)) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const TPtr ty = tp(y, dt & 1), to = tp(out, dt & 2);
    const bool vec = (C % 4 == 0) && (ld_y % 4 == 0) && (ld_out % 4 == 0) && al_ok(y, dt & 1) && al_ok(out, dt & 2);
    if (vec && (C % 8 == 0) && v8_ok(y, ld_y, dt & 1) && v8_ok(out, ld_out, dt & 2))
        hipLaunchKernelGGL(k_act_fwd<8>, dim3(ew_blocks(npix * (C / 8))), dim3(256), 0, st, ty, ld_y, to, ld_out, npix, C,
                           act, drop_p, seed);
    else if (vec)
        hipLaunchKernelGGL(k_act_fwd<4>, dim3(ew_blocks(npix * (C / 4))), dim3(256), 0, st, ty, ld_y, to, ld_out, npix, C,
                           act, drop_p, seed);
    else
        hipLaunchKernelGGL(k_act_fwd<1>, dim3(ew_blocks(npix * C)), dim3(256), 0, st, ty, ld_y, to, ld_out, npix, C, act,
                           drop_p, seed);
    return pg_launch_status();
}

int pg_act_fwd(const float* y, int ld_y, float* out, int ld_out, long npix, int C, int act, float drop_p,
               uint64_t seed, void* stream) {
    return pg_act_fwd_t(y, ld_y, out, ld_out, npix, C, act, drop_p, seed, stream, 0);
}

int pg_act_bwd_t(const void* g1, int ld_g1, const void* g2, int ld_g2, const void* a, int ld_a, void* dy, int ld_dy, long npix,
                 int C, int act, float drop_p, uint64_t seed, void* stream, int dt) {
    if (!g1 || !dy || npix <= 0 || C <= 0 || ld_g1 < C || ld_dy < C) return PG_EINVAL;
    if ((g2 && ld_g2 < C) || (a && ld_a < C) || (!a && act != PG_ACT_NONE)) return PG_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || act < 0 || act > PG_ACT_SIGMOID || (dt & ~15)) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const TPtr tg1 = tp(g1, dt & 1), tg2 = tp(g2, dt & 2), ta = tp(a, dt & 4), tdy = tp(dy, dt & 8);
    const bool vec = (C % 4 == 0) && (ld_g1 % 4 == 0) && (ld_dy % 4 == 0) && al_ok(g1, dt & 1) && al_ok(dy, dt & 8) &&
                     (!g2 || ((ld_g2 % 4 == 0) && al_ok(g2, dt & 2))) && (!a || ((ld_a % 4 == 0) && al_ok(a, dt & 4)));
    if (vec && (C % 8 == 0) && v8_ok(g1, ld_g1, dt & 1) && v8_ok(g2, ld_g2, dt & 2) && v8_ok(a, ld_a, dt & 4) && v8_ok(dy, ld_dy, dt & 8))
        hipLaunchKernelGGL(k_act_bwd<8>, dim3(ew_blocks(npix * (C / 8))), dim3(256), 0, st, tg1, ld_g1, tg2, ld_g2, ta, ld_a,
                           tdy, ld_dy, npix, C, act, drop_p, seed);
    else if (vec)
        hipLaunchKernelGGL(k_act_bwd<4>, dim3(ew_blocks(npix * (C / 4))), dim3(256), 0, st, tg1, ld_g1, tg2, ld_g2, ta, ld_a,
                           tdy, ld_dy, npix, C, act, drop_p, seed);
    else
        hipLaunchKernelGGL(k_act_bwd<1>, dim3(ew_blocks(npix * C)), dim3(256), 0, st, tg1, ld_g1, tg2, ld_g2, ta, ld_a, tdy,
                           ld_dy, npix, C, act, drop_p, seed);
    return pg_launch_status();
}

int pg_act_bwd(const float* g1, int ld_g1, const float* g2, int ld_g2, const float* a, int ld_a, float* dy,
               int ld_dy, long npix, int C, int act, float drop_p, uint64_t seed, void* stream) {
    return pg_act_bwd_t(g1, ld_g1, g2, ld_g2, a, ld_a, dy, ld_dy, npix, C, act, drop_p, seed, stream, 0);
}

int pg_softmax_fwd(const float* y, int ld_y, float* out, int ld_out, long npix, int C, void* stream) {
    if (!y || !out || npix <= 0 || C <= 0 || ld_y < C || ld_out < C) return PG_EINVAL;
    hipLaunchKernelGGL(k_softmax_fwd, dim3(ew_blocks(npix)), dim3(256), 0, (hipStream_t)stream, y, ld_y, out, ld_out, npix,
                       C);
    return pg_launch_status();
}

int pg_softmax_bwd(const float* g1, int ld_g1, const float* g2, int ld_g2, const float* out, int ld_out, float* dy,
                   int ld_dy, long npix, int C, void* stream) {
    if (!g1 || !out || !dy || npix <= 0 || C <= 0 || ld_g1 < C || ld_out < C || ld_dy < C || (g2 && ld_g2 < C))
        return PG_EINVAL;
    hipLaunchKernelGGL(k_softmax_bwd, dim3(ew_blocks(npix)), dim3(256), 0, (hipStream_t)stream, g1, ld_g1, g2, ld_g2, out,
                       ld_out, dy, ld_dy, npix, C);
    return pg_launch_status();
}

int pg_dropout_mask(float* mask, long nelem, float drop_p, uint64_t seed, void* stream) {
    if (!mask || nelem <= 0 || drop_p < 0.f || drop_p >= 1.f) return PG_EINVAL;
    hipLaunchKernelGGL(k_dropout_mask, dim3(ew_blocks(nelem)), dim3(256), 0, (hipStream_t)stream, mask, nelem, drop_p,
                       seed);
    return pg_launch_status();
}

}  // extern "C"
