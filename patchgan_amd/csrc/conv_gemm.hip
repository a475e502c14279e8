// conv_gemm.hip -- the three GEMM-shaped 4x4 convolution kernels of the patchGAN hot path for gfx950.
//
//   big2small : small[n,p,q,a]   = sum_{tap,b} big[n,s*p-1+kh,s*q-1+kw,b] * P[tap][a][b]      (conv fwd, convT dgrad)
//   small2big : big[n,h,w,b]     = sum_{tap,a} small[n,(h+1-kh)/s,(w+1-kw)/s,a] * P[tap][a][b]  (convT fwd, conv dgrad)
//   wgrad     : dP[tap][a][b]    = sum_{n,p,q} small[n,p,q,a] * big[n,s*p-1+kh,s*q-1+kw,b]     (both weight grads)
//
// Each is an im2col-free implicit GEMM: 256-thread workgroups (4 waves of 64), operand tiles staged
// global -> registers -> LDS (prefetch of chunk c+1 issued before the MFMAs of chunk c), fp32 MFMA
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD), deterministic split-K through slabs.
// NHWC: a K-chunk of 32 consecutive (tap, channel) indices is 128 contiguous bytes of one input pixel.
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l holds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// K-contiguous operands are read from LDS as one ds_read_b128 per lane per 8 k: lane (row, h) takes
// k = 8*kk + 4*h + j (j = 0..3) for MFMA j, the same permutation of K on both operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <alloca.h>
#include <new>
#include <vector>
#include "patchgan_hip.h"
#include "pg_common.h"
#include "conv_wino.h"
#include "conv_bf16.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Geom {
    int N, Hb, Wb, Hs, Ws, Ca, Cb, s;
};

constexpr int KC = 32;        // K-chunk (floats)
constexpr int LDK = KC + 4;   // LDS row pitch of a K-contiguous tile: 36 floats -> conflict-free ds_read_b128

__device__ __forceinline__ f32x4 ld4(const float* p, bool ok) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return ok ? v : z;
}

// ------------------------------------------------------------------------------------------------
// big2small: rows = small pixels (n,p,q), cols = a, K = (tap, b) with b fastest.
// ------------------------------------------------------------------------------------------------
template <int MR, int NR, int WM, int WN>
__global__ __launch_bounds__(256) void k_big2small(const float* __restrict__ big, int ld_big,
                                                   const float* __restrict__ P, float* __restrict__ out, int ld_out,
                                                   long slab_stride, Geom g, int chunks_per_slice, int veck,
                                                   const float* __restrict__ bias, int act) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int AI = BM / 32, BI = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    float* As = smem;
    float* Bs = smem + BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int M = g.N * g.Hs * g.Ws, K = 16 * g.Cb;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int nchunks = (K + KC - 1) / KC;
    const int c_begin = blockIdx.z * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    const int kq = tid & 7, r0 = tid >> 3;
    int a_noff[AI], a_h0[AI], a_w0[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        int m = m0 + r0 + 32 * i;
        if (m < M) {
            int n = m / (g.Hs * g.Ws);
            int rem = m - n * (g.Hs * g.Ws);
            int p = rem / g.Ws, q = rem - p * g.Ws;
            a_noff[i] = n * g.Hb * g.Wb;
            a_h0[i] = g.s * p - 1;
            a_w0[i] = g.s * q - 1;
        } else {
            a_noff[i] = 0;
            a_h0[i] = -1000000;
            a_w0[i] = 0;
        }
    }

    f32x4 ra[AI], rb[BI];
    // (tap, b) of this thread's float4 advance by KC per chunk: no division in the K loop when Cb >= KC
    int cur_tap = (c_begin * KC + kq * 4) / g.Cb;
    int cur_b = (c_begin * KC + kq * 4) - cur_tap * g.Cb;
    const bool inc_ok = g.Cb >= KC;
    auto load_chunk = [&](int c) {
        const int k = c * KC + kq * 4;
        if (veck) {
            int tap, b;
            if (inc_ok) {
                tap = cur_tap;
                b = cur_b;
                cur_b += KC;
                if (cur_b >= g.Cb) {
                    cur_b -= g.Cb;
                    ++cur_tap;
                }
            } else {
                tap = k / g.Cb;
                b = k - tap * g.Cb;
            }
            const int kh = tap >> 2, kw = tap & 3;
            const bool kok = k < K;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                int h = a_h0[i] + kh, w = a_w0[i] + kw;
                bool ok = kok && (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
                const float* p = ok ? big + ((long)(a_noff[i] + h * g.Wb + w) * ld_big + b) : big;
                ra[i] = ld4(p, ok);
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                int a = n0 + r0 + 32 * i;
                bool ok = kok && a < g.Ca;
                const float* p = ok ? P + ((long)(tap * g.Ca + a) * g.Cb + b) : P;
                rb[i] = ld4(p, ok);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = k + e;
                const int tap = ke / g.Cb, b = ke - tap * g.Cb;
                const int kh = tap >> 2, kw = tap & 3;
                const bool kok = ke < K;
#pragma unroll
                for (int i = 0; i < AI; ++i) {
                    int h = a_h0[i] + kh, w = a_w0[i] + kw;
                    bool ok = kok && (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
                    const float* p = ok ? big + ((long)(a_noff[i] + h * g.Wb + w) * ld_big + b) : big;
                    float v = *p;
                    ra[i][e] = ok ? v : 0.f;
                }
#pragma unroll
                for (int i = 0; i < BI; ++i) {
                    int a = n0 + r0 + 32 * i;
                    bool ok = kok && a < g.Ca;
                    const float* p = ok ? P + ((long)(tap * g.Ca + a) * g.Cb + b) : P;
                    float v = *p;
                    rb[i][e] = ok ? v : 0.f;
                }
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(r0 + 32 * i) * LDK + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(r0 + 32 * i) * LDK + kq * 4]) = rb[i];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        if (more) load_chunk(c + 1);
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
            f32x4 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MR + i) * 32 + lrow) * LDK + kk * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * NR + j) * 32 + lrow) * LDK + kk * 8 + lh * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    float* o = out + (long)blockIdx.z * slab_stride;
    const bool fin = (slab_stride == 0);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
            const float bv = (fin && bias != nullptr && col < g.Ca) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + (wm * MR + i) * 32 + row;
                if (m < M && col < g.Ca) {
                    float v = acc[i][j][r];
                    if (fin) v = pg_act(v + bv, act);
                    o[(long)m * ld_out + col] = v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// small2big: per output-parity class (stride 2: 4 classes of 2x2 taps; stride 1: 1 class of 4x4 taps):
// rows = big pixels of the class, cols = b, K = (tloc, a) with a fastest.  A is K-contiguous (small).
// P[tap][a][:] rows are N-contiguous in memory: each lane owns one column n and fetches 4 consecutive k with
// scalar loads (a wave-load is 64 consecutive floats of one weight row), so the tile lands K-contiguous in
// LDS and the MFMA loop is the same ds_read_b128 loop as big2small.
// ------------------------------------------------------------------------------------------------
template <int MR, int NR, int WM, int WN>
__global__ __launch_bounds__(256) void k_small2big(const float* __restrict__ small, int ld_small,
                                                   const float* __restrict__ P, float* __restrict__ out, int ld_out,
                                                   long slab_stride, Geom g, int chunks_per_slice, int veck, int vecn,
                                                   const float* __restrict__ bias, int act) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int AI = BM / 32;
    constexpr int BG = 256 / BN;        // thread groups along k
    constexpr int NQ = 8 / BG;          // k-quads per thread per chunk
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    float* As = smem;
    float* Bs = smem + BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;

    const int ncls = (g.s == 2) ? 4 : 1;
    const int cls = blockIdx.z % ncls, slice = blockIdx.z / ncls;
    const int ah = (g.s == 2) ? (cls >> 1) : 0, aw = (g.s == 2) ? (cls & 1) : 0;
    const int T = (g.s == 2) ? 2 : 4;                    // taps per dim in this class
    const int Hc = (g.s == 2) ? (g.Hb - ah + 1) / 2 : g.Hb;
    const int Wc = (g.s == 2) ? (g.Wb - aw + 1) / 2 : g.Wb;
    const int kh0 = (g.s == 2) ? (1 - ah) : 0, kw0 = (g.s == 2) ? (1 - aw) : 0;
    const int Mc = g.N * Hc * Wc, K = T * T * g.Ca;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= Mc) return;   // uniform per block (classes can differ in size for odd Hb/Wb)
    const int nchunks = (K + KC - 1) / KC;
    const int c_begin = slice * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    const int kq = tid & 7, r0 = tid >> 3;
    int a_noff[AI], a_ib[AI], a_jb[AI];   // small-pixel base: ih = ib - th, iw = jb - tw
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        int m = m0 + r0 + 32 * i;
        if (m < Mc) {
            int n = m / (Hc * Wc);
            int rem = m - n * (Hc * Wc);
            int ii = rem / Wc, jj = rem - ii * Wc;
            a_noff[i] = n * g.Hs * g.Ws;
            a_ib[i] = (g.s == 2) ? ii + ah : ii + 1;
            a_jb[i] = (g.s == 2) ? jj + aw : jj + 1;
        } else {
            a_noff[i] = 0;
            a_ib[i] = -1000000;
            a_jb[i] = 0;
        }
    }
    const int bn = tid % BN, bg = tid / BN;       // this thread's B column and k-group
    const int ncol = n0 + bn;
    const bool n_ok = ncol < g.Cb;

    // incremental (tloc, a) for the A float4 and for each B quad (valid when Ca >= KC and Ca % 4 == 0)
    const bool inc_ok = veck && g.Ca >= KC;
    int a_tl = (c_begin * KC + kq * 4) / g.Ca;
    int a_a = (c_begin * KC + kq * 4) - a_tl * g.Ca;
    int b_tl[NQ], b_a[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int k = c_begin * KC + 4 * (bg + BG * i);
        b_tl[i] = k / g.Ca;
        b_a[i] = k - b_tl[i] * g.Ca;
    }

    f32x4 ra[AI], rb[NQ];
    auto load_chunk = [&](int c) {
        const int k = c * KC + kq * 4;
        if (veck) {
            int tl, a;
            if (inc_ok) {
                tl = a_tl;
                a = a_a;
                a_a += KC;
                if (a_a >= g.Ca) {
                    a_a -= g.Ca;
                    ++a_tl;
                }
            } else {
                tl = k / g.Ca;
                a = k - tl * g.Ca;
            }
            const int th = tl / T, tw = tl - th * T;
            const bool kok = k < K;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                int ih = a_ib[i] - th, iw = a_jb[i] - tw;
                bool ok = kok && (unsigned)ih < (unsigned)g.Hs && (unsigned)iw < (unsigned)g.Ws;
                const float* p = ok ? small + ((long)(a_noff[i] + ih * g.Ws + iw) * ld_small + a) : small;
                ra[i] = ld4(p, ok);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = k + e;
                const int tl = ke / g.Ca, a = ke - tl * g.Ca;
                const int th = tl / T, tw = tl - th * T;
                const bool kok = ke < K;
#pragma unroll
                for (int i = 0; i < AI; ++i) {
                    int ih = a_ib[i] - th, iw = a_jb[i] - tw;
                    bool ok = kok && (unsigned)ih < (unsigned)g.Hs && (unsigned)iw < (unsigned)g.Ws;
                    const float* p = ok ? small + ((long)(a_noff[i] + ih * g.Ws + iw) * ld_small + a) : small;
                    float v = *p;
                    ra[i][e] = ok ? v : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int kb = c * KC + 4 * (bg + BG * i);
            if (inc_ok) {
                const int tl = b_tl[i], a = b_a[i];
                b_a[i] += KC;
                if (b_a[i] >= g.Ca) {
                    b_a[i] -= g.Ca;
                    ++b_tl[i];
                }
                const int th = tl / T, tw = tl - th * T;
                const int tap = (kh0 + g.s * th) * 4 + (kw0 + g.s * tw);
                const bool ok = n_ok && kb < K;
                const float* col = ok ? P + (long)(tap * g.Ca + a) * g.Cb + ncol : P;
                const long st = ok ? g.Cb : 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = col[e * st];
                    rb[i][e] = ok ? v : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ke = kb + e;
                    const int tl = ke / g.Ca, a = ke - tl * g.Ca;
                    const int th = tl / T, tw = tl - th * T;
                    const int tap = (kh0 + g.s * th) * 4 + (kw0 + g.s * tw);
                    const bool ok = n_ok && ke < K;
                    float v = *(ok ? P + (long)(tap * g.Ca + a) * g.Cb + ncol : P);
                    rb[i][e] = ok ? v : 0.f;
                }
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(r0 + 32 * i) * LDK + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NQ; ++i) *reinterpret_cast<f32x4*>(&Bs[bn * LDK + 4 * (bg + BG * i)]) = rb[i];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        if (more) load_chunk(c + 1);
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
            f32x4 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MR + i) * 32 + lrow) * LDK + kk * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * NR + j) * 32 + lrow) * LDK + kk * 8 + lh * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    // epilogue: row m of the class -> big pixel (n, s*ii+ah, s*jj+aw); slabs mirror the [N,Hb,Wb,Cb] layout.
    const bool fin = (slab_stride == 0);
    float* o = out + (long)slice * slab_stride;
    const int ldo = fin ? ld_out : g.Cb;
    float bv[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int col = n0 + (wn * NR + j) * 32 + lrow;
        bv[j] = (fin && bias != nullptr && col < g.Cb) ? bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        // rows of this lane: base + {0..3} + 8*{0..3} + 4*lh ; decode the first, then step (the 16 rows of a lane
        // are at most 28 apart, so a carry chain in (jj, ii, n) is cheaper than 16 divisions)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int m = m0 + (wm * MR + i) * 32 + row;
            if (m < Mc) {
                int n = m / (Hc * Wc);
                int rem = m - n * (Hc * Wc);
                int ii = rem / Wc, jj = rem - ii * Wc;
                int h = (g.s == 2) ? 2 * ii + ah : ii, w = (g.s == 2) ? 2 * jj + aw : jj;
                float* orow = o + (long)((n * g.Hb + h) * g.Wb + w) * ldo;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int col = n0 + (wn * NR + j) * 32 + lrow;
                    if (col < g.Cb) {
                        float v = acc[i][j][r];
                        if (fin) v = pg_act(v + bv[j], act);
                        orow[col] = v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad: per tap, rows = a, cols = b, K = small pixels.  Both operands are contiguous along their
// non-K dim, so LDS tiles are [k][m] / [k][n] and fragments are single ds_read_b32 (natural K order).
// grid: x = tilesA*tilesB, y = 16 taps, z = K slices.  Output slab z is [16][Ca][Cb].
// ------------------------------------------------------------------------------------------------
template <int MR, int NR, int WM, int WN>
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ small, int ld_small,
                                               const float* __restrict__ big, int ld_big,
                                               float* __restrict__ out, long slab_stride, Geom g,
                                               int chunks_per_slice, int tilesB, int vecm, int vecn) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int AQ = BM / 4, AROWS = 256 / AQ, AI = KC / AROWS;
    constexpr int BQ = BN / 4, BROWS = 256 / BQ, BI = KC / BROWS;
    __shared__ __attribute__((aligned(16))) float smem[KC * LDA + KC * LDB];
    float* As = smem;
    float* Bs = smem + KC * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int tile_a = blockIdx.x / tilesB, tile_b = blockIdx.x % tilesB;
    const int m0 = tile_a * BM, n0 = tile_b * BN;
    const int tap = blockIdx.y, kh = tap >> 2, kw = tap & 3;
    const int Kp = g.N * g.Hs * g.Ws;   // pixels
    const int nchunks = (Kp + KC - 1) / KC;
    const int c_begin = blockIdx.z * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    const int aq = tid % AQ, arow0 = tid / AQ;
    const int bq = tid % BQ, brow0 = tid / BQ;
    const bool inc_ok = g.Ws >= 16 && g.Hs >= 2;
    int r_n[BI], r_p[BI], r_q[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int pix = c_begin * KC + brow0 + BROWS * i;
        r_n[i] = pix / (g.Hs * g.Ws);
        const int rem = pix - r_n[i] * (g.Hs * g.Ws);
        r_p[i] = rem / g.Ws;
        r_q[i] = rem - r_p[i] * g.Ws;
    }

    f32x4 ra[AI], rb[BI];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int pix = c * KC + arow0 + AROWS * i;
            const int a = m0 + aq * 4;
            const bool pok = pix < Kp;
            const float* row = small + (long)pix * ld_small;
            if (vecm) {
                bool ok = pok && a < g.Ca;
                ra[i] = ld4(ok ? row + a : small, ok);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bool ok = pok && (a + e) < g.Ca;
                    float v = *(ok ? row + a + e : small);
                    ra[i][e] = ok ? v : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int pix = c * KC + brow0 + BROWS * i;
            const int b = n0 + bq * 4;
            bool pok = pix < Kp;
            int n, p, q;
            if (inc_ok) {   // carry chain instead of two divisions per row and chunk
                n = r_n[i];
                p = r_p[i];
                q = r_q[i];
                r_q[i] += KC;
                while (r_q[i] >= g.Ws) {
                    r_q[i] -= g.Ws;
                    ++r_p[i];
                }
                while (r_p[i] >= g.Hs) {
                    r_p[i] -= g.Hs;
                    ++r_n[i];
                }
            } else {
                n = pix / (g.Hs * g.Ws);
                int rem = pix - n * (g.Hs * g.Ws);
                p = rem / g.Ws;
                q = rem - p * g.Ws;
            }
            int h = g.s * p - 1 + kh, w = g.s * q - 1 + kw;
            pok = pok && (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
            const float* row = big + (long)((n * g.Hb + h) * g.Wb + w) * ld_big;
            if (vecn) {
                bool ok = pok && b < g.Cb;
                rb[i] = ld4(ok ? row + b : big, ok);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bool ok = pok && (b + e) < g.Cb;
                    float v = *(ok ? row + b + e : big);
                    rb[i][e] = ok ? v : 0.f;
                }
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(arow0 + AROWS * i) * LDA + aq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(brow0 + BROWS * i) * LDB + bq * 4]) = rb[i];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        if (more) load_chunk(c + 1);
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            float af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i] = As[(kk * 2 + lh) * LDA + (wm * MR + i) * 32 + lrow];
#pragma unroll
            for (int j = 0; j < NR; ++j) bf[j] = Bs[(kk * 2 + lh) * LDB + (wn * NR + j) * 32 + lrow];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    float* o = out + (long)blockIdx.z * slab_stride + (long)tap * g.Ca * g.Cb;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int a = m0 + (wm * MR + i) * 32 + row;
                if (a < g.Ca && col < g.Cb) o[(long)a * g.Cb + col] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// wgrad with the 16 taps folded into the GEMM N dimension, for layers with very few channels on one side
// (the image-facing layers: enc0 / d0 have Cb = 3..10, dec6 has Cb = output_nc, the D head has Ca = 1).  The
// per-tap kernel above would pad those channels to a 32-wide MFMA tile 16 times over; here N = 16*C is dense.
//   MODE 1 (Cb <= 8): rows m = a, cols n = (tap, b), K = small pixels
//        A[k][m] = small[pix][a]              B[k][n] = big[img, s*p-1+kh, s*q-1+kw][b]
//   MODE 2 (Ca <= 8): rows m = b, cols n = (tap, a), K = big pixels
//        A[k][m] = big[pix][b]                B[k][n] = small[img, (h+1-kh)/s, (w+1-kw)/s][a]
// Output element (tap, a, b) -> slab[(tap*Ca + a)*Cb + b].  X = the row tensor, Y = the gathered tensor.
// ------------------------------------------------------------------------------------------------
template <int MR, int NR, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void k_wgrad_tapn(const float* __restrict__ X, int ld_x,
                                                    const float* __restrict__ Y, int ld_y, float* __restrict__ out,
                                                    long slab_stride, Geom g, int chunks_per_slice, int tilesN,
                                                    int vecm, int vecy) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int AQ = BM / 4, AROWS = 256 / AQ, AI = KC / AROWS;
    constexpr int BQ = BN / 4, BROWS = 256 / BQ, BI = KC / BROWS;
    __shared__ __attribute__((aligned(16))) float smem[KC * LDA + KC * LDB];
    float* As = smem;
    float* Bs = smem + KC * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int tile_m = blockIdx.x / tilesN, tile_n = blockIdx.x % tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int Cx = (MODE == 1) ? g.Ca : g.Cb;            // channels of the row tensor  (GEMM M)
    const int Cy = (MODE == 1) ? g.Cb : g.Ca;            // channels of the gathered tensor
    const int Hx = (MODE == 1) ? g.Hs : g.Hb, Wx = (MODE == 1) ? g.Ws : g.Wb;
    const int Ndim = 16 * Cy;
    const int Kp = g.N * Hx * Wx;
    const int nchunks = (Kp + KC - 1) / KC;
    const int c_begin = blockIdx.z * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    const int aq = tid % AQ, arow0 = tid / AQ;
    const int bq = tid % BQ, brow0 = tid / BQ;
    // this thread's four gathered columns are fixed for the whole K loop
    int e_kh[4], e_kw[4], e_c[4];
    bool e_ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = n0 + bq * 4 + e;
        const int tap = n / Cy;
        e_ok[e] = n < Ndim;
        e_c[e] = n - tap * Cy;
        e_kh[e] = tap >> 2;
        e_kw[e] = tap & 3;
    }

    const bool inc_ok = Wx >= 16 && Hx >= 2;
    int r_n[BI], r_p[BI], r_q[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int pix = c_begin * KC + brow0 + BROWS * i;
        r_n[i] = pix / (Hx * Wx);
        const int rem = pix - r_n[i] * (Hx * Wx);
        r_p[i] = rem / Wx;
        r_q[i] = rem - r_p[i] * Wx;
    }

    f32x4 ra[AI], rb[BI];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int pix = c * KC + arow0 + AROWS * i;
            const int m = m0 + aq * 4;
            const bool pok = pix < Kp;
            const float* row = X + (long)pix * ld_x;
            if (vecm) {
                bool ok = pok && m < Cx;
                ra[i] = ld4(ok ? row + m : X, ok);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bool ok = pok && (m + e) < Cx;
                    float v = *(ok ? row + m + e : X);
                    ra[i][e] = ok ? v : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int pix = c * KC + brow0 + BROWS * i;
            const bool pok = pix < Kp;
            int img, r, cc;
            if (inc_ok) {
                img = r_n[i];
                r = r_p[i];
                cc = r_q[i];
                r_q[i] += KC;
                while (r_q[i] >= Wx) {
                    r_q[i] -= Wx;
                    ++r_p[i];
                }
                while (r_p[i] >= Hx) {
                    r_p[i] -= Hx;
                    ++r_n[i];
                }
            } else {
                img = pix / (Hx * Wx);
                const int rem = pix - img * (Hx * Wx);
                r = rem / Wx;
                cc = rem - r * Wx;
            }
            if (MODE == 1 && vecy) {
                // Cb == 4: this thread's four columns are the four channels of ONE tap: one 16-byte load
                const int h = g.s * r - 1 + e_kh[0], w = g.s * cc - 1 + e_kw[0];
                const bool ok = pok && e_ok[0] && (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
                rb[i] = ld4(ok ? Y + (long)((img * g.Hb + h) * g.Wb + w) * ld_y : Y, ok);
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool ok = pok && e_ok[e];
                long off;
                if (MODE == 1) {
                    const int h = g.s * r - 1 + e_kh[e], w = g.s * cc - 1 + e_kw[e];
                    ok = ok && (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
                    off = (long)((img * g.Hb + h) * g.Wb + w) * ld_y + e_c[e];
                } else {
                    const int hh = r + 1 - e_kh[e], ww = cc + 1 - e_kw[e];
                    ok = ok && hh >= 0 && ww >= 0;
                    int p = hh, q = ww;
                    if (g.s == 2) {
                        ok = ok && ((hh & 1) == 0) && ((ww & 1) == 0);
                        p = hh >> 1;
                        q = ww >> 1;
                    }
                    ok = ok && p < g.Hs && q < g.Ws;
                    off = (long)((img * g.Hs + p) * g.Ws + q) * ld_y + e_c[e];
                }
                float v = *(ok ? Y + off : Y);
                rb[i][e] = ok ? v : 0.f;
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(arow0 + AROWS * i) * LDA + aq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(brow0 + BROWS * i) * LDB + bq * 4]) = rb[i];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        if (more) load_chunk(c + 1);
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            float af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i] = As[(kk * 2 + lh) * LDA + (wm * MR + i) * 32 + lrow];
#pragma unroll
            for (int j = 0; j < NR; ++j) bf[j] = Bs[(kk * 2 + lh) * LDB + (wn * NR + j) * 32 + lrow];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    float* o = out + (long)blockIdx.z * slab_stride;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int n = n0 + (wn * NR + j) * 32 + lrow;
        if (n >= Ndim) continue;
        const int tap = n / Cy, cy = n - tap * Cy;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + (wm * MR + i) * 32 + row;
                if (m < Cx) {
                    const long idx = (MODE == 1) ? ((long)(tap * g.Ca + m) * g.Cb + cy) : ((long)(tap * g.Ca + cy) * g.Cb + m);
                    o[idx] = acc[i][j][r];
                }
            }
    }
}

// ================================================================================================
// Fast variants (the hot path whenever channels are multiples of 4, buffers are 16-byte aligned and every
// tensor is < 2 GiB): same tiling and MFMA loop as the generic kernels above, but
//   * operand gathers are raw buffer loads (hardware range check: an out-of-range offset returns 0), so the
//     zero padding, ragged tiles and K tails cost one v_cndmask instead of a divergent branch per load,
//   * all offsets are 32-bit element offsets from a per-row base computed once per tile; tap validity is a
//     16-bit mask per row (one shift+and per load),
//   * the loads of chunk c+1 are spread over the four 8-k steps of chunk c, inside ONE basic block, so the
//     scheduler interleaves their address arithmetic with the MFMAs instead of running it ahead of them.
// ================================================================================================
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = 0x7fffffff;
// Scheduling fence that only vector-memory instructions may not cross (LLVM sched_barrier mask: ALU | VALU | SALU |
// DS | DS-read | DS-write may; MFMA and VMEM may not).  Without it hipcc sinks the next chunk's buffer loads to the END of the MFMA
// sequence (to shorten their destination registers' live ranges), so their latency lands on the barrier instead of
// under 48 MFMAs.
#define PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)
// element offset -> byte offset without signed overflow (sentinel offsets exceed INT_MAX/4 on purpose)
__device__ __forceinline__ int b4(int elem_off) { return (int)((unsigned)elem_off << 2); }
__device__ __forceinline__ int xcd_remap(int bid, int nblk) { return pg_xcd_remap(bid, nblk); }   // pg_common.h

// byte offset of a gather element, forced out of range (bit 31 set: >= 2 GiB > any descriptor here) when !ok.
// Written as an OR so the offset arithmetic stays unconditional: with `ok ? off : OOB` hipcc sinks the arithmetic into
// an exec-masked region per load, which splits the MFMA loop into many scheduling regions.
__device__ __forceinline__ int voff(int elem_off, bool ok) {
    return (int)(((unsigned)elem_off << 2) | (ok ? 0u : 0x80000000u));
}

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

__device__ __noinline__ float pg_act_slow(float v, int act) { return pg_act(v, act); }
__device__ __forceinline__ float pg_act_epi(float v, int act) {
    if (act == PG_ACT_NONE) return v;
    if (act == PG_ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    if (act == PG_ACT_RELU) return v > 0.f ? v : 0.f;
    return pg_act_slow(v, act);
}

// ONE = true turns the kernel into a plain row GEMM  out[m][a] = sum_b big[m][b] * P[a][b]  (a 1x1 "convolution" over
// the N*Hb*Wb pixels of `big`): the first half of the taps-folded-into-N path for layers with <= 8 channels on the
// output side, whose second half is a col2im / tap-gather pass (k_col2im_small2big, k_gather_big2small).
template <int MR, int NR, int WM, int WN, bool ONE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_b2s_fast(const float* __restrict__ big, int ld_big,
                                                  const float* __restrict__ P, float* __restrict__ out, int ld_out,
                                                  long slab_stride, Geom g, int chunks_per_slice,
                                                  const float* __restrict__ bias, int act, int big_bytes, int p_bytes) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int AI = BM / 32, BI = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    float* As = smem;
    float* Bs = smem + BM * LDK;
    const __amdgpu_buffer_rsrc_t rbig = __builtin_amdgcn_make_buffer_rsrc((void*)big, 0, big_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)P, 0, p_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int M = ONE ? g.N * g.Hb * g.Wb : g.N * g.Hs * g.Ws;
    const int K = ONE ? g.Cb : 16 * g.Cb;               // K % 32 == 0 (Cb % 4 == 0; ONE requires Cb % 32 == 0)
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
    const int nchunks = K / KC;
    const int c_begin = blockIdx.z * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    const int kq = tid & 7, r0 = tid >> 3;
    int a_off[AI], a_mask[AI], b_off[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + 32 * i;
        const int mm = min(m, M - 1);
        if (ONE) {
            a_off[i] = mm * ld_big;
            a_mask[i] = (m < M) ? 1 : 0;
            continue;
        }
        const int n = mm / (g.Hs * g.Ws);
        const int rem = mm - n * (g.Hs * g.Ws);
        const int p = rem / g.Ws, q = rem - p * g.Ws;
        const int h0 = g.s * p - 1, w0 = g.s * q - 1;
        a_off[i] = ((n * g.Hb + h0) * g.Wb + w0) * ld_big;
        int wv = 0, mask = 0;          // tap (kh, kw) is inside the image iff row kh and column kw both are
#pragma unroll
        for (int t = 0; t < 4; ++t) wv |= ((unsigned)(w0 + t) < (unsigned)g.Wb) ? (1 << t) : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) mask |= ((unsigned)(h0 + t) < (unsigned)g.Hb) ? (wv << (4 * t)) : 0;
        a_mask[i] = (m < M) ? mask : 0;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int a = n0 + r0 + 32 * i;
        b_off[i] = (a < g.Ca) ? a * g.Cb : 0x10000000;
    }
    const int CaCb = g.Ca * g.Cb;
    int cur_tap = (c_begin * KC + kq * 4) / g.Cb;
    int cur_b = (c_begin * KC + kq * 4) - cur_tap * g.Cb;

    f32x4 ra[AI], rb[BI];
    int tapoff = 0, pboff = 0, ctap = 0;
    auto next_tap = [&]() {   // (tap, b) of the next chunk for this thread's float4; Cb >= KC: at most one wrap
        const int tap = cur_tap, b = cur_b;
        cur_b += KC;
        const bool wrap = cur_b >= g.Cb;
        cur_b = wrap ? cur_b - g.Cb : cur_b;
        cur_tap = wrap ? cur_tap + 1 : cur_tap;
        ctap = tap;
        tapoff = ((tap >> 2) * g.Wb + (tap & 3)) * ld_big + b;
        pboff = tap * CaCb + b;
    };
    auto load_a = [&](int i, bool on) {
        const bool ok = on && ((a_mask[i] >> ctap) & 1);
        ra[i] = bload4(rbig, voff(a_off[i] + tapoff, ok));
    };
    auto load_b = [&](int i, bool on) { rb[i] = bload4(rP, voff(b_off[i] + pboff, on)); };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(r0 + 32 * i) * LDK + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(r0 + 32 * i) * LDK + kq * 4]) = rb[i];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        next_tap();
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(i, true);
#pragma unroll
        for (int i = 0; i < BI; ++i) load_b(i, true);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        next_tap();
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
            f32x4 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MR + i) * 32 + lrow) * LDK + kk * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * NR + j) * 32 + lrow) * LDK + kk * 8 + lh * 4]);
            if (kk < AI) load_a(kk, more);
            if (kk < BI) load_b(kk, more);
            PIN_VMEM();
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    float* o = out + (long)blockIdx.z * slab_stride;
    const bool fin = (slab_stride == 0);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
            const float bv = (fin && bias != nullptr && col < g.Ca) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + (wm * MR + i) * 32 + row;
                if (m < M && col < g.Ca) {
                    float v = acc[i][j][r];
                    if (fin) v = pg_act_epi(v + bv, act);
                    o[(long)m * ld_out + col] = v;
                }
            }
        }
}

template <int MR, int NR, int WM, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_s2b_fast(const float* __restrict__ small, int ld_small,
                                                  const float* __restrict__ P, float* __restrict__ out, int ld_out,
                                                  long slab_stride, Geom g, int chunks_per_slice,
                                                  const float* __restrict__ bias, int act, int small_bytes,
                                                  int p_bytes) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int AI = BM / 32;
    constexpr int BG = 256 / BN;        // thread groups along k
    constexpr int NQ = 8 / BG;          // k-quads per thread per chunk
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    float* As = smem;
    float* Bs = smem + BM * LDK;
    const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)small, 0, small_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)P, 0, p_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;

    const int ncls = (g.s == 2) ? 4 : 1;
    const int cls = blockIdx.z % ncls, slice = blockIdx.z / ncls;
    const int ah = (g.s == 2) ? (cls >> 1) : 0, aw = (g.s == 2) ? (cls & 1) : 0;
    const int T = (g.s == 2) ? 2 : 4, Tsh = (g.s == 2) ? 1 : 2;
    const int Hc = (g.s == 2) ? (g.Hb - ah + 1) / 2 : g.Hb;
    const int Wc = (g.s == 2) ? (g.Wb - aw + 1) / 2 : g.Wb;
    const int kh0 = (g.s == 2) ? (1 - ah) : 0, kw0 = (g.s == 2) ? (1 - aw) : 0;
    const int Mc = g.N * Hc * Wc, K = T * T * g.Ca;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
    if (m0 >= Mc) return;
    const int nchunks = (K + KC - 1) / KC;
    const int c_begin = slice * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    const int kq = tid & 7, r0 = tid >> 3;
    int a_off[AI], a_mask[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + 32 * i;
        const int mm = min(m, Mc - 1);
        const int n = mm / (Hc * Wc);
        const int rem = mm - n * (Hc * Wc);
        const int ii = rem / Wc, jj = rem - ii * Wc;
        const int ib = (g.s == 2) ? ii + ah : ii + 1, jb = (g.s == 2) ? jj + aw : jj + 1;
        a_off[i] = ((n * g.Hs + ib) * g.Ws + jb) * ld_small;
        int wv = 0, mask = 0;          // local tap (th, tw) reads small pixel (ib - th, jb - tw)
#pragma unroll
        for (int t = 0; t < 4; ++t) wv |= (t < T && (unsigned)(jb - t) < (unsigned)g.Ws) ? (1 << t) : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) mask |= (t < T && (unsigned)(ib - t) < (unsigned)g.Hs) ? (wv << (T * t)) : 0;
        a_mask[i] = (m < Mc) ? mask : 0;
    }
    const int bn = tid % BN, bg = tid / BN;
    const int ncol = n0 + bn;
    const int ncol_off = (ncol < g.Cb) ? ncol : 0x10000000;
    const int CaCb = g.Ca * g.Cb;

    // incremental (tloc, a) per k-quad: A float4 (index 0) and the NQ quads of B (1..NQ); Ca >= KC: <= 1 wrap
    int q_tl[NQ + 1], q_a[NQ + 1];
    {
        const int k = c_begin * KC + kq * 4;
        q_tl[0] = k / g.Ca;
        q_a[0] = k - q_tl[0] * g.Ca;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int k = c_begin * KC + 4 * (bg + BG * i);
        q_tl[i + 1] = k / g.Ca;
        q_a[i + 1] = k - q_tl[i + 1] * g.Ca;
    }
    auto quad = [&](int qi, int& tl, int& a) {   // returns (tloc, a) of quad qi for the chunk being loaded
        tl = q_tl[qi];
        a = q_a[qi];
        const int na = a + KC;
        const bool wrap = na >= g.Ca;
        q_a[qi] = wrap ? na - g.Ca : na;
        q_tl[qi] = wrap ? tl + 1 : tl;
    };

    f32x4 ra[AI], rb[NQ];
    int a_tl = 0, a_koff = OOB;
    auto next_a = [&](int c) {
        const int k = c * KC + kq * 4;
        int tl, a;
        quad(0, tl, a);
        a_tl = tl;
        const int th = tl >> Tsh, tw = tl & (T - 1);
        a_koff = (k < K) ? (a - (th * g.Ws + tw) * ld_small) : 0x20000000;
    };
    auto load_a = [&](int i, bool on) {
        const bool ok = on && ((a_mask[i] >> a_tl) & 1) && (a_koff < 0x10000000);
        ra[i] = bload4(rsm, voff(a_off[i] + a_koff, ok));
    };
    auto load_b = [&](int i, int c, bool on) {
        const int kb = c * KC + 4 * (bg + BG * i);
        int tl, a;
        quad(i + 1, tl, a);
        if (BN >= 64) {   // a wave shares one k-quad (bg = tid / BN is wave-uniform): keep its decode on the scalar unit
            tl = __builtin_amdgcn_readfirstlane(tl);
            a = __builtin_amdgcn_readfirstlane(a);
        }
        const int th = tl >> Tsh, tw = tl & (T - 1);
        const int tap = (kh0 + g.s * th) * 4 + (kw0 + g.s * tw);
        const bool ok = on && kb < K;
        const int rowoff = tap * CaCb + a * g.Cb;
        const int base = voff(rowoff + ncol_off, ok);
        const int st = g.Cb * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[i][e] = bload1(rP, base + e * st);
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(r0 + 32 * i) * LDK + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NQ; ++i) *reinterpret_cast<f32x4*>(&Bs[bn * LDK + 4 * (bg + BG * i)]) = rb[i];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        next_a(c_begin);
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(i, true);
#pragma unroll
        for (int i = 0; i < NQ; ++i) load_b(i, c_begin, true);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        next_a(c + 1);
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
            f32x4 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MR + i) * 32 + lrow) * LDK + kk * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * NR + j) * 32 + lrow) * LDK + kk * 8 + lh * 4]);
            if (kk < AI) load_a(kk, more);
            if (kk < NQ) load_b(kk, c + 1, more);
            PIN_VMEM();
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    const bool fin = (slab_stride == 0);
    float* o = out + (long)slice * slab_stride;
    const int ldo = fin ? ld_out : g.Cb;
    float bv[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int col = n0 + (wn * NR + j) * 32 + lrow;
        bv[j] = (fin && bias != nullptr && col < g.Cb) ? bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        // decode the lane's first row with two divisions, then walk the other 15 (at most 28 rows further) with a
        // carry chain: 64 divisions per thread in the epilogue cost as much as several K-chunks
        const int mb = m0 + (wm * MR + i) * 32 + 4 * lh;
        const int mbc = min(mb, Mc - 1);
        const int nb0 = mbc / (Hc * Wc);
        const int remb = mbc - nb0 * (Hc * Wc);
        const int ib0 = remb / Wc, jb0 = remb - ib0 * Wc;
        const bool chain = Wc >= 16 && Hc >= 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int m = m0 + (wm * MR + i) * 32 + row;
            if (m < Mc) {
                int n, ii, jj;
                if (chain) {
                    const int d = (r & 3) + 8 * (r >> 2);      // 0..27 rows past the decoded one
                    jj = jb0 + d;
                    ii = ib0;
                    n = nb0;
                    bool w1 = jj >= Wc;
                    jj = w1 ? jj - Wc : jj;
                    ii = w1 ? ii + 1 : ii;
                    w1 = jj >= Wc;
                    jj = w1 ? jj - Wc : jj;
                    ii = w1 ? ii + 1 : ii;
                    w1 = ii >= Hc;
                    ii = w1 ? ii - Hc : ii;
                    n = w1 ? n + 1 : n;
                } else {
                    n = m / (Hc * Wc);
                    const int rem = m - n * (Hc * Wc);
                    ii = rem / Wc;
                    jj = rem - ii * Wc;
                }
                const int h = (g.s == 2) ? 2 * ii + ah : ii, w = (g.s == 2) ? 2 * jj + aw : jj;
                float* orow = o + (long)((n * g.Hb + h) * g.Wb + w) * ldo;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int col = n0 + (wn * NR + j) * 32 + lrow;
                    if (col < g.Cb) {
                        float v = acc[i][j][r];
                        if (fin) v = pg_act_epi(v + bv[j], act);
                        orow[col] = v;
                    }
                }
            }
        }
    }
}

template <int MR, int NR, int WM, int WN, bool POW2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wgrad_fast(const float* __restrict__ small, int ld_small,
                                                    const float* __restrict__ big, int ld_big,
                                                    float* __restrict__ out, long slab_stride, Geom g,
                                                    int chunks_per_slice, int tilesB, int small_bytes, int big_bytes) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int AQ = BM / 4, AROWS = 256 / AQ, AI = KC / AROWS;
    constexpr int BQ = BN / 4, BROWS = 256 / BQ, BI = KC / BROWS;
    __shared__ __attribute__((aligned(16))) float smem[KC * LDA + KC * LDB];
    float* As = smem;
    float* Bs = smem + KC * LDA;
    const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)small, 0, small_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbig = __builtin_amdgcn_make_buffer_rsrc((void*)big, 0, big_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int tile_a = blockIdx.x / tilesB, tile_b = blockIdx.x % tilesB;
    const int m0 = tile_a * BM, n0 = tile_b * BN;
    const int tap = blockIdx.y, kh = tap >> 2, kw = tap & 3;
    const int Kp = g.N * g.Hs * g.Ws;
    const int nchunks = (Kp + KC - 1) / KC;
    const int c_begin = blockIdx.z * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    const int aq = tid % AQ, arow0 = tid / AQ;
    const int bq = tid % BQ, brow0 = tid / BQ;
    const int a_col = (m0 + aq * 4 < g.Ca) ? m0 + aq * 4 : 0x10000000;      // small rows beyond Kp are out of range by
    const int b_col = (n0 + bq * 4 < g.Cb) ? n0 + bq * 4 : 0x10000000;      // construction of small_bytes
    // pixel -> (img, p, q): shifts when Hs, Ws are powers of two (every UNet layer), else a branch-free carry chain
    // (needs Ws >= 16 and Hs >= 2: a step of KC = 32 pixels wraps q at most twice and p at most once)
    const int lgW = 31 - __builtin_clz(g.Ws), lgH = 31 - __builtin_clz(g.Hs);
    int r_n[BI], r_p[BI], r_q[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int pix = c_begin * KC + brow0 + BROWS * i;
        r_n[i] = pix / (g.Hs * g.Ws);
        const int rem = pix - r_n[i] * (g.Hs * g.Ws);
        r_p[i] = rem / g.Ws;
        r_q[i] = rem - r_p[i] * g.Ws;
    }

    f32x4 ra[AI], rb[BI];
    auto load_a = [&](int i, int c, bool on) {
        const int pix = c * KC + arow0 + AROWS * i;
        const bool ok = on && pix < Kp;
        ra[i] = bload4(rsm, voff(pix * ld_small + a_col, ok));
    };
    auto load_b = [&](int i, int c, bool on) {
        const int pix = c * KC + brow0 + BROWS * i;
        int n, p, q;
        if (POW2) {
            q = pix & (g.Ws - 1);
            p = (pix >> lgW) & (g.Hs - 1);
            n = pix >> (lgW + lgH);
        } else {
            n = r_n[i];
            p = r_p[i];
            q = r_q[i];
            int nq = q + KC, np = p, nn = n;
            bool w = nq >= g.Ws;
            nq = w ? nq - g.Ws : nq;
            np = w ? np + 1 : np;
            w = nq >= g.Ws;
            nq = w ? nq - g.Ws : nq;
            np = w ? np + 1 : np;
            w = np >= g.Hs;
            np = w ? np - g.Hs : np;
            nn = w ? nn + 1 : nn;
            r_q[i] = nq;
            r_p[i] = np;
            r_n[i] = nn;
        }
        const int h = g.s * p - 1 + kh, w = g.s * q - 1 + kw;
        const bool ok = on && pix < Kp && (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
        rb[i] = bload4(rbig, voff(((n * g.Hb + h) * g.Wb + w) * ld_big + b_col, ok));
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(arow0 + AROWS * i) * LDA + aq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(brow0 + BROWS * i) * LDB + bq * 4]) = rb[i];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(i, c_begin, true);
#pragma unroll
        for (int i = 0; i < BI; ++i) load_b(i, c_begin, true);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
#pragma unroll
        for (int kq4 = 0; kq4 < 4; ++kq4) {          // 4 groups of 4 k-pairs
            if (kq4 < AI) load_a(kq4, c + 1, more);
            if (kq4 < BI) load_b(kq4, c + 1, more);
            if (kq4 + 4 < AI) load_a(kq4 + 4, c + 1, more);
            if (kq4 + 4 < BI) load_b(kq4 + 4, c + 1, more);
            PIN_VMEM();
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int kk = kq4 * 4 + k4;
                float af[MR], bf[NR];
#pragma unroll
                for (int i = 0; i < MR; ++i) af[i] = As[(kk * 2 + lh) * LDA + (wm * MR + i) * 32 + lrow];
#pragma unroll
                for (int j = 0; j < NR; ++j) bf[j] = Bs[(kk * 2 + lh) * LDB + (wn * NR + j) * 32 + lrow];
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    float* o = out + (long)blockIdx.z * slab_stride + (long)tap * g.Ca * g.Cb;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int a = m0 + (wm * MR + i) * 32 + row;
                if (a < g.Ca && col < g.Cb) o[(long)a * g.Cb + col] = acc[i][j][r];
            }
        }
}


// ------------------------------------------------------------------------------------------------
// big2small for image-facing layers with 1..3 (5..8: 64-pixel tiles) big-side channels (enc0 / d0 forward, dec6 data gradient):
// the whole contraction K = 16*CB <= 48 (128) fits one LDS tile, so the kernel is one shot -- gather 128 pixels x K and 64 x K weights,
// one barrier, K/2 MFMAs per 32x32 tile, bias + activation, store -- and is bound by the HBM write of its output.
// ------------------------------------------------------------------------------------------------
template <int CB, int MI = 2>      // MI: 32-row MFMA tiles per wave: 128 (CB <= 4) or 64 (CB <= 8, K <= 128) pixels per workgroup
__global__ __launch_bounds__(256) void k_b2s_tapk(const float* __restrict__ big, int ld_big, const float* __restrict__ P,
                                                  float* __restrict__ out, int ld_out, Geom g, const float* __restrict__ bias,
                                                  int act, int vec4, int vec_out) {
    constexpr int K = 16 * CB, LDT = K + 4, TM = 64 * MI;
    __shared__ __attribute__((aligned(16))) float smem[(TM + 64) * LDT];
    float* As = smem;
    float* Bs = smem + TM * LDT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lh = lane >> 5;
    const int M = g.N * g.Hs * g.Ws;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * 64;
    // A: TM pixels x 16 taps, one (pixel, tap) item = CB consecutive floats; a thread owns one pixel and 8 (4) taps
    {
        constexpr int TPT = 16 * TM / 256;
        const int r = tid & (TM - 1), t0 = (tid / TM) * TPT;
        const int m = m0 + r;
        const int mm = min(m, M - 1);
        const int n = mm / (g.Hs * g.Ws);
        const int rem = mm - n * (g.Hs * g.Ws);
        const int p = rem / g.Ws, q = rem - p * g.Ws;
        const int h0 = g.s * p - 1, w0 = g.s * q - 1;
        const float* base = big + ((long)(n * g.Hb + h0) * g.Wb + w0) * ld_big;
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
            const int tap = t0 + tt, kh = tap >> 2, kw = tap & 3;
            const bool ok = m < M && (unsigned)(h0 + kh) < (unsigned)g.Hb && (unsigned)(w0 + kw) < (unsigned)g.Wb;
            const float* src = base + ((long)kh * g.Wb + kw) * ld_big;
            float* dst = &As[r * LDT + tap * CB];
            if (CB == 4 && vec4) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(dst) = ok ? *reinterpret_cast<const f32x4*>(src) : z;
            } else {
#pragma unroll
                for (int c = 0; c < CB; ++c) dst[c] = ok ? src[c] : 0.f;
            }
        }
    }
    // B: 64 output channels x K, Bs[a][tap*CB + c] = P[(tap*Ca + a)*CB + c]
    for (int item = tid; item < 64 * 16; item += 256) {
        const int a = item & 63, tap = item >> 6;
        const bool ok = n0 + a < g.Ca;
#pragma unroll
        for (int c = 0; c < CB; ++c) Bs[a * LDT + tap * CB + c] = ok ? P[((long)tap * g.Ca + n0 + a) * CB + c] : 0.f;
    }
    __syncthreads();
    f32x16 acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < K / 8; ++kk) {
        f32x4 af[MI], bf;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MI + i) * 32 + lrow) * LDT + kk * 8 + lh * 4]);
        bf = *reinterpret_cast<const f32x4*>(&Bs[(wn * 32 + lrow) * LDT + kk * 8 + lh * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[e], af[i][e], acc[i], 0, 0, 0);   // D[channel][pixel]
    }
    // transposed product (weights as the A operand): a lane owns pixel lrow of its tile and, per accumulator quad, 4 consecutive
    // channels -- one 16-byte store instead of four 4-byte ones (the kernel is bound by its output stores)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + (wm * MI + i) * 32 + lrow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = n0 + wn * 32 + 8 * q + 4 * lh;
            if (m >= M || ch >= g.Ca) continue;
            if (vec_out) {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (bias != nullptr) bv = *reinterpret_cast<const f32x4*>(bias + ch);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pg_act_epi(acc[i][4 * q + e] + bv[e], act);
                *reinterpret_cast<f32x4*>(out + (long)m * ld_out + ch) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ch + e < g.Ca) out[(long)m * ld_out + ch + e] = pg_act_epi(acc[i][4 * q + e] + (bias != nullptr ? bias[ch + e] : 0.f), act);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same layers (1..4 big-side channels) as a PERSISTENT, software-pipelined kernel.  The one-shot form above serialises, per
// workgroup, gather -> LDS -> 64 MFMAs per wave -> stores, and the co-resident workgroups of a CU run those phases in lockstep:
// measured on d0 at batch 32 (ablation builds, tools/ab_tapkp.sh) the phase times simply ADD -- 19 us of loads / staging, 28 us of
// MFMAs (= the fp32 MFMA peak for the layer's 4.3 GFLOP) and 17 us of stores give the 67-77 us of the launch: nothing overlaps.
// Here ONE wave keeps all three in flight: a workgroup holds its weight FRAGMENTS in registers for its whole life, walks pixel
// tiles t = blockIdx.x, + gridDim.x, ... (neighbouring workgroups work on neighbouring tiles: the 4-row input windows overlap in
// L2) with two accumulator sets and two LDS buffers, and inside the MFMA sequence of tile t it issues, one group per k-step,
// the bias / activation / 16-byte stores of tile t - 1 (the other accumulator set), while the gather loads of tile t + 1 are in
// flight; their LDS stores go to the other buffer after the last MFMA has been issued.  One barrier per tile.
// ------------------------------------------------------------------------------------------------
// Branch-free: gathers are raw buffer loads whose offset is forced out of range for padding / ragged rows (the hardware returns 0),
// stores are raw buffer stores dropped the same way, the activation is a template parameter -- the whole tile step is ONE basic block,
// so hipcc keeps the interleave below (an earlier form with `ok ? *src : 0` loads compiled to a branch and an `s_waitcnt vmcnt(0)` per
// load: the phases ran strictly one after the other again).
// ACT: 0 none, 1 LeakyReLU(0.2) (other activations: the one-shot kernel).  WIDE: one 16-byte load per tap (CB == 4, or CB == 3 inside
// pixels of stride % 4 == 0: the 4th float is dropped), else CB scalar loads.
// STATS (K5: InstanceNorm statistics from the conv epilogue, unet.py:19-20; ACT == 0, no bias): next to the stores of a finished tile each
// wave also emits the sums / sums of squares of ITS 64 pixels x 32 channels -- a lane adds its two row tiles (fp32: one rounding per
// pair), the 64 lanes exchange them through a wave-private LDS patch, lane (slot, half) sums 32 pixels in fp64 and writes
// part[((n * chunks + chunk) * Ca + c) * 2 + {0, 1}], chunk = 2 * (tile within the sample) + wm.  Tiles never straddle samples
// (Hs * Ws % 128 == 0, checked by the host); fixed order, deterministic.
template <int CB, int ACT, bool WIDE, bool STATS = false>
__global__ __launch_bounds__(256, STATS ? 1 : 2) void k_b2s_tapkp(const float* __restrict__ big, int ld_big, const float* __restrict__ P,
                                                      float* __restrict__ out, int ld_out, Geom g, const float* __restrict__ bias,
                                                      int big_bytes, int out_bytes, int ntiles, double* __restrict__ part = nullptr) {
    constexpr int K = 16 * CB, LDT = K + 4, TM = 128, MI = 2, TPT = 8, NKK = K / 8, PC = WIDE ? 4 : CB;
    constexpr int SROW = 68;       // stats patch: 32 value slots x (64 lanes + 4 pad) floats per wave
    __shared__ __attribute__((aligned(16))) float smem[2 * TM * LDT + (STATS ? 4 * 32 * SROW : 0)];
    const __amdgpu_buffer_rsrc_t rbig = __builtin_amdgcn_make_buffer_rsrc((void*)big, 0, big_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lh = lane >> 5;
    const int M = g.N * g.Hs * g.Ws, HWs = g.Hs * g.Ws;
    const int n0 = blockIdx.y * 64;
    // weight fragments of this wave's 32 output channels, for the whole kernel: bfr[kk][e] = W[a][k = 8 kk + 4 lh + e], k = tap * CB + c
    f32x4 bfr[NKK];
    {
        const int a = min(n0 + wn * 32 + lrow, g.Ca - 1);        // (rows beyond Ca are computed and never stored)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            if (CB == 4) {                 // k = 4 tap + c with tap = 2 kk + lh: the tap's four channels are 16 contiguous, aligned bytes
                bfr[kk] = *reinterpret_cast<const f32x4*>(P + ((long)(2 * kk + lh) * g.Ca + a) * 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = kk * 8 + lh * 4 + e, tap = k / CB, c = k - tap * CB;
                    bfr[kk][e] = P[((long)tap * g.Ca + a) * CB + c];
                }
            }
        }
    }
    // bias of this lane's four channel quads (channels n0 + 32 wn + 8 q + 4 lh .. + 3; Ca % 4 == 0)
    f32x4 bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ch = min(n0 + wn * 32 + 8 * q + 4 * lh, g.Ca - 4);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        bv[q] = bias != nullptr ? *reinterpret_cast<const f32x4*>(bias + ch) : z;
    }
    // gather role of this thread: pixel r of the tile, taps t0 .. t0 + 7 (16 * 128 items / 256 threads)
    const int r = tid & (TM - 1), t0 = (tid / TM) * TPT;
    float pre[TPT][PC];
    auto gather = [&](int tile) {
        const int m = tile * TM + r;
        const int mm = min(m, M - 1);
        const int n = mm / HWs;
        const int rem = mm - n * HWs;
        const int p = rem / g.Ws, q = rem - p * g.Ws;
        const int h0 = g.s * p - 1, w0 = g.s * q - 1;
        const int base = ((n * g.Hb + h0) * g.Wb + w0) * ld_big;       // element offset of the window's first pixel (may be negative: masked)
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
            const int tap = t0 + tt, kh = tap >> 2, kw = tap & 3;
            const bool ok = (m < M) & ((unsigned)(h0 + kh) < (unsigned)g.Hb) & ((unsigned)(w0 + kw) < (unsigned)g.Wb);
            const int eo = base + (kh * g.Wb + kw) * ld_big;
            if (WIDE) {
                const f32x4 v = bload4(rbig, voff(eo, ok));
#pragma unroll
                for (int c = 0; c < 4; ++c) pre[tt][c] = v[c];
            } else {
#pragma unroll
                for (int c = 0; c < CB; ++c) pre[tt][c] = bload1(rbig, voff(eo + c, ok));
            }
        }
    };
    auto stage = [&](float* As) {
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
            float* dst = &As[r * LDT + (t0 + tt) * CB];
            if (CB == 4) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = pre[tt][c];
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int c = 0; c < CB; ++c) dst[c] = pre[tt][c];
            }
        }
    };
    // store group gi = (i, q) of a finished tile: pixel lrow of row tile i, channels 8 q + 4 lh .. + 3 of this wave's 32
    auto store_group = [&](const f32x16 (&ac)[MI], int tile_done, int gi) {
        const int i = gi >> 2, q = gi & 3;
        const int m = tile_done * TM + (wm * MI + i) * 32 + lrow;
        const int ch = n0 + wn * 32 + 8 * q + 4 * lh;
        const bool ok = (m < M) & (ch < g.Ca);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = ac[i][4 * q + e] + bv[q][e];
            v[e] = ACT == 1 ? (x > 0.f ? x : 0.2f * x) : x;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, voff(m * ld_out + ch, ok), 0, 0);
    };
    // STATS: the two halves of the exchange (wave-private LDS patch: program order + lgkmcnt order them, no barrier)
    float* spatch = smem + 2 * TM * LDT + wave * 32 * SROW;
    auto stats_put = [&](const f32x16 (&ac)[MI]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            spatch[j * SROW + lane] = ac[0][j] + ac[1][j];
            spatch[(16 + j) * SROW + lane] = __builtin_fmaf(ac[0][j], ac[0][j], ac[1][j] * ac[1][j]);
        }
    };
    auto stats_get = [&](int tile_done) {
        // this lane: slot v = lane & 31 (0..15: sum of accumulator register v, 16..31: its squares), half h = lane >> 5 of the writers
        const int v = lane & 31, h = lane >> 5;
        double sum = 0.0;
#pragma unroll
        for (int p4 = 0; p4 < 8; ++p4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&spatch[v * SROW + h * 32 + p4 * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += (double)t[e];
        }
        const int rr = v & 15;
        const int ch = n0 + wn * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
        const int tps = HWs / TM;                              // tiles per sample
        const int n = tile_done / tps, chunk = 2 * (tile_done - n * tps) + wm;
        if (tile_done < ntiles && ch < g.Ca) part[(((long)n * (2 * tps) + chunk) * g.Ca + ch) * 2 + (v >> 4)] = sum;
    };
    // one tile: MFMAs of `tile` from As into `cur`, the stores of `prev_tile` (accumulators `prev`) woven between the k-steps, then
    // the next tile's staged registers into the other LDS buffer
    auto step = [&](f32x16 (&cur)[MI], const f32x16 (&prev)[MI], int prev_tile, const float* As, float* As_next, int next2) {
        if (STATS) stats_put(prev);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) cur[i][rr] = 0.f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            f32x4 af[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MI + i) * 32 + lrow) * LDT + kk * 8 + lh * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i) cur[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[kk][e], af[i][e], cur[i], 0, 0, 0);   // D[channel][pixel]
#pragma unroll
            for (int gi = (kk * 8) / NKK; gi < ((kk + 1) * 8) / NKK; ++gi) store_group(prev, prev_tile, gi);
            if (STATS && kk == NKK / 2) stats_get(prev_tile);
            __builtin_amdgcn_sched_barrier(0);         // keep the stores of the previous tile BETWEEN the MFMA groups
        }
        stage(As_next);            // (registers of the tile after this one; zeros beyond the last tile)
        __syncthreads();           // As_next is complete; every wave is done reading As (it is overwritten one step later)
        gather(next2);             // the tile two steps ahead (>= ntiles: every load out of range, no traffic)
    };
    float* A0 = smem;
    float* A1 = smem + TM * LDT;
    f32x16 accA[MI], accB[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) accB[i][rr] = 0.f;
    const int G = gridDim.x, big_tile = 0x3fffff;      // big_tile * TM >= M: every row masked
    int tile = blockIdx.x;
    gather(tile);
    stage(A0);
    __syncthreads();
    gather(tile + G < ntiles ? tile + G : big_tile);
    int prev_tile = big_tile;      // nothing to store in the first step
    for (;;) {
        step(accA, accB, prev_tile, A0, A1, tile + 2 * G < ntiles ? tile + 2 * G : big_tile);
        prev_tile = tile;
        tile += G;
        if (tile >= ntiles) {
#pragma unroll
            for (int gi = 0; gi < 8; ++gi) store_group(accA, prev_tile, gi);
            if (STATS) {
                stats_put(accA);
                stats_get(prev_tile);
            }
            break;
        }
        step(accB, accA, prev_tile, A1, A0, tile + 2 * G < ntiles ? tile + 2 * G : big_tile);
        prev_tile = tile;
        tile += G;
        if (tile >= ntiles) {
#pragma unroll
            for (int gi = 0; gi < 8; ++gi) store_group(accB, prev_tile, gi);
            if (STATS) {
                stats_put(accB);
                stats_get(prev_tile);
            }
            break;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the same image-facing layers (enc0, d0: Cb = 3 / 4; dec6: Cb = output_nc) with the 16 taps folded into N, as a
// PERSISTENT kernel: dP[(tap, b)][a] = sum_pix small[pix][a] * big[window(pix), tap][b].  The taps-in-N kernel k_wgrad_tapn<.., 1> cuts
// the pixel dimension into ~1000 slices of 8 chunks (one 16-KB slab each, a 16-MB reduce) and every workgroup pays a prologue and an
// epilogue for 256 pixels; it ran at 45-74 us on layers whose MFMA and HBM times are ~15 us each.  Here <= 512 workgroups walk
// 128-pixel tiles t = blockIdx.x, + gridDim.x, ..., accumulate their BM x BN block of dP over all of them in ONE accumulator set,
// and write one slab at the end (fixed-order reduce over the workgroups as before: deterministic).  Per tile: the small-side tile
// [128 pixels][BM channels] (contiguous rows) and the gathered window tile [128 pixels][16 CB] come through registers (branch-free
// buffer loads, issued for tile t + 1 before the MFMAs of tile t) into LDS; both MFMA operands are read with ds_read_b32 (pixel =
// the MFMA's k index).  WM x WN waves: 2 x 2 (64 channels x 64 columns) for CB >= 3, 4 x 1 (128 channels x 32 columns) for CB <= 2.
// ------------------------------------------------------------------------------------------------
template <int CB, bool WIDE, int WM>
__global__ __launch_bounds__(256, 2) void k_wgrad_tapnp(const float* __restrict__ small, int ld_small, const float* __restrict__ big,
                                                        int ld_big, float* __restrict__ slabs, long slab_stride, Geom g,
                                                        int small_bytes, int big_bytes, int ntiles) {
    constexpr int WN = 4 / WM, BM = 32 * WM, K = 16 * CB, TM = 128, TPT = 8, PC = WIDE ? 4 : CB;
    constexpr int LDA = BM + 4, LDB = 32 * WN + 4;          // (columns >= K of the window tile are zero)
    constexpr int AQ = BM / 4, ALD = TM * AQ / 256;         // float4 loads of the small-side tile per thread
    __shared__ __attribute__((aligned(16))) float smem[TM * LDA + TM * LDB];
    float* As = smem;
    float* Bs = smem + TM * LDA;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)small, 0, small_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)big, 0, big_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int M = g.N * g.Hs * g.Ws, HWs = g.Hs * g.Ws;
    const int a0 = blockIdx.y * BM;
    // the window tile's pad columns are written once
    for (int i = tid; i < TM * LDB; i += 256) Bs[i] = 0.f;
    __syncthreads();
    const int r = tid & (TM - 1), t0 = (tid / TM) * TPT;    // gather role: pixel r of the tile, taps t0 .. t0 + 7
    f32x4 pa[ALD];
    float pb[TPT][PC];
    auto fetch = [&](int tile) {
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int idx = tid + 256 * i, pix = idx / AQ, q4 = idx - pix * AQ;
            const int m = tile * TM + pix, a = a0 + q4 * 4;
            pa[i] = bload4(rs, voff(m * ld_small + a, (m < M) & (a < g.Ca)));
        }
        const int m = tile * TM + r;
        const int mm = min(m, M - 1);
        const int n = mm / HWs;
        const int rem = mm - n * HWs;
        const int p = rem / g.Ws, q = rem - p * g.Ws;
        const int h0 = g.s * p - 1, w0 = g.s * q - 1;
        const int base = ((n * g.Hb + h0) * g.Wb + w0) * ld_big;
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
            const int tap = t0 + tt, kh = tap >> 2, kw = tap & 3;
            const bool ok = (m < M) & ((unsigned)(h0 + kh) < (unsigned)g.Hb) & ((unsigned)(w0 + kw) < (unsigned)g.Wb);
            const int eo = base + (kh * g.Wb + kw) * ld_big;
            if (WIDE) {
                const f32x4 v = bload4(rb, voff(eo, ok));
#pragma unroll
                for (int c = 0; c < 4; ++c) pb[tt][c] = v[c];
            } else {
#pragma unroll
                for (int c = 0; c < CB; ++c) pb[tt][c] = bload1(rb, voff(eo + c, ok));
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int idx = tid + 256 * i, pix = idx / AQ, q4 = idx - pix * AQ;
            *reinterpret_cast<f32x4*>(&As[pix * LDA + q4 * 4]) = pa[i];
        }
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
            float* dst = &Bs[r * LDB + (t0 + tt) * CB];
            if (CB == 4) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = pb[tt][c];
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int c = 0; c < CB; ++c) dst[c] = pb[tt][c];
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[rr] = 0.f;
    const int G = gridDim.x, big_tile = 0x3fffff;
    int tile = blockIdx.x;
    fetch(tile);
    for (; tile < ntiles; tile += G) {
        stage();
        __syncthreads();
        fetch(tile + G < ntiles ? tile + G : big_tile);     // in flight under the MFMAs below (beyond the last tile: every load masked)
        __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll 16
        for (int s2 = 0; s2 < TM / 2; ++s2) {
            const float af = As[(2 * s2 + lh) * LDA + wm * 32 + lrow];
            const float bf = Bs[(2 * s2 + lh) * LDB + wn * 32 + lrow];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);          // D[channel a][column (tap, b)]
        }
        __syncthreads();
    }
    // one slab per workgroup column: slab[(tap * Ca + a) * Cb + b]
    float* o = slabs + (long)blockIdx.x * slab_stride;
    const int col = wn * 32 + lrow;
    if (col < K) {
        const int tap = col / CB, b = col - tap * CB;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int a = a0 + wm * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            if (a < g.Ca) o[((long)tap * g.Ca + a) * CB + b] = acc[rr];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose2d(4, 2, 1) onto <= 4 channels (the generator's head dec6, the discriminator's first-layer data gradient) in ONE pass,
// taps folded into N:
//     D[p][(b, tap)] = small[p][:] . P[tap][:][b]                      (MFMA 16x16x4: rows = small pixels, one 16-column tile = the 16
//                                                                       taps of output channel b)
//     big[n, h, w, b] = act(bias[b] + the four D entries that reach (h, w))
// The two-launch form (row GEMM k_b2s_fast<.., true> + k_col2im_s2_lds) writes D to HBM and reads it back: 16.8 / 67 MB each way at
// batch 16, 57 / 70 us for layers whose own traffic is 138 / 84 MB (22 / 13 us).  Here a workgroup owns a TF_H x TF_W block of small
// pixels of one sample, keeps its D block in LDS, and writes the (2 TF_H - 2) x (2 TF_W - 2) output pixels all of whose four taps lie
// inside the block; blocks step by TF_H - 1 / TF_W - 1 small pixels (one row / column of every block is recomputed by its
// neighbour: 18 % more MFMA work on a memory-bound kernel, re-read from L2), and start at small pixel -1 (zeros) so the image border
// needs no special case.  PERSISTENT: <= 512 workgroups walk blocks blockIdx.x, + gridDim.x, ...; the lanes' operand rows come
// straight from global memory in MFMA fragment order (lane (pixel r, k group q) loads 16 bytes = 4 consecutive channels, which feed
// four MFMAs; the matching weight fragments are read from an LDS image of P with the same k order), no LDS staging of `small`, and
// block t + 1's loads are issued before block t's MFMAs (two register sets).
// ------------------------------------------------------------------------------------------------
constexpr int TF_H = 8, TF_W = 16;
typedef __bf16 tf_bf16x8 __attribute__((ext_vector_type(8)));
// BF: `small` is a bf16 tensor (PG_ALGO_BF16 with bf16 activation storage): a lane's 16 bytes are 8 consecutive channels = its operand of
// one v_mfma_f32_16x16x32_bf16, the weights are rounded to bf16 as they are laid into LDS ([(b, tap)][k], k contiguous), fp32 accumulation
// and an fp32 result as everywhere in that mode.  KK = operand loads per pixel: Ca / 16 (fp32) or Ca / 32 (bf16).
template <int CB, int KK, bool BF>
__global__ __launch_bounds__(256, 2) void k_s2b_tapnf(const void* __restrict__ small_v, int ld_small, const float* __restrict__ P,
                                                      const float* __restrict__ bias, float* __restrict__ big, int ld_big, Geom g, int act,
                                                      int small_bytes, int big_bytes, int nbh, int nbw, int nblocks, int vec4, int cb_total,
                                                      int b0) {
    // (5 .. 8 output channels run as two launches: channels b0 .. b0 + CB - 1 of cb_total each)
    constexpr int Ca = (BF ? 32 : 16) * KK;
    constexpr int SK = 16 * CB + 4;      // fp32: k-row pitch of the weight image: the four k groups of a fragment read land in four bank quarters
    constexpr int SKB = Ca + 8;          // bf16: (b, tap)-row pitch in elements (16 bytes of padding: conflict-free ds_read_b128 per 16 lanes)
    // D block in LDS: [pixel][tap][channel], so a lane's CB channels of one (pixel, tap) are one ds_write / one ds_read (CB = 4: 16 bytes)
    constexpr int SD = (CB % 4 == 0) ? 16 * CB + 4 : (CB % 2 == 0) ? 16 * CB + 2 : 16 * CB + 1;
    constexpr int OH = 2 * TF_H - 2, OW = 2 * TF_W - 2;
    constexpr int BS_BYTES = BF ? 16 * CB * SKB * 2 : Ca * SK * 4;
    __shared__ __attribute__((aligned(16))) char Bs_raw[BS_BYTES];
    __shared__ __attribute__((aligned(16))) float Ds[TF_H * TF_W * SD];
    float* const Bs = reinterpret_cast<float*>(Bs_raw);
    __bf16* const Bh = reinterpret_cast<__bf16*>(Bs_raw);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)small_v, 0, small_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)big, 0, big_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    for (int i = tid; i < Ca * CB * 16; i += 256) {
        if constexpr (BF) {                                  // Bh[(b, tap)][k] = bf16(P[tap][k][b])
            const int k = i % Ca, bt = i / Ca, tap = bt & 15, b = bt >> 4;
            Bh[bt * SKB + k] = (__bf16)P[((long)tap * Ca + k) * cb_total + b0 + b];
        } else {                                             // Bs[k][b][tap] = P[tap][k][b]
            const int tap = i & 15, b = (i >> 4) % CB, k = i / (16 * CB);
            Bs[k * SK + b * 16 + tap] = P[((long)tap * Ca + k) * cb_total + b0 + b];
        }
    }
    float bv[CB];
#pragma unroll
    for (int b = 0; b < CB; ++b) bv[b] = bias ? bias[b0 + b] : 0.f;
    __syncthreads();

    const int G = gridDim.x;
    int blk = xcd_remap(blockIdx.x, G);
    // this lane's two pixel rows of block `b`: rows 2 * wave + {0, 1} of the block, column c
    auto issue = [&](f32x4 (&a)[2][KK], int b) {
        const bool on = b < nblocks;
        const int bb = on ? b : 0;
        const int bj = bb % nbw, t = bb / nbw;
        const int bi = t % nbh, n = t / nbh;
        const int j = bj * (TF_W - 1) - 1 + c;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int i = bi * (TF_H - 1) - 1 + 2 * wave + rt;
            const bool ok = on && (unsigned)i < (unsigned)g.Hs && (unsigned)j < (unsigned)g.Ws;
            const int off = ((n * g.Hs + i) * g.Ws + j) * ld_small + q * (BF ? 8 : 4);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if constexpr (BF)
                    a[rt][kk] = bload4(rS, (int)(((unsigned)(off + kk * 32) << 1) | (ok ? 0u : 0x80000000u)));
                else
                    a[rt][kk] = bload4(rS, voff(off + kk * 16, ok));
            }
        }
    };
    auto body = [&](f32x4 (&cur)[2][KK], f32x4 (&nxt)[2][KK], int b) {
        issue(nxt, b + G);
        PIN_VMEM();
        f32x4 acc[2][CB];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[rt][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                tf_bf16x8 bf[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) bf[cb] = *reinterpret_cast<const tf_bf16x8*>(Bh + (cb * 16 + c) * SKB + kk * 32 + q * 8);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[rt][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tf_bf16x8, cur[rt][kk]), bf[cb], acc[rt][cb], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float bf[CB];
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) bf[cb] = Bs[(kk * 16 + q * 4 + e) * SK + cb * 16 + c];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            acc[rt][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[rt][kk][e], bf[cb], acc[rt][cb], 0, 0, 0);
                }
        }
        // D block -> LDS: lane (tap c, group q) holds pixels 4 q + {0..3} of its two block rows
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) Ds[((2 * wave + rt) * TF_W + q * 4 + e) * SD + c * CB + cb] = acc[rt][cb][e];
        __syncthreads();
        const int bj = b % nbw, t = b / nbw;
        const int bi = t % nbh, n = t / nbh;
        const int h0 = 2 * (bi * (TF_H - 1) - 1) + 1, w0 = 2 * (bj * (TF_W - 1) - 1) + 1;
        for (int o = tid; o < OH * OW; o += 256) {
            const int oh = o / OW, ow = o - oh * OW;
            const int h = h0 + oh, w = w0 + ow;
            // output row h0 + oh (h0 odd): oh even -> rows oh/2 (kh 2) and oh/2 + 1 (kh 0); oh odd -> rows (oh+1)/2 (kh 1) and (oh-1)/2 (kh 3)
            const int la = (oh + 1) >> 1, ka = (oh & 1) ? 1 : 2, lb = (oh & 1) ? la - 1 : la + 1, kb = (oh & 1) ? 3 : 0;
            const int ma = (ow + 1) >> 1, ua = (ow & 1) ? 1 : 2, mb = (ow & 1) ? ma - 1 : ma + 1, ub = (ow & 1) ? 3 : 0;
            const float* d00 = Ds + (la * TF_W + ma) * SD + (ka * 4 + ua) * CB;
            const float* d01 = Ds + (la * TF_W + mb) * SD + (ka * 4 + ub) * CB;
            const float* d10 = Ds + (lb * TF_W + ma) * SD + (kb * 4 + ua) * CB;
            const float* d11 = Ds + (lb * TF_W + mb) * SD + (kb * 4 + ub) * CB;
            const bool ok = (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
            const int eo = ((n * g.Hb + h) * g.Wb + w) * ld_big + b0;
            float v[CB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) v[cb] = pg_act_epi(bv[cb] + ((d00[cb] + d01[cb]) + (d10[cb] + d11[cb])), act);
            if (CB % 4 == 0 && vec4) {
#pragma unroll
                for (int c4 = 0; c4 < CB / 4; ++c4) {
                    const f32x4 o4 = {v[(4 * c4) % CB], v[(4 * c4 + 1) % CB], v[(4 * c4 + 2) % CB], v[(4 * c4 + 3) % CB]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), rO, voff(eo + 4 * c4, ok), 0, 0);
                }
            } else {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[cb]), rO, voff(eo + cb, ok), 0, 0);
            }
        }
        __syncthreads();
    };
    f32x4 A0[2][KK], A1[2][KK];
    issue(A0, blk);
    while (blk < nblocks) {
        body(A0, A1, blk);
        blk += G;
        if (blk >= nblocks) break;
        body(A1, A0, blk);
        blk += G;
    }
}

// ================================================================================================
// bf16 variants (PG_ALGO_BF16; the "next" row f2, BASELINE config 4): tensors stay fp32 in HBM and in the C ABI -- fp32
// master weights, fp32 InstanceNorm statistics, fp32 accumulation -- but operand tiles are rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) as they are staged into LDS and multiplied on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate),
// which moves these kernels from MFMA-bound to load-bound.  Same tiling, gathers and split-K as the fast fp32 kernels.
// LDS rows are 32 bf16 + 8 pad = 80 B: conflict-free ds_read_b128 of a lane's 8 consecutive k.
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LDKH = KC + 8;
__device__ __forceinline__ bf16x4 to_bf16(f32x4 v) {
    bf16x4 r;
    r[0] = (__bf16)v[0];
    r[1] = (__bf16)v[1];
    r[2] = (__bf16)v[2];
    r[3] = (__bf16)v[3];
    return r;
}

// bf16 activation storage: 8-byte loads of 4 bf16 (byte offset = element offset * 2), stored to LDS as they are
__device__ __forceinline__ bf16x4 bload2h(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}
__device__ __forceinline__ int voffh(int elem_off, bool ok) {
    return (int)(((unsigned)elem_off << 1) | (ok ? 0u : 0x80000000u));
}
// epilogue store of one value into an fp32 or a bf16 tensor (element index idx)
__device__ __forceinline__ void store_io(float* base, long idx, float v, int out_bf) {
    if (out_bf)
        reinterpret_cast<__bf16*>(base)[idx] = (__bf16)v;
    else
        base[idx] = v;
}

// HIN: the activation operand(s) are stored as bf16 in HBM (bf16 storage mode) instead of fp32 rounded in the kernel
template <int MR, int NR, int WM, int WN, bool ONE, bool HIN = false>
__global__ __launch_bounds__(256) void k_b2s_bf16(const float* __restrict__ big, int ld_big,
                                                  const float* __restrict__ P, float* __restrict__ out, int ld_out,
                                                  long slab_stride, Geom g, int chunks_per_slice,
                                                  const float* __restrict__ bias, int act, int big_bytes, int p_bytes, int out_bf) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int AI = BM / 32, BI = BN / 32;
    __shared__ __attribute__((aligned(16))) __bf16 smem[(BM + BN) * LDKH];
    __bf16* As = smem;
    __bf16* Bs = smem + BM * LDKH;
    const __amdgpu_buffer_rsrc_t rbig = __builtin_amdgcn_make_buffer_rsrc((void*)big, 0, big_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)P, 0, p_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int M = ONE ? g.N * g.Hb * g.Wb : g.N * g.Hs * g.Ws;
    const int K = ONE ? g.Cb : 16 * g.Cb;               // K % 32 == 0 (Cb % 4 == 0; ONE requires Cb % 32 == 0)
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
    const int nchunks = K / KC;
    const int c_begin = blockIdx.z * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    const int kq = tid & 7, r0 = tid >> 3;
    int a_off[AI], a_mask[AI], b_off[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + 32 * i;
        const int mm = min(m, M - 1);
        if (ONE) {
            a_off[i] = mm * ld_big;
            a_mask[i] = (m < M) ? 1 : 0;
            continue;
        }
        const int n = mm / (g.Hs * g.Ws);
        const int rem = mm - n * (g.Hs * g.Ws);
        const int p = rem / g.Ws, q = rem - p * g.Ws;
        const int h0 = g.s * p - 1, w0 = g.s * q - 1;
        a_off[i] = ((n * g.Hb + h0) * g.Wb + w0) * ld_big;
        int wv = 0, mask = 0;          // tap (kh, kw) is inside the image iff row kh and column kw both are
#pragma unroll
        for (int t = 0; t < 4; ++t) wv |= ((unsigned)(w0 + t) < (unsigned)g.Wb) ? (1 << t) : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) mask |= ((unsigned)(h0 + t) < (unsigned)g.Hb) ? (wv << (4 * t)) : 0;
        a_mask[i] = (m < M) ? mask : 0;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int a = n0 + r0 + 32 * i;
        b_off[i] = (a < g.Ca) ? a * g.Cb : 0x10000000;
    }
    const int CaCb = g.Ca * g.Cb;
    int cur_tap = (c_begin * KC + kq * 4) / g.Cb;
    int cur_b = (c_begin * KC + kq * 4) - cur_tap * g.Cb;

    f32x4 ra[AI], rb[BI];
    bf16x4 rah[AI];
    int tapoff = 0, pboff = 0, ctap = 0;
    auto next_tap = [&]() {   // (tap, b) of the next chunk for this thread's float4; Cb >= KC: at most one wrap
        const int tap = cur_tap, b = cur_b;
        cur_b += KC;
        const bool wrap = cur_b >= g.Cb;
        cur_b = wrap ? cur_b - g.Cb : cur_b;
        cur_tap = wrap ? cur_tap + 1 : cur_tap;
        ctap = tap;
        tapoff = ((tap >> 2) * g.Wb + (tap & 3)) * ld_big + b;
        pboff = tap * CaCb + b;
    };
    auto load_a = [&](int i, bool on) {
        const bool ok = on && ((a_mask[i] >> ctap) & 1);
        if constexpr (HIN)
            rah[i] = bload2h(rbig, voffh(a_off[i] + tapoff, ok));
        else
            ra[i] = bload4(rbig, voff(a_off[i] + tapoff, ok));
    };
    auto load_b = [&](int i, bool on) { rb[i] = bload4(rP, voff(b_off[i] + pboff, on)); };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<bf16x4*>(&As[(r0 + 32 * i) * LDKH + kq * 4]) = HIN ? rah[i] : to_bf16(ra[i]);
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<bf16x4*>(&Bs[(r0 + 32 * i) * LDKH + kq * 4]) = to_bf16(rb[i]);
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        next_tap();
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(i, true);
#pragma unroll
        for (int i = 0; i < BI; ++i) load_b(i, true);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        next_tap();
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(i, more);
#pragma unroll
        for (int i = 0; i < BI; ++i) load_b(i, more);
        PIN_VMEM();
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            bf16x8 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&As[((wm * MR + i) * 32 + lrow) * LDKH + ks * 16 + lh * 8]);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[((wn * NR + j) * 32 + lrow) * LDKH + ks * 16 + lh * 8]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    float* o = out + (long)blockIdx.z * slab_stride;
    const bool fin = (slab_stride == 0);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
            const float bv = (fin && bias != nullptr && col < g.Ca) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + (wm * MR + i) * 32 + row;
                if (m < M && col < g.Ca) {
                    float v = acc[i][j][r];
                    if (fin) v = pg_act_epi(v + bv, act);
                    store_io(o, (long)m * ld_out + col, v, out_bf);
                }
            }
        }
}

template <int MR, int NR, int WM, int WN, bool HIN = false>
__global__ __launch_bounds__(256) void k_s2b_bf16(const float* __restrict__ small, int ld_small,
                                                  const float* __restrict__ P, float* __restrict__ out, int ld_out,
                                                  long slab_stride, Geom g, int chunks_per_slice,
                                                  const float* __restrict__ bias, int act, int small_bytes,
                                                  int p_bytes, int out_bf) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int AI = BM / 32;
    constexpr int BG = 256 / BN;        // thread groups along k
    constexpr int NQ = 8 / BG;          // k-quads per thread per chunk
    __shared__ __attribute__((aligned(16))) __bf16 smem[(BM + BN) * LDKH];
    __bf16* As = smem;
    __bf16* Bs = smem + BM * LDKH;
    const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)small, 0, small_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)P, 0, p_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;

    const int ncls = (g.s == 2) ? 4 : 1;
    const int cls = blockIdx.z % ncls, slice = blockIdx.z / ncls;
    const int ah = (g.s == 2) ? (cls >> 1) : 0, aw = (g.s == 2) ? (cls & 1) : 0;
    const int T = (g.s == 2) ? 2 : 4, Tsh = (g.s == 2) ? 1 : 2;
    const int Hc = (g.s == 2) ? (g.Hb - ah + 1) / 2 : g.Hb;
    const int Wc = (g.s == 2) ? (g.Wb - aw + 1) / 2 : g.Wb;
    const int kh0 = (g.s == 2) ? (1 - ah) : 0, kw0 = (g.s == 2) ? (1 - aw) : 0;
    const int Mc = g.N * Hc * Wc, K = T * T * g.Ca;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
    if (m0 >= Mc) return;
    const int nchunks = (K + KC - 1) / KC;
    const int c_begin = slice * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    const int kq = tid & 7, r0 = tid >> 3;
    int a_off[AI], a_mask[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + 32 * i;
        const int mm = min(m, Mc - 1);
        const int n = mm / (Hc * Wc);
        const int rem = mm - n * (Hc * Wc);
        const int ii = rem / Wc, jj = rem - ii * Wc;
        const int ib = (g.s == 2) ? ii + ah : ii + 1, jb = (g.s == 2) ? jj + aw : jj + 1;
        a_off[i] = ((n * g.Hs + ib) * g.Ws + jb) * ld_small;
        int wv = 0, mask = 0;          // local tap (th, tw) reads small pixel (ib - th, jb - tw)
#pragma unroll
        for (int t = 0; t < 4; ++t) wv |= (t < T && (unsigned)(jb - t) < (unsigned)g.Ws) ? (1 << t) : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) mask |= (t < T && (unsigned)(ib - t) < (unsigned)g.Hs) ? (wv << (T * t)) : 0;
        a_mask[i] = (m < Mc) ? mask : 0;
    }
    const int bn = tid % BN, bg = tid / BN;
    const int ncol = n0 + bn;
    const int ncol_off = (ncol < g.Cb) ? ncol : 0x10000000;
    const int CaCb = g.Ca * g.Cb;

    // incremental (tloc, a) per k-quad: A float4 (index 0) and the NQ quads of B (1..NQ); Ca >= KC: <= 1 wrap
    int q_tl[NQ + 1], q_a[NQ + 1];
    {
        const int k = c_begin * KC + kq * 4;
        q_tl[0] = k / g.Ca;
        q_a[0] = k - q_tl[0] * g.Ca;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int k = c_begin * KC + 4 * (bg + BG * i);
        q_tl[i + 1] = k / g.Ca;
        q_a[i + 1] = k - q_tl[i + 1] * g.Ca;
    }
    auto quad = [&](int qi, int& tl, int& a) {   // returns (tloc, a) of quad qi for the chunk being loaded
        tl = q_tl[qi];
        a = q_a[qi];
        const int na = a + KC;
        const bool wrap = na >= g.Ca;
        q_a[qi] = wrap ? na - g.Ca : na;
        q_tl[qi] = wrap ? tl + 1 : tl;
    };

    f32x4 ra[AI], rb[NQ];
    bf16x4 rah[AI];
    int a_tl = 0, a_koff = OOB;
    auto next_a = [&](int c) {
        const int k = c * KC + kq * 4;
        int tl, a;
        quad(0, tl, a);
        a_tl = tl;
        const int th = tl >> Tsh, tw = tl & (T - 1);
        a_koff = (k < K) ? (a - (th * g.Ws + tw) * ld_small) : 0x20000000;
    };
    auto load_a = [&](int i, bool on) {
        const bool ok = on && ((a_mask[i] >> a_tl) & 1) && (a_koff < 0x10000000);
        if constexpr (HIN)
            rah[i] = bload2h(rsm, voffh(a_off[i] + a_koff, ok));
        else
            ra[i] = bload4(rsm, voff(a_off[i] + a_koff, ok));
    };
    auto load_b = [&](int i, int c, bool on) {
        const int kb = c * KC + 4 * (bg + BG * i);
        int tl, a;
        quad(i + 1, tl, a);
        if (BN >= 64) {   // a wave shares one k-quad (bg = tid / BN is wave-uniform): keep its decode on the scalar unit
            tl = __builtin_amdgcn_readfirstlane(tl);
            a = __builtin_amdgcn_readfirstlane(a);
        }
        const int th = tl >> Tsh, tw = tl & (T - 1);
        const int tap = (kh0 + g.s * th) * 4 + (kw0 + g.s * tw);
        const bool ok = on && kb < K;
        const int rowoff = tap * CaCb + a * g.Cb;
        const int base = voff(rowoff + ncol_off, ok);
        const int st = g.Cb * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[i][e] = bload1(rP, base + e * st);
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<bf16x4*>(&As[(r0 + 32 * i) * LDKH + kq * 4]) = HIN ? rah[i] : to_bf16(ra[i]);
#pragma unroll
        for (int i = 0; i < NQ; ++i) *reinterpret_cast<bf16x4*>(&Bs[bn * LDKH + 4 * (bg + BG * i)]) = to_bf16(rb[i]);
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        next_a(c_begin);
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(i, true);
#pragma unroll
        for (int i = 0; i < NQ; ++i) load_b(i, c_begin, true);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        next_a(c + 1);
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(i, more);
#pragma unroll
        for (int i = 0; i < NQ; ++i) load_b(i, c + 1, more);
        PIN_VMEM();
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            bf16x8 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&As[((wm * MR + i) * 32 + lrow) * LDKH + ks * 16 + lh * 8]);
#pragma unroll
            for (int j = 0; j < NR; ++j)
                bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[((wn * NR + j) * 32 + lrow) * LDKH + ks * 16 + lh * 8]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    const bool fin = (slab_stride == 0);
    float* o = out + (long)slice * slab_stride;
    const int ldo = fin ? ld_out : g.Cb;
    float bv[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int col = n0 + (wn * NR + j) * 32 + lrow;
        bv[j] = (fin && bias != nullptr && col < g.Cb) ? bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        // decode the lane's first row with two divisions, then walk the other 15 (at most 28 rows further) with a
        // carry chain: 64 divisions per thread in the epilogue cost as much as several K-chunks
        const int mb = m0 + (wm * MR + i) * 32 + 4 * lh;
        const int mbc = min(mb, Mc - 1);
        const int nb0 = mbc / (Hc * Wc);
        const int remb = mbc - nb0 * (Hc * Wc);
        const int ib0 = remb / Wc, jb0 = remb - ib0 * Wc;
        const bool chain = Wc >= 16 && Hc >= 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int m = m0 + (wm * MR + i) * 32 + row;
            if (m < Mc) {
                int n, ii, jj;
                if (chain) {
                    const int d = (r & 3) + 8 * (r >> 2);      // 0..27 rows past the decoded one
                    jj = jb0 + d;
                    ii = ib0;
                    n = nb0;
                    bool w1 = jj >= Wc;
                    jj = w1 ? jj - Wc : jj;
                    ii = w1 ? ii + 1 : ii;
                    w1 = jj >= Wc;
                    jj = w1 ? jj - Wc : jj;
                    ii = w1 ? ii + 1 : ii;
                    w1 = ii >= Hc;
                    ii = w1 ? ii - Hc : ii;
                    n = w1 ? n + 1 : n;
                } else {
                    n = m / (Hc * Wc);
                    const int rem = m - n * (Hc * Wc);
                    ii = rem / Wc;
                    jj = rem - ii * Wc;
                }
                const int h = (g.s == 2) ? 2 * ii + ah : ii, w = (g.s == 2) ? 2 * jj + aw : jj;
                const long orow = (long)((n * g.Hb + h) * g.Wb + w) * ldo;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int col = n0 + (wn * NR + j) * 32 + lrow;
                    if (col < g.Cb) {
                        float v = acc[i][j][r];
                        if (fin) v = pg_act_epi(v + bv[j], act);
                        store_io(o, orow + col, v, out_bf);
                    }
                }
            }
        }
    }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
// 8 consecutive k (rows p .. p + 7*pitch) of this lane's column as an MFMA bf16 fragment: two ds_read_b64_tr_b16
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* p, int pitch) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * pitch));
    // (whole-vector casts: an element-by-element short -> __bf16 bit_cast of the two results is miscompiled by this hipcc
    // into v_perm / v_mov of the first dword only)
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

template <int MR, int NR, int WM, int WN, bool POW2, bool HIN = false>
__global__ __launch_bounds__(256) void k_wgrad_bf16(const float* __restrict__ small, int ld_small,
                                                    const float* __restrict__ big, int ld_big,
                                                    float* __restrict__ out, long slab_stride, Geom g,
                                                    int chunks_per_slice, int tilesB, int small_bytes, int big_bytes) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int LDA = BM + 8, LDB = BN + 8;     // bf16 elements; rows stay 8-byte aligned
    constexpr int AQ = BM / 4, AROWS = 256 / AQ, AI = KC / AROWS;
    constexpr int BQ = BN / 4, BROWS = 256 / BQ, BI = KC / BROWS;
    __shared__ __attribute__((aligned(16))) __bf16 smem[KC * LDA + KC * LDB];
    __bf16* As = smem;
    __bf16* Bs = smem + KC * LDA;
    const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)small, 0, small_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbig = __builtin_amdgcn_make_buffer_rsrc((void*)big, 0, big_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int tile_a = blockIdx.x / tilesB, tile_b = blockIdx.x % tilesB;
    const int m0 = tile_a * BM, n0 = tile_b * BN;
    const int tap = blockIdx.y, kh = tap >> 2, kw = tap & 3;
    const int Kp = g.N * g.Hs * g.Ws;
    const int nchunks = (Kp + KC - 1) / KC;
    const int c_begin = blockIdx.z * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);

    // transposed fragment reads: 16-lane group gr = lane >> 4 covers columns (gr & 1) * 16 .. +15 and k rows (gr >> 1) * 8 .. +7
    const int tr_row = ((lane >> 4) >> 1) * 8 + ((lane & 15) >> 2), tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const int aq = tid % AQ, arow0 = tid / AQ;
    const int bq = tid % BQ, brow0 = tid / BQ;
    const int a_col = (m0 + aq * 4 < g.Ca) ? m0 + aq * 4 : 0x10000000;      // small rows beyond Kp are out of range by
    const int b_col = (n0 + bq * 4 < g.Cb) ? n0 + bq * 4 : 0x10000000;      // construction of small_bytes
    // pixel -> (img, p, q): shifts when Hs, Ws are powers of two (every UNet layer), else a branch-free carry chain
    // (needs Ws >= 16 and Hs >= 2: a step of KC = 32 pixels wraps q at most twice and p at most once)
    const int lgW = 31 - __builtin_clz(g.Ws), lgH = 31 - __builtin_clz(g.Hs);
    int r_n[BI], r_p[BI], r_q[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int pix = c_begin * KC + brow0 + BROWS * i;
        r_n[i] = pix / (g.Hs * g.Ws);
        const int rem = pix - r_n[i] * (g.Hs * g.Ws);
        r_p[i] = rem / g.Ws;
        r_q[i] = rem - r_p[i] * g.Ws;
    }

    f32x4 ra[AI], rb[BI];
    bf16x4 rah[AI], rbh[BI];
    auto load_a = [&](int i, int c, bool on) {
        const int pix = c * KC + arow0 + AROWS * i;
        const bool ok = on && pix < Kp;
        if constexpr (HIN)
            rah[i] = bload2h(rsm, voffh(pix * ld_small + a_col, ok));
        else
            ra[i] = bload4(rsm, voff(pix * ld_small + a_col, ok));
    };
    auto load_b = [&](int i, int c, bool on) {
        const int pix = c * KC + brow0 + BROWS * i;
        int n, p, q;
        if (POW2) {
            q = pix & (g.Ws - 1);
            p = (pix >> lgW) & (g.Hs - 1);
            n = pix >> (lgW + lgH);
        } else {
            n = r_n[i];
            p = r_p[i];
            q = r_q[i];
            int nq = q + KC, np = p, nn = n;
            bool w = nq >= g.Ws;
            nq = w ? nq - g.Ws : nq;
            np = w ? np + 1 : np;
            w = nq >= g.Ws;
            nq = w ? nq - g.Ws : nq;
            np = w ? np + 1 : np;
            w = np >= g.Hs;
            np = w ? np - g.Hs : np;
            nn = w ? nn + 1 : nn;
            r_q[i] = nq;
            r_p[i] = np;
            r_n[i] = nn;
        }
        const int h = g.s * p - 1 + kh, w = g.s * q - 1 + kw;
        const bool ok = on && pix < Kp && (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
        if constexpr (HIN)
            rbh[i] = bload2h(rbig, voffh(((n * g.Hb + h) * g.Wb + w) * ld_big + b_col, ok));
        else
            rb[i] = bload4(rbig, voff(((n * g.Hb + h) * g.Wb + w) * ld_big + b_col, ok));
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<bf16x4*>(&As[(arow0 + AROWS * i) * LDA + aq * 4]) = HIN ? rah[i] : to_bf16(ra[i]);
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<bf16x4*>(&Bs[(brow0 + BROWS * i) * LDB + bq * 4]) = HIN ? rbh[i] : to_bf16(rb[i]);
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(i, c_begin, true);
#pragma unroll
        for (int i = 0; i < BI; ++i) load_b(i, c_begin, true);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
#pragma unroll
        for (int i = 0; i < AI; ++i) load_a(i, c + 1, more);
#pragma unroll
        for (int i = 0; i < BI; ++i) load_b(i, c + 1, more);
        PIN_VMEM();
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            // tiles are [k][m] (K is the pixel axis here and both operands are contiguous along channels in memory, so there is
            // no K-contiguous image to stage): the fragments -- 8 consecutive k of the lane's column -- come from two transposed
            // LDS reads (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block, lane 4q+p addresses row q / columns
            // 4p.., lane i receives column i of the four rows; verified with tools/tr_probe.hip) instead of 8 ds_read_u16
            bf16x8 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i] = tr_frag(&As[(ks * 16 + tr_row) * LDA + (wm * MR + i) * 32 + tr_col], LDA);
#pragma unroll
            for (int j = 0; j < NR; ++j) bf[j] = tr_frag(&Bs[(ks * 16 + tr_row) * LDB + (wn * NR + j) * 32 + tr_col], LDB);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    float* o = out + (long)blockIdx.z * slab_stride + (long)tap * g.Ca * g.Cb;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int a = m0 + (wm * MR + i) * 32 + row;
                if (a < g.Ca && col < g.Cb) o[(long)a * g.Cb + col] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Second halves of the taps-folded-into-N forward paths.  D is the row GEMM's output:
//   small2big:  D[small pixel][(tap, b)]  ->  big[n,h,w,b] = act(sum over the taps that reach (h,w) + bias[b])
//   big2small:  D[big pixel][(tap, a)]    ->  small[n,p,q,a] = act(sum_{kh,kw in range} D[(s*p-1+kh, s*q-1+kw)] + bias[a])
// and the weight re-layout the first needs: W'[(tap*Cb + b)][a] = P[tap][a][b] (a few KB).
// ------------------------------------------------------------------------------------------------
// The same gather for stride 2 with the D rows staged through LDS: a workgroup owns the 2*CT_H x 2*CT_W big pixels of one sample
// that descend from a CT_H x CT_W block of small pixels, loads the (CT_H + 2) x (CT_W + 2) small pixels' D rows that reach them
// once, coalesced (a row is 16 * Cb contiguous floats), and sums the four taps of every output from LDS.  (The kernel above reads
// each D row in 16-byte pieces from four different pixels' threads: 1.3 - 2.3 TB/s at 512 x 512.)
constexpr int CT_H = 4, CT_W = 16;
__global__ __launch_bounds__(256) void k_col2im_s2_lds(const float* __restrict__ D, const float* __restrict__ bias, float* __restrict__ big,
                                                       int ld_big, Geom g, int act, int tiles_h, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) float rows[];   // [(CT_H + 2) * (CT_W + 2)][Nc]
    const int Nc = 16 * g.Cb, nq = Nc >> 2;
    int t = blockIdx.x;
    const int tw = t % tiles_w;
    t /= tiles_w;
    const int th = t % tiles_h, n = t / tiles_h;
    const int a0 = th * CT_H, b0 = tw * CT_W;                     // first small pixel of the block; staged rows start one earlier
    constexpr int RW = CT_W + 2, NR_ = (CT_H + 2) * RW;
    for (int i = threadIdx.x; i < NR_ * nq; i += 256) {
        const int r = i / nq, q = i - r * nq;
        const int ih = a0 - 1 + r / RW, iw = b0 - 1 + r % RW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ih < (unsigned)g.Hs && (unsigned)iw < (unsigned)g.Ws)
            v = *reinterpret_cast<const f32x4*>(D + (long)((n * g.Hs + ih) * g.Ws + iw) * Nc + 4 * q);
        *reinterpret_cast<f32x4*>(rows + r * Nc + 4 * q) = v;
    }
    __syncthreads();
    const int npx = 2 * CT_H * 2 * CT_W;
    for (int i = threadIdx.x; i < npx * g.Cb; i += 256) {
        const int b = i % g.Cb, p = i / g.Cb;
        const int hl = p / (2 * CT_W), wl = p - hl * (2 * CT_W);
        const int h = 2 * a0 + hl, w = 2 * b0 + wl;
        if (h >= g.Hb || w >= g.Wb) continue;
        // big row h = 2a + r: taps kh with (h + 1 - kh) even -> kh = 1 - r + 2j (j = 0, 1), small row a + r - j
        const int rh = hl & 1, rw = wl & 1, al = hl >> 1, bl = wl >> 1;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kh = 1 - rh + 2 * j, ihl = al + rh - j + 1;             // + 1: the staged block starts at a0 - 1
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int kw = 1 - rw + 2 * k, iwl = bl + rw - k + 1;
                acc += rows[(ihl * RW + iwl) * Nc + (kh * 4 + kw) * g.Cb + b];      // out-of-image small pixels were staged as zeros
            }
        }
        if (bias) acc += bias[b];
        big[((long)(n * g.Hb + h) * g.Wb + w) * ld_big + b] = pg_act_epi(acc, act);
    }
}

__global__ void k_col2im_small2big(const float* __restrict__ D, const float* __restrict__ bias, float* __restrict__ big,
                                   int ld_big, Geom g, int act) {
    const long total = (long)g.N * g.Hb * g.Wb * g.Cb;
    const int Nc = 16 * g.Cb;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx % g.Cb);
        const long pix = idx / g.Cb;
        const int n = (int)(pix / (g.Hb * g.Wb));
        const int rem = (int)(pix - (long)n * g.Hb * g.Wb);
        const int h = rem / g.Wb, w = rem - h * g.Wb;
        float acc = 0.f;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
            const int hh = h + 1 - kh;
            if (hh < 0 || (g.s == 2 && (hh & 1))) continue;
            const int ih = g.s == 2 ? hh >> 1 : hh;
            if (ih >= g.Hs) continue;
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int ww = w + 1 - kw;
                if (ww < 0 || (g.s == 2 && (ww & 1))) continue;
                const int iw = g.s == 2 ? ww >> 1 : ww;
                if (iw >= g.Ws) continue;
                acc += D[(long)((n * g.Hs + ih) * g.Ws + iw) * Nc + (kh * 4 + kw) * g.Cb + b];
            }
        }
        if (bias) acc += bias[b];
        big[pix * ld_big + b] = pg_act_epi(acc, act);
    }
}

__global__ void k_gather_big2small(const float* __restrict__ D, const float* __restrict__ bias,
                                   float* __restrict__ small, int ld_small, Geom g, int act) {
    const long total = (long)g.N * g.Hs * g.Ws * g.Ca;
    const int Nc = 16 * g.Ca;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int a = (int)(idx % g.Ca);
        const long pix = idx / g.Ca;
        const int n = (int)(pix / (g.Hs * g.Ws));
        const int rem = (int)(pix - (long)n * g.Hs * g.Ws);
        const int p = rem / g.Ws, q = rem - p * g.Ws;
        float acc = 0.f;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
            const int h = g.s * p - 1 + kh;
            if ((unsigned)h >= (unsigned)g.Hb) continue;
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int w = g.s * q - 1 + kw;
                if ((unsigned)w >= (unsigned)g.Wb) continue;
                acc += D[(long)((n * g.Hb + h) * g.Wb + w) * Nc + (kh * 4 + kw) * g.Ca + a];
            }
        }
        if (bias) acc += bias[a];
        small[pix * ld_small + a] = pg_act_epi(acc, act);
    }
}

__global__ void k_pack_taps_b(const float* __restrict__ P, float* __restrict__ Wp, int Ca, int Cb) {
    const int total = 16 * Ca * Cb;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int a = i % Ca;                 // Wp[(tap*Cb + b)*Ca + a]
        const int tb = i / Ca;
        const int b = tb % Cb, tap = tb / Cb;
        Wp[i] = P[(long)(tap * Ca + a) * Cb + b];
    }
}

// ------------------------------------------------------------------------------------------------
// split-K reduce (+ bias + activation):  out[r*ld_out + c] = act(sum_z slab[z][r*cols + c] + bias[c])
// ------------------------------------------------------------------------------------------------
__global__ void k_slab_reduce(const float* __restrict__ slabs, long slab_stride, int S, float* __restrict__ out,
                              int ld_out, long rows, int cols, const float* __restrict__ bias, int act, int out_bf) {
    const long total = rows * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        float v = slabs[i];
        for (int z = 1; z < S; ++z) v += slabs[(long)z * slab_stride + i];
        if (bias) v += bias[c];
        store_io(out, r * ld_out + c, pg_act(v, act), out_bf);
    }
}

__global__ void k_slab_reduce4(const float* __restrict__ slabs, long slab_stride, int S, float* __restrict__ out,
                               int ld_out, long rows, int cols, const float* __restrict__ bias, int act, int out_bf, pg_epi_mul mul) {
    const int cq = cols >> 2;
    const long total = rows * cq;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cq;
        const int c = (int)(i - r * cq) << 2;
        const long off = r * cols + c;
        f32x4 v = *reinterpret_cast<const f32x4*>(slabs + off);
        int z = 1;
        for (; z + 8 <= S; z += 8) {        // eight independent loads in flight, summed in slab order
            f32x4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(slabs + (long)(z + k) * slab_stride + off);
#pragma unroll
            for (int k = 0; k < 8; ++k) v += t[k];
        }
        for (; z < S; ++z) v += *reinterpret_cast<const f32x4*>(slabs + (long)z * slab_stride + off);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pg_act_epi(v[e], act);
        if (mul.t) {
            f32x4 tv;
            if (out_bf) {
                const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(mul.t) + r * mul.ld + c);
                tv = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            } else {
                tv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(mul.t) + r * mul.ld + c);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= pg_act_grad_from_out(tv[e], mul.act);
        }
        if (out_bf)
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + r * ld_out + c) = to_bf16(v);
        else
            *reinterpret_cast<f32x4*>(out + r * ld_out + c) = v;
    }
}

// many slabs, few outputs (the taps-in-N weight gradients: up to 512 slabs of a few thousand floats): 8 z-lanes per
// output share the slab loop, combined through LDS in lane order (deterministic)
__global__ __launch_bounds__(256) void k_slab_reduce_z(const float* __restrict__ slabs, long slab_stride, int S,
                                                       float* __restrict__ out, int ld_out, long rows, int cols) {
    __shared__ float red[8][32];
    const int ol = threadIdx.x & 31, zl = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + ol;
    const long total = rows * cols;
    float v = 0.f;
    if (i < total) {
        int z = zl;
        for (; z + 56 < S; z += 64) {          // eight independent loads in flight, added in slab order
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = slabs[(long)(z + 8 * k) * slab_stride + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) v += t[k];
        }
        for (; z < S; z += 8) v += slabs[(long)z * slab_stride + i];
    }
    red[zl][ol] = v;
    __syncthreads();
    if (zl == 0 && i < total) {
        float t = red[0][ol];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][ol];
        const long r = i / cols;
        out[r * ld_out + (i - r * cols)] = t;
    }
}

// column sums of a [rows][C] matrix (pixel stride ld): partial[chunk][c] over row chunks (bias gradient).
// 256 threads = 64 channels x 4 row lanes; fixed-order combine.
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ x, int ld, long rows, int C,
                                                        long rows_per_chunk, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long r_begin = blockIdx.x * rows_per_chunk;
    const long r_end = min(rows, r_begin + rows_per_chunk);
    float s = 0.f;
    if (c < C)
        for (long r = r_begin + rl; r < r_end; r += 4) s += x[r * ld + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) partial[(long)blockIdx.x * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// float4 variant (C % 4 == 0, 16-byte aligned rows): 256 threads = 16 channel quads x 16 row lanes, four independent
// loads in flight per lane; fixed-order combine.
template <bool HB>      // HB: x holds bf16 elements (bf16 activation storage)
__global__ __launch_bounds__(256) void k_colsum_partial4(const float* __restrict__ x, int ld, long rows, int C,
                                                         long rows_per_chunk, float* __restrict__ partial) {
    __shared__ f32x4 red[16][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cl * 4;
    const long r_begin = blockIdx.x * rows_per_chunk;
    const long r_end = min(rows, r_begin + rows_per_chunk);
    auto ld4 = [&](long idx) -> f32x4 {
        if constexpr (HB) {
            const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(x) + idx);
            return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        } else {
            return *reinterpret_cast<const f32x4*>(x + idx);
        }
    };
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c < C) {
        long r = r_begin + rl;
        for (; r + 48 < r_end; r += 64) {
            s0 += ld4(r * ld + c);
            s1 += ld4((r + 16) * ld + c);
            s2 += ld4((r + 32) * ld + c);
            s3 += ld4((r + 48) * ld + c);
        }
        for (; r < r_end; r += 16) s0 += ld4(r * ld + c);
    }
    red[rl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < C) {
        f32x4 t = red[0][cl];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += red[i][cl];
        *reinterpret_cast<f32x4*>(partial + (long)blockIdx.x * C + c) = t;
    }
}

// ------------------------------------------------------------------------------------------------
// direct kernels: one thread per output element, no LDS, no MFMA.  Any channel count / alignment.
// ------------------------------------------------------------------------------------------------
__global__ void k_big2small_direct(const float* __restrict__ big, int ld_big, const float* __restrict__ P,
                                   const float* __restrict__ bias, float* __restrict__ small, int ld_small, Geom g,
                                   int act) {
    const long total = (long)g.N * g.Hs * g.Ws * g.Ca;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int a = (int)(idx % g.Ca);
        const long m = idx / g.Ca;
        const int n = (int)(m / (g.Hs * g.Ws));
        const int rem = (int)(m - (long)n * g.Hs * g.Ws);
        const int p = rem / g.Ws, q = rem - p * g.Ws;
        float acc = 0.f;
        for (int kh = 0; kh < 4; ++kh) {
            const int h = g.s * p - 1 + kh;
            if ((unsigned)h >= (unsigned)g.Hb) continue;
            for (int kw = 0; kw < 4; ++kw) {
                const int w = g.s * q - 1 + kw;
                if ((unsigned)w >= (unsigned)g.Wb) continue;
                const float* xi = big + (long)((n * g.Hb + h) * g.Wb + w) * ld_big;
                const float* wi = P + (long)((kh * 4 + kw) * g.Ca + a) * g.Cb;
                for (int b = 0; b < g.Cb; ++b) acc = fmaf(xi[b], wi[b], acc);
            }
        }
        if (bias) acc += bias[a];
        small[m * ld_small + a] = pg_act(acc, act);
    }
}

// small -> big from a ONE-channel `small` (the discriminator head's data gradient, disc.py:45: Conv2d(512, 1, 4, 1, 1) backward):
// out[pixel][b] = sum over the <= 16 taps that reach the pixel of small[pixel'] * P[tap][0][b] -- 16 FMAs per output, bound by the
// HBM write of `big`.  A thread owns 4 consecutive b (its 16 tap weights = 16 float4 in registers) and walks pixels; the tap
// values of a pixel are the same address for every thread of the workgroup (one broadcast load each).  Output fp32 or bf16
// (out_bf); mul: times f'(t) of the layer below (pg_epi_mul; t in the output's storage type).  Many short workgroups: the per-pixel
// chain (16 loads, FMAs, store) is latency-bound, ~8 workgroups per CU hide it (64-pixel workgroups: 70 us, 7-pixel ones: 40 us;
// a 4-pixel software pipeline per thread: slower, 64 more registers).
// Stride-1 form (the discriminator head's data gradient, disc.py:45: Cout = 1, 4x4, stride 1): a workgroup first stages its SAMPLE's
// whole one-channel map (Hs x Ws floats, a few KB) into LDS inside a zero border of two, so that the 16 taps of a pixel are 16
// unmasked, conflict-free broadcast reads -- the generic kernel below fetched them with 16 global loads per pixel AND PER WAVE (64
// identical addresses each: ~1 M broadcast loads per launch through the address path, 1.4 TB/s of output where stores alone run 6).
// grid (workgroups per sample, N); a thread keeps its channel quad's 16 weight vectors in registers and walks pixels.
__global__ __launch_bounds__(256) void k_s2b_ca1_s1(const float* __restrict__ small, int ld_small, const float* __restrict__ P,
                                                    const float* __restrict__ bias, float* __restrict__ big, int ld_big, Geom g, int act,
                                                    int out_bf, pg_epi_mul mul, int pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [(Hs + 4)][(Ws + 4)], the map at (2, 2)
    const int n = blockIdx.y, PW = g.Ws + 4, HWs = g.Hs * g.Ws, HWb = g.Hb * g.Wb;
    for (int i = threadIdx.x; i < (g.Hs + 4) * PW; i += 256) xs[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < HWs; i += 256) {
        const int ih = i / g.Ws, iw = i - ih * g.Ws;
        xs[(ih + 2) * PW + iw + 2] = small[(long)(n * HWs + i) * ld_small];
    }
    const int cq = g.Cb >> 2;
    const int qd = threadIdx.x % cq, np = 256 / cq;
    const int lane_p = (cq % 64 == 0) ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / cq) : (int)threadIdx.x / cq;
    const int b = qd * 4;
    f32x4 w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = *reinterpret_cast<const f32x4*>(P + (long)t * g.Cb + b);      // Ca == 1: P[tap][0][b]
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + b);
    __syncthreads();
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HWb, p0 + pix_per_block);
    for (int pix = p0 + lane_p; pix < p1; pix += np) {
        const int h = pix / g.Wb, wq = pix - h * g.Wb;
        // tap (kh, kw) reads small[h + 1 - kh][wq + 1 - kw] = xs[h + 3 - kh][wq + 3 - kw]
        const float* xr = xs + (h + 3) * PW + wq + 3;
        f32x4 acc = bv;
        // The tap values never reach a packed multiply straight from the LDS return: `acc += x * w` compiles to v_pk_mul_f32 with the
        // scalar x broadcast through op_sel, and where that takes its LOW half from the ODD register of a pair that a ds_read2_b32
        // has just returned, lanes 48-63 of the low half were seen to use the register's OLD content although s_waitcnt lgkmcnt had
        // been satisfied (one tap of e0 / e2 dropped, ~1 call in 40, only next to another process on the GPU: EXPERIMENTS.md,
        // "k_s2b_ca1_s1: cause").  A plain VALU read of the returned register is safe (v_readfirstlane where the wave's pixel is
        // uniform -- the values then live in scalar registers --, a v_mov_b32 otherwise); tests/test_codeobj_cpu.py scans the
        // library for the failing operand form.
        float xv[16];
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) xv[kh * 4 + kw] = xr[-kh * PW - kw];
        if (cq % 64 == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) xv[t] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xv[t])));
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) asm volatile("v_mov_b32 %0, %1" : "=v"(xv[t]) : "v"(xv[t]));
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc += xv[t] * w[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = pg_act_epi(acc[e], act);
        const long m = (long)n * HWb + pix;
        if (mul.t) {
            f32x4 tv;
            if (out_bf) {
                const bf16x4 hv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(mul.t) + m * mul.ld + b);
                tv = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
            } else {
                tv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(mul.t) + m * mul.ld + b);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] *= pg_act_grad_from_out(tv[e], mul.act);
        }
        if (out_bf)
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(big) + m * ld_big + b) = to_bf16(acc);
        else
            *reinterpret_cast<f32x4*>(big + m * ld_big + b) = acc;
    }
}

__global__ __launch_bounds__(256) void k_s2b_ca1(const float* __restrict__ small, int ld_small, const float* __restrict__ P,
                                                 const float* __restrict__ bias, float* __restrict__ big, int ld_big, Geom g, int act,
                                                 int out_bf, pg_epi_mul mul, int pix_per_block) {
    constexpr int PP = 2;                                          // pixels per thread and trip
    const int cq = g.Cb >> 2;                                      // channel quads
    // host: 256 % cq == 0, cq <= 256.  With >= 64 quads (Cb >= 256: the discriminator head at ndf >= 32) a wave works on ONE pixel:
    // its pixel index is made provably wave-uniform (readfirstlane), so the pixel decode, the 16 tap addresses and the tap loads
    // themselves run on the scalar unit (s_load) instead of once per lane -- the kernel was bound by that per-lane integer work
    const int qd = threadIdx.x % cq, np = 256 / cq;
    const int lane_p = (cq % 64 == 0) ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / cq) : (int)threadIdx.x / cq;
    const int b = qd * 4;
    f32x4 w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = *reinterpret_cast<const f32x4*>(P + (long)t * g.Cb + b);      // Ca == 1: P[tap][0][b]
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + b);
    const int total = g.N * g.Hb * g.Wb;                           // (host: < 2^31; 32-bit pixel decode: a 64-bit division per pixel
    const int p0 = blockIdx.x * pix_per_block;                     //  was the single most expensive thing in this kernel)
    const int p1 = min(total, p0 + pix_per_block);
    const int HWb = g.Hb * g.Wb;
    for (int m0 = p0 + lane_p; m0 < p1; m0 += np * PP) {
        // the taps' inputs: UNCONDITIONAL loads from clamped addresses, masked afterwards -- all of them (and the multiplier's loads)
        // are in flight together; `ok ? small[..] : 0` compiled to a branch and a wait per tap
        float xv[PP][16];
        f32x4 tv[PP];
        int mm[PP];
        bool live[PP];
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int m = m0 + u * np;
            live[u] = m < p1;
            mm[u] = live[u] ? m : p1 - 1;
            const int n = mm[u] / HWb;
            const int rem = mm[u] - n * HWb;
            const int h = rem / g.Wb, wq = rem - h * g.Wb;
#pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const int hh = h + 1 - kh;
                const int ih = (g.s == 2) ? hh >> 1 : hh;
                const bool okh = (hh >= 0) & !((g.s == 2) & (hh & 1)) & (ih < g.Hs);
                const int ihc = min(max(ih, 0), g.Hs - 1);
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int ww = wq + 1 - kw;
                    const int iw = (g.s == 2) ? ww >> 1 : ww;
                    const bool ok = okh & (ww >= 0) & !((g.s == 2) & (ww & 1)) & (iw < g.Ws);
                    const int iwc = min(max(iw, 0), g.Ws - 1);
                    const float x = small[(long)((n * g.Hs + ihc) * g.Ws + iwc) * ld_small];
                    xv[u][kh * 4 + kw] = ok ? x : 0.f;
                }
            }
            if (mul.t) {
                if (out_bf) {
                    const bf16x4 hv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(mul.t) + (long)mm[u] * mul.ld + b);
                    tv[u] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                } else {
                    tv[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(mul.t) + (long)mm[u] * mul.ld + b);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            f32x4 acc = bv;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += xv[u][t] * w[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = pg_act_epi(acc[e], act);
            if (mul.t) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] *= pg_act_grad_from_out(tv[u][e], mul.act);
            }
            if (live[u]) {
                if (out_bf)
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(big) + (long)mm[u] * ld_big + b) = to_bf16(acc);
                else
                    *reinterpret_cast<f32x4*>(big + (long)mm[u] * ld_big + b) = acc;
            }
        }
    }
}

__global__ void k_small2big_direct(const float* __restrict__ small, int ld_small, const float* __restrict__ P,
                                   const float* __restrict__ bias, float* __restrict__ big, int ld_big, Geom g,
                                   int act) {
    const long total = (long)g.N * g.Hb * g.Wb * g.Cb;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx % g.Cb);
        const long m = idx / g.Cb;
        const int n = (int)(m / (g.Hb * g.Wb));
        const int rem = (int)(m - (long)n * g.Hb * g.Wb);
        const int h = rem / g.Wb, w = rem - h * g.Wb;
        float acc = 0.f;
        for (int kh = 0; kh < 4; ++kh) {
            const int hh = h + 1 - kh;
            if (hh < 0 || (hh % g.s) != 0) continue;
            const int ih = hh / g.s;
            if (ih >= g.Hs) continue;
            for (int kw = 0; kw < 4; ++kw) {
                const int ww = w + 1 - kw;
                if (ww < 0 || (ww % g.s) != 0) continue;
                const int iw = ww / g.s;
                if (iw >= g.Ws) continue;
                const float* xi = small + (long)((n * g.Hs + ih) * g.Ws + iw) * ld_small;
                const float* wi = P + (long)(kh * 4 + kw) * g.Ca * g.Cb + b;
                for (int a = 0; a < g.Ca; ++a) acc = fmaf(xi[a], wi[(long)a * g.Cb], acc);
            }
        }
        if (bias) acc += bias[b];
        big[m * ld_big + b] = pg_act(acc, act);
    }
}

// one thread per (slice, tap, a, b); pixels of the slice summed serially (deterministic)
__global__ void k_wgrad_direct(const float* __restrict__ small, int ld_small, const float* __restrict__ big,
                               int ld_big, float* __restrict__ out, long slab_stride, Geom g, long pix_per_slice) {
    const long per = 16L * g.Ca * g.Cb;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= per) return;
    const int b = (int)(idx % g.Cb);
    const int a = (int)((idx / g.Cb) % g.Ca);
    const int tap = (int)(idx / ((long)g.Cb * g.Ca));
    const int kh = tap >> 2, kw = tap & 3;
    const long Kp = (long)g.N * g.Hs * g.Ws;
    const long p_begin = blockIdx.y * pix_per_slice;
    const long p_end = min(Kp, p_begin + pix_per_slice);
    float acc = 0.f;
    for (long pix = p_begin; pix < p_end; ++pix) {
        const int n = (int)(pix / (g.Hs * g.Ws));
        const int rem = (int)(pix - (long)n * g.Hs * g.Ws);
        const int p = rem / g.Ws, q = rem - p * g.Ws;
        const int h = g.s * p - 1 + kh, w = g.s * q - 1 + kw;
        if ((unsigned)h >= (unsigned)g.Hb || (unsigned)w >= (unsigned)g.Wb) continue;
        acc = fmaf(small[pix * ld_small + a], big[(long)((n * g.Hb + h) * g.Wb + w) * ld_big + b], acc);
    }
    out[(long)blockIdx.y * slab_stride + idx] = acc;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Optional per-launch timing (bench.py's roofline leg): pg_conv_time_next() arms two caller-owned events; the next conv
// entry point on this thread records them immediately before and after its main GEMM kernel (not the split-K reduce,
// not the bias column-sum), then disarms.  Thread-local; no effect unless armed.
thread_local hipEvent_t t_ev0 = nullptr, t_ev1 = nullptr;
thread_local hipEvent_t t_ev2 = nullptr, t_ev3 = nullptr;   // second pair: the second GEMM of pg_conv4x4_bwd_big
struct TimedLaunch {
    hipStream_t st;
    hipEvent_t e1;
    explicit TimedLaunch(hipStream_t s) : st(s), e1(t_ev1) {
        if (t_ev0) (void)hipEventRecord(t_ev0, st);
        t_ev0 = nullptr;
        t_ev1 = nullptr;
    }
    ~TimedLaunch() {
        if (e1) (void)hipEventRecord(e1, st);
    }
};

// bytes spanned by a [pixels][C] view with pixel stride ld (what the buffer descriptor of the fast kernels covers)
inline long tensor_bytes(long pixels, int ld, int C, bool bf = false) { return ((pixels - 1) * (long)ld + C) * (bf ? 2 : 4); }
inline bool aligned_io(const void* p, bool bf) { return (reinterpret_cast<uintptr_t>(p) & (bf ? 7 : 15)) == 0; }
constexpr long FAST_LIMIT = 0x60000000L;   // 1.5 GiB: keeps every 32-bit byte offset, incl. the +0x40000000 sentinel, < 2^32
constexpr long FAST_P_LIMIT = 0x40000000L; // packed weights: the invalid-row sentinel (+1 GiB) must land beyond the block
inline bool force_generic() {
    static const bool v = pg_exp_env("PATCHGAN_GENERIC_KERNELS") != nullptr;   // debugging aid: disable the fast variants
    return v;
}

bool geom_ok(const pg_conv_geom* g) {
    if (!g) return false;
    if (g->N <= 0 || g->Hb <= 0 || g->Wb <= 0 || g->Hs <= 0 || g->Ws <= 0 || g->Ca <= 0 || g->Cb <= 0) return false;
    if (g->stride != 1 && g->stride != 2) return false;
    // small = floor((big + 2 - 4)/s) + 1 (nn.Conv2d) ; for stride 2 the big side may also be 2*small (convT)
    const int hs = (g->Hb - 2) / g->stride + 1, ws = (g->Wb - 2) / g->stride + 1;
    if (g->Hb < 2 || g->Wb < 2) return false;
    if (hs != g->Hs || ws != g->Ws) return false;
    // 32-bit index headroom inside the kernels
    const double big_elems = (double)g->N * g->Hb * g->Wb, small_elems = (double)g->N * g->Hs * g->Ws;
    if (big_elems > 2.0e9 || small_elems > 2.0e9) return false;
    if (16.0 * g->Ca * g->Cb > 2.0e9) return false;
    return true;
}

Geom to_geom(const pg_conv_geom* g) { return Geom{g->N, g->Hb, g->Wb, g->Hs, g->Ws, g->Ca, g->Cb, g->stride}; }

struct Tile {
    int id, bm, bn;
};
// tile ids: 0 = 128x128, 1 = 128x64, 2 = 128x32, 3 = 64x128, 4 = 64x64
Tile pick_tile(long rows, int cols) {
    if (rows > 64) {
        if (cols > 64) return {0, 128, 128};
        if (cols > 32) return {1, 128, 64};
        return {2, 128, 32};
    }
    if (cols > 64) return {3, 64, 128};
    if (cols > 32) return {4, 64, 64};
    return {2, 128, 32};
}

// Split K until two workgroups per CU exist: one 4-wave workgroup per CU leaves the MFMA pipe idle while that workgroup stages its next tile
// (measured 55 vs 80 TFLOP/s on the same kernel ALONE on the chip at 1 vs 2 workgroups per CU).  Round 6: in the two-stream training step the
// kernel is not alone and a target of 256 measured FASTER (cfg2 step, same box, target 128 / 192 / 256 / 320 / 384 / 512 / 1024: 7.48 / 7.13 /
// 7.03 / 7.20 / 7.13 / 7.14 / 7.08 ms) -- not adopted: the other slices are another order of the sums, and the cfg2 loss curve, a chaotic
// quantity, then sits at 1.8e-4 from the reference's (step 3) instead of inside the north star's 1e-4 (tests/test_step_gpu.py).
constexpr int TARGET_BLOCKS = 512;

int pick_split(long tiles, int nchunks, int min_chunks) {
    static const int target = pg_exp_env("PATCHGAN_SPLIT_TARGET") ? atoi(pg_exp_env("PATCHGAN_SPLIT_TARGET")) : TARGET_BLOCKS;
    if (tiles >= target) return 1;
    static const int minc = pg_exp_env("PATCHGAN_SPLIT_MINCHUNKS") ? atoi(pg_exp_env("PATCHGAN_SPLIT_MINCHUNKS")) : 0;
    if (minc > 0) min_chunks = minc;
    long s = (target + tiles - 1) / tiles;
    long smax = nchunks / min_chunks;
    if (smax < 1) smax = 1;
    if (s > smax) s = smax;
    if (s > 1024) s = 1024;
    return (int)s;
}

struct Plan {
    Tile t;
    int tiles_m, tiles_n, ncls, nchunks, split, cps;
    long out_elems;   // elements of one slab
};

// A 128x128 tiling that yields 256..511 workgroups would need split-K 2 (slab write + reduce pass) to reach two
// workgroups per CU; the 128x64 tile reaches the same occupancy without the slabs at the same MFMA efficiency.
Tile refine_tile(Tile t, long rows, int cols, int ncls) {
    static const bool off = pg_exp_env("PATCHGAN_TILE_REFINE") == nullptr;   // measured slower than split-K 2: off by default
    if (off || t.id != 0) return t;
    const long tiles = ((rows + 127) / 128) * ((cols + 127) / 128) * ncls;
    if (tiles >= 256 && tiles < 512) return {1, 128, 64};
    return t;
}

Plan plan_b2s(const pg_conv_geom* g) {
    Plan p;
    const long M = (long)g->N * g->Hs * g->Ws;
    p.t = refine_tile(pick_tile(M, g->Ca), M, g->Ca, 1);
    p.tiles_m = (int)((M + p.t.bm - 1) / p.t.bm);
    p.tiles_n = (g->Ca + p.t.bn - 1) / p.t.bn;
    p.ncls = 1;
    p.nchunks = (16 * g->Cb + KC - 1) / KC;
    p.split = pick_split((long)p.tiles_m * p.tiles_n, p.nchunks, 8);
    p.out_elems = M * g->Ca;
    return p;
}

Plan plan_s2b(const pg_conv_geom* g) {
    Plan p;
    p.ncls = (g->stride == 2) ? 4 : 1;
    const int Hc = (g->stride == 2) ? (g->Hb + 1) / 2 : g->Hb, Wc = (g->stride == 2) ? (g->Wb + 1) / 2 : g->Wb;
    const long Mc = (long)g->N * Hc * Wc;   // largest class
    p.t = refine_tile(pick_tile(Mc, g->Cb), Mc, g->Cb, p.ncls);
    p.tiles_m = (int)((Mc + p.t.bm - 1) / p.t.bm);
    p.tiles_n = (g->Cb + p.t.bn - 1) / p.t.bn;
    const int taps = (g->stride == 2) ? 4 : 16;
    p.nchunks = (taps * g->Ca + KC - 1) / KC;
    p.split = pick_split((long)p.tiles_m * p.tiles_n * p.ncls, p.nchunks, 8);
    p.out_elems = (long)g->N * g->Hb * g->Wb * g->Cb;
    return p;
}

// wgrad kernel choice: 0 = one GEMM per tap, 1 / 2 = taps folded into N (few big-side / small-side channels)
int wgrad_mode(const pg_conv_geom* g) {
    if (g->Cb <= 8) return 1;
    if (g->Ca <= 8) return 2;
    return 0;
}

Plan plan_wgrad(const pg_conv_geom* g) {
    Plan p;
    const int mode = wgrad_mode(g);
    long Kp;
    if (mode == 0) {
        p.t = pick_tile(g->Ca, g->Cb);
        p.tiles_m = (g->Ca + p.t.bm - 1) / p.t.bm;
        p.tiles_n = (g->Cb + p.t.bn - 1) / p.t.bn;
        p.ncls = 16;
        Kp = (long)g->N * g->Hs * g->Ws;
    } else {
        const int Mdim = (mode == 1) ? g->Ca : g->Cb, Ndim = 16 * ((mode == 1) ? g->Cb : g->Ca);
        p.t = pick_tile(Mdim, Ndim);
        p.tiles_m = (Mdim + p.t.bm - 1) / p.t.bm;
        p.tiles_n = (Ndim + p.t.bn - 1) / p.t.bn;
        p.ncls = 1;
        Kp = (mode == 1) ? (long)g->N * g->Hs * g->Ws : (long)g->N * g->Hb * g->Wb;
    }
    p.nchunks = (int)((Kp + KC - 1) / KC);
    p.split = pick_split((long)p.tiles_m * p.tiles_n * p.ncls, p.nchunks, 8);
    if (mode != 0) {
        // the taps-in-N kernels stream a huge K (every pixel) into one or two tiles: latency-bound per workgroup, so they
        // want far more, shorter slices than the MFMA-bound kernels (measured 83 us at 512 slices vs 187 us at 128)
        static const int tt = pg_exp_env("PATCHGAN_TAPN_SPLIT") ? atoi(pg_exp_env("PATCHGAN_TAPN_SPLIT")) : 1024;
        long want = tt / ((long)p.tiles_m * p.tiles_n);
        long smax = p.nchunks / 4;
        if (want > smax) want = smax;
        p.split = want < 1 ? 1 : (int)want;
    }
    p.out_elems = 16L * g->Ca * g->Cb;
    return p;
}

constexpr int DIRECT_WGRAD_SLICES = 64;
constexpr int COLSUM_CHUNKS = 1024;

void clamp_split(Plan& p, size_t ws_bytes, size_t reserved) {
    size_t avail = ws_bytes > reserved ? ws_bytes - reserved : 0;
    long smax = (long)(avail / (sizeof(float) * (size_t)p.out_elems));
    if (p.split > 1 && smax < p.split) p.split = smax < 2 ? 1 : (int)smax;
    p.cps = (p.nchunks + p.split - 1) / p.split;
    p.split = (p.nchunks + p.cps - 1) / p.cps;   // drop empty slices
}

int launch_reduce(const float* slabs, long slab_stride, int S, float* out, int ld_out, long rows, int cols,
                  const float* bias, int act, hipStream_t st, int out_bf = 0, pg_epi_mul mul = pg_epi_mul{nullptr, 0, 0}) {
    if (S >= 64 && !bias && act == PG_ACT_NONE && rows * cols <= (1L << 22) && !out_bf && !mul.t) {
        const long total = rows * cols;
        hipLaunchKernelGGL(k_slab_reduce_z, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, slabs, slab_stride, S, out,
                           ld_out, rows, cols);
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    if ((cols % 4 == 0) && (ld_out % 4 == 0) && (slab_stride % 4 == 0) && aligned16(slabs) &&
        ((reinterpret_cast<uintptr_t>(out) & (out_bf ? 7 : 15)) == 0) && (!bias || aligned16(bias))) {
        const long total4 = rows * (cols / 4);
        int blocks = (int)std::min<long>((total4 + 255) / 256, 16384);
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_slab_reduce4, dim3(blocks), dim3(256), 0, st, slabs, slab_stride, S, out, ld_out, rows, cols,
                           bias, act, out_bf, mul);
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    if (mul.t) return PG_EINVAL;       // (the callers that pass a multiplier satisfy the alignment of the vector form)
    const long total = rows * cols;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_slab_reduce, dim3(blocks), dim3(256), 0, st, slabs, slab_stride, S, out, ld_out, rows, cols,
                       bias, act, out_bf);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

#define PG_DISPATCH_TILE(KERNEL, tile_id, grid, st, ...)                                                   \
    switch (tile_id) {                                                                                     \
        case 0: hipLaunchKernelGGL((KERNEL<2, 2, 2, 2>), grid, dim3(256), 0, st, __VA_ARGS__); break;      \
        case 1: hipLaunchKernelGGL((KERNEL<2, 1, 2, 2>), grid, dim3(256), 0, st, __VA_ARGS__); break;      \
        case 2: hipLaunchKernelGGL((KERNEL<1, 1, 4, 1>), grid, dim3(256), 0, st, __VA_ARGS__); break;      \
        case 3: hipLaunchKernelGGL((KERNEL<1, 2, 2, 2>), grid, dim3(256), 0, st, __VA_ARGS__); break;      \
        default: hipLaunchKernelGGL((KERNEL<1, 1, 2, 2>), grid, dim3(256), 0, st, __VA_ARGS__); break;     \
    }

#define PG_DISPATCH_TAPN(MODE, tile_id, grid, st, ...)                                                              \
    switch (tile_id) {                                                                                             \
        case 0: hipLaunchKernelGGL((k_wgrad_tapn<2, 2, 2, 2, MODE>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        case 1: hipLaunchKernelGGL((k_wgrad_tapn<2, 1, 2, 2, MODE>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        case 2: hipLaunchKernelGGL((k_wgrad_tapn<1, 1, 4, 1, MODE>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        case 3: hipLaunchKernelGGL((k_wgrad_tapn<1, 2, 2, 2, MODE>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        default: hipLaunchKernelGGL((k_wgrad_tapn<1, 1, 2, 2, MODE>), grid, dim3(256), 0, st, __VA_ARGS__); break; \
    }

// taps-folded-into-N forward paths: eligibility and workspace (floats)
inline bool s2b_tapn_ok(const Geom& g) { return g.Cb <= 8 && g.Ca % KC == 0 && !force_generic(); }
// the one-pass form of it (k_s2b_tapnf): stride 2 onto <= 8 channels (5 .. 8: two launches of <= 4) from 32 / 64 / 128
inline bool tapnf_enabled() {
    static const bool off = pg_exp_env("PATCHGAN_NO_TAPNF") != nullptr;
    return !off;
}
inline bool s2b_tapnf_ok(const Geom& g) {
    return g.s == 2 && g.Cb <= 8 && (g.Ca == 32 || g.Ca == 64 || g.Ca == 128) && !force_generic() && tapnf_enabled();
}
struct Tune;
inline bool s2b_tapnf_bf_ok(const Geom& g, int algo_full, const Tune& t);      // the same kernel on a bf16 `small` (PG_ALGO_BF16), below Tune
// one launch of k_s2b_tapnf (the caller has checked eligibility, alignment and the 32-bit offset limits)
static int launch_tapnf(bool bf, const void* small, int ld_small, const float* P, const float* bias, float* big, int ld_big, const Geom& g,
                        int act, long small_bytes, long big_bytes, hipStream_t st);
inline bool b2s_tapn_ok(const Geom& g) { return g.Ca <= 8 && g.Cb % KC == 0 && !force_generic(); }
inline size_t s2b_tapn_ws(const Geom& g) {
    return ((size_t)16 * g.Cb * g.Ca + (size_t)g.N * g.Hs * g.Ws * 16 * g.Cb) * sizeof(float) + 256;
}
inline size_t b2s_tapn_ws(const Geom& g) { return (size_t)g.N * g.Hb * g.Wb * 16 * g.Ca * sizeof(float); }

#define PG_DISPATCH_B2SF(ONE, tile_id, grid, st, ...)                                                              \
    switch (tile_id) {                                                                                            \
        case 0: hipLaunchKernelGGL((k_b2s_fast<2, 2, 2, 2, ONE>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        case 1: hipLaunchKernelGGL((k_b2s_fast<2, 1, 2, 2, ONE>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        case 2: hipLaunchKernelGGL((k_b2s_fast<1, 1, 4, 1, ONE>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        case 3: hipLaunchKernelGGL((k_b2s_fast<1, 2, 2, 2, ONE>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        default: hipLaunchKernelGGL((k_b2s_fast<1, 1, 2, 2, ONE>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
    }

#define PG_DISPATCH_B2SH(HIN, tile_id, grid, st, ...)                                                                    \
    switch (tile_id) {                                                                                                  \
        case 0: hipLaunchKernelGGL((k_b2s_bf16<2, 2, 2, 2, false, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
        case 1: hipLaunchKernelGGL((k_b2s_bf16<2, 1, 2, 2, false, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
        case 2: hipLaunchKernelGGL((k_b2s_bf16<1, 1, 4, 1, false, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
        case 3: hipLaunchKernelGGL((k_b2s_bf16<1, 2, 2, 2, false, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
        default: hipLaunchKernelGGL((k_b2s_bf16<1, 1, 2, 2, false, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
    }
#define PG_DISPATCH_S2BH(HIN, tile_id, grid, st, ...)                                                              \
    switch (tile_id) {                                                                                            \
        case 0: hipLaunchKernelGGL((k_s2b_bf16<2, 2, 2, 2, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        case 1: hipLaunchKernelGGL((k_s2b_bf16<2, 1, 2, 2, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        case 2: hipLaunchKernelGGL((k_s2b_bf16<1, 1, 4, 1, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        case 3: hipLaunchKernelGGL((k_s2b_bf16<1, 2, 2, 2, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        default: hipLaunchKernelGGL((k_s2b_bf16<1, 1, 2, 2, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
    }
#define PG_DISPATCH_WGH(POW2, HIN, tile_id, grid, st, ...)                                                              \
    switch (tile_id) {                                                                                                 \
        case 0: hipLaunchKernelGGL((k_wgrad_bf16<2, 2, 2, 2, POW2, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        case 1: hipLaunchKernelGGL((k_wgrad_bf16<2, 1, 2, 2, POW2, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        case 2: hipLaunchKernelGGL((k_wgrad_bf16<1, 1, 4, 1, POW2, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        case 3: hipLaunchKernelGGL((k_wgrad_bf16<1, 2, 2, 2, POW2, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        default: hipLaunchKernelGGL((k_wgrad_bf16<1, 1, 2, 2, POW2, HIN>), grid, dim3(256), 0, st, __VA_ARGS__); break; \
    }

#define PG_DISPATCH_WGF(POW2, tile_id, grid, st, ...)                                                              \
    switch (tile_id) {                                                                                            \
        case 0: hipLaunchKernelGGL((k_wgrad_fast<2, 2, 2, 2, POW2>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        case 1: hipLaunchKernelGGL((k_wgrad_fast<2, 1, 2, 2, POW2>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        case 2: hipLaunchKernelGGL((k_wgrad_fast<1, 1, 4, 1, POW2>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        case 3: hipLaunchKernelGGL((k_wgrad_fast<1, 2, 2, 2, POW2>), grid, dim3(256), 0, st, __VA_ARGS__); break;  \
        default: hipLaunchKernelGGL((k_wgrad_fast<1, 1, 2, 2, POW2>), grid, dim3(256), 0, st, __VA_ARGS__); break; \
    }

inline bool tapkp_enabled() {          // PATCHGAN_TAPK_ONESHOT=1 (experiment): the one-shot k_b2s_tapk instead of the persistent k_b2s_tapkp
    static const bool off = [] {
        const char* e = pg_exp_env("PATCHGAN_TAPK_ONESHOT");
        return e && e[0] == '1';
    }();
    return !off;
}
// persistent taps-in-N weight gradient (k_wgrad_tapnp): <= 4 big-side channels, enough 128-pixel tiles to give every workgroup a few
inline int wgrad_tapnp_slabs(const Geom& g) {
    const long tiles = ((long)g.N * g.Hs * g.Ws + 127) / 128;
    const long rows = (g.Ca + (g.Cb <= 2 ? 127 : 63)) / (g.Cb <= 2 ? 128 : 64);
    return (int)std::max<long>(1, std::min<long>(tiles, 512 / rows));
}
inline bool wgrad_tapnp_ok(const Geom& g) {
    return g.Cb <= 4 && g.Ca % 4 == 0 && (long)g.N * g.Hs * g.Ws >= 4096 && (long)g.N * g.Hs * g.Ws < 0x3fffffL * 64 && !force_generic() &&
           tapkp_enabled();
}
// chunks per sample of the persistent image-facing kernel's InstanceNorm partial sums (2 per 128-pixel tile: one per wave row); 0: n/a
inline int tapkp_stats_chunks(const Geom& g) {
    const long hw = (long)g.Hs * g.Ws;
    if (g.Cb > 4 || g.Ca % 4 != 0 || hw % 128 != 0 || (g.Ca <= 8 && g.Cb % KC == 0) || force_generic()) return 0;
    return (int)(2 * (hw / 128));
}
inline bool tapk_enabled() {
    static const bool off = [] {
        const char* e = pg_exp_env("PATCHGAN_NO_TAPK");
        return e && e[0] == '1';
    }();
    return !off;
}
// Per-call tuning: the PG_TUNE_* bits of the `algo` argument over the process-wide defaults (the PATCHGAN_* environment
// switches, read once: they exist for same-device A/B timing; tests and callers use the bits).
struct Tune {
    bool wino;      // any Winograd path
    bool winow;     // stride-1 Winograd weight gradient
    int wino2;      // polyphase stride-2 forward / data gradient: 0 off, 1 wherever the geometry allows, 2 size heuristic
    int wino2w;     // polyphase stride-2 weight gradient: same codes
    int mo1;        // stride-1 tile edge: 0 heuristic, 2 / 3 pinned
    int dma;        // stride-1 GEMM staging: 0 registers, 1 LDS-DMA ring for F(3x3,4x4), 2 also for 64-tile F(2x2,4x4)
    bool bf16x;     // PG_ALGO_BF16 on bf16 tensors: the LDS-DMA kernels of conv_bf16.hip (off: the register-staged k_*_bf16)
    int bf16ring;   // ... their staging: 1 three-stage ring of 32-wide chunks, 0 one buffer of 64-wide chunks, -1 per-layer default
    bool s3;        // polyphase Winograd GEMMs in split-bf16 form (k_wino_bgemm_s3) instead of v_mfma_f32_32x32x2_f32 (k_wino_bgemm)
    bool s3w;       // the Winograd weight-gradient GEMMs likewise (k_wino_wgrad_gemm_s3 instead of k_wino_wgrad_gemm)
    bool s3r;       // the row-fused stride-1 F(3x3,4x4) GEMM likewise (k_wino_gemm_row_s3 instead of k_wino_gemm_row)
};
inline int env_int(const char* name, int dflt) {
    const char* e = pg_exp_env(name);
    return e ? atoi(e) : dflt;
}
inline Tune tune_of(int algo) {
    static const Tune env = [] {
        Tune t;
        t.wino = env_int("PATCHGAN_NO_WINOGRAD", 0) != 1;
        t.winow = env_int("PATCHGAN_NO_WINOGRAD_WGRAD", 0) != 1;
        t.wino2 = env_int("PATCHGAN_WINO2", 2);
        t.wino2w = env_int("PATCHGAN_WINO2_WGRAD", 2);
        t.mo1 = env_int("PATCHGAN_WINO1_TILE", 0);
        t.dma = pg_wino_dma_mode();
        t.bf16x = env_int("PATCHGAN_NO_BF16X", 0) != 1;
        t.bf16ring = -1;      // (pinned only per call: PG_TUNE_BF16X_RING / _FLAT)
        t.s3 = env_int("PATCHGAN_S3", 1) != 0;
        t.s3w = env_int("PATCHGAN_S3W", 1) != 0;
        t.s3r = env_int("PATCHGAN_S3R", 1) != 0;
        return t;
    }();
    Tune t = env;
    if (algo & PG_TUNE_WINO_OFF) t.wino = false;
    if (algo & PG_TUNE_WINOW_OFF) t.winow = false;
    if (algo & PG_TUNE_WINO2_ALL) t.wino2 = 1;
    if (algo & PG_TUNE_WINO2_OFF) t.wino2 = 0;
    if (algo & PG_TUNE_WINO2W_ALL) t.wino2w = 1;
    if (algo & PG_TUNE_WINO2W_OFF) t.wino2w = 0;
    if (algo & PG_TUNE_WINO1_F2) t.mo1 = 2;
    if (algo & PG_TUNE_WINO1_F3) t.mo1 = 3;
    if (algo & PG_TUNE_WINO_DMA) t.dma = 2;
    if (algo & PG_TUNE_BF16X_OFF) t.bf16x = false;
    if (algo & PG_TUNE_BF16X_RING) t.bf16ring = 1;
    if (algo & PG_TUNE_BF16X_FLAT) t.bf16ring = 0;
    if (algo & PG_TUNE_S3_OFF) t.s3 = t.s3w = t.s3r = false;
    if (force_generic()) t.wino = false;
    return t;
}
// every path on, for sizing a workspace that serves any tuning
inline Tune tune_widest(int mo1) { return Tune{true, true, 1, 1, mo1, 0, true, -1, true, true, true}; }

// stride-1 layers only: forward is a pad-1 correlation big -> small, the data gradient a pad-2 correlation small -> big
inline bool wino_b2s_ok(const Geom& g, const Tune& t) {
    return g.s == 1 && t.wino && pg_wino_geom_ok(g.N, g.Hs, g.Ws, g.Cb, g.Ca, t.mo1);
}
inline bool wino_s2b_ok(const Geom& g, const Tune& t) {
    return g.s == 1 && t.wino && pg_wino_geom_ok(g.N, g.Hb, g.Wb, g.Ca, g.Cb, t.mo1);
}
// stride-2 big -> small polyphase Winograd: where the channel counts are large against the tile count (measured per
// layer of cfg2, tools/layer_bench.py) unless the tuning forces it on / off
inline bool wino2_b2s_ok(const Geom& g, const Tune& t) {
    if (g.s != 2 || t.wino2 == 0 || !t.wino || !pg_wino2_geom_ok(g.N, g.Hs, g.Ws, g.Ca, g.Cb)) return false;
    if (t.wino2 == 1) return true;
    // measured on the cfg2 layers (F(3x3,2x2)): 128->256 ch -31 %, 256->512 on 16x16 -11 %, 256->1024 -25 %, 128->512 -35 %,
    // 64->256 -18 %, 64->128 -10..-16 %; 512->512 on 8x8 and smaller maps slower (weight transform dominates)
    return pg_wino2_tiles_b2s(g.N, g.Hs, g.Ws) >= 512 && g.Cb >= 64 && g.Ca >= 128;
}
inline bool wino2_s2b_ok(const Geom& g, const Tune& t) {
    if (g.s != 2 || t.wino2 == 0 || !t.wino || !pg_wino2c_geom_ok(g.N, g.Hb, g.Wb, g.Ca, g.Cb)) return false;
    if (t.wino2 == 1) return true;
    // measured on the cfg2 layers (F(3x3,2x2), shared windows): 128->64 ch -16..-18 %, 256->128 -23..-27 %, 512->256 -22 %,
    // 1024->256 -28 %, 512->128 -38 %, 256->64 -27 %; 8x8 maps and smaller: no gain (weight transform dominates)
    return pg_wino2_tiles_s2b(g.N, g.Hb, g.Wb) >= 512 && g.Cb >= 64 && g.Ca >= 128;
}
inline bool wino2_wgrad_ok(const Geom& g, const Tune& t) {
    if (g.s != 2 || t.wino2w == 0 || !t.wino || !pg_wino2_wgrad_geom_ok(g.N, g.Hs, g.Ws, g.Ca, g.Cb)) return false;
    if (t.wino2w == 1) return true;
    // measured on the cfg2 layers (64x64 output tiles unless 128x128 ones alone fill the chip): 256x128 ch -16 %, 512x256 -20 %,
    // 512x128 -17 %, 256x64 -15 %, 1024x256 +-0; 128 small-side channels: +35 % in round 1, -8 ... -13 % since the XCD-contiguous
    // 4-waves-per-SIMD form of the GEMM (enc1 184 -> 170, d1 at 2N 319 -> 277 us, before counting the V it now gets from the forward
    // call) -- taken where the K dimension is long (>= 4096 tiles)
    const long T = (long)g.N * ((g.Hs + 2) / 3) * ((g.Ws + 2) / 3);
    return g.Cb >= 64 && ((T >= 512 && g.Ca >= 256) || (T >= 4096 && g.Ca >= 128));
}
inline bool wino_wgrad_ok(const Geom& g, const Tune& t) {
    return g.s == 1 && t.winow && t.wino && pg_wino_wgrad_geom_ok(g.N, g.Hs, g.Ws, g.Ca, g.Cb);
}

// small -> big from a one-channel small (fp32) onto Cb % 4 == 0 channels, 256 % (Cb / 4) == 0: k_s2b_ca1; fp32 output under any
// MFMA algo, bf16 output (PG_IO_BIG_BF16 alone) under PG_ALGO_BF16
inline bool s2b_ca1_ok(const Geom& g, int algo_full) {
    const int a = algo_full & PG_ALGO_MASK, io = algo_full & PG_IO_MASK;
    if (g.Ca != 1 || g.Cb % 4 != 0 || g.Cb > 1024 || 256 % (g.Cb / 4) != 0 || a == PG_ALGO_DIRECT || force_generic()) return false;
    return io == 0 || (io == PG_IO_BIG_BF16 && a == PG_ALGO_BF16);
}
// PG_ALGO_BF16 with the input activation stored as bf16: the LDS-DMA kernels of conv_bf16.hip (dir 0: big -> small, 1: small -> big)
inline bool bf16x_ok(const Geom& g, int dir, int algo_full, const Tune& t) {
    if ((algo_full & PG_ALGO_MASK) != PG_ALGO_BF16 || !t.bf16x || force_generic()) return false;
    if (!(algo_full & (dir == 0 ? PG_IO_BIG_BF16 : PG_IO_SMALL_BF16))) return false;
    // big -> small from a few-channel `big`: the 8-channel-pixel form (dir 2; the caller also checks ld_big == 8)
    if (dir == 0 && g.Cb <= 8) return pg_bf16x_geom_ok(2, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s);
    return pg_bf16x_geom_ok(dir, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s);
}
// small -> big onto a few-channel `big` from a bf16 `small`: row GEMM on the bf16 kernels (dir 3) + k_col2im_small2big; fp32 output
inline bool bf16x_s2b_tapn_ok(const Geom& g, int algo_full, const Tune& t) {
    if ((algo_full & PG_ALGO_MASK) != PG_ALGO_BF16 || !t.bf16x || force_generic()) return false;
    if ((algo_full & PG_IO_MASK) != PG_IO_SMALL_BF16 || g.Cb > 8) return false;
    return pg_bf16x_geom_ok(3, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, 16 * g.Cb, g.s);
}
inline bool s2b_tapnf_bf_ok(const Geom& g, int algo_full, const Tune& t) {
    if ((algo_full & PG_ALGO_MASK) != PG_ALGO_BF16 || !t.bf16x || (algo_full & PG_IO_MASK) != PG_IO_SMALL_BF16) return false;
    return s2b_tapnf_ok(g);
}
static int launch_tapnf(bool bf, const void* small, int ld_small, const float* P, const float* bias, float* big, int ld_big, const Geom& g,
                        int act, long small_bytes, long big_bytes, hipStream_t st) {
    const int nbh = (g.Hb + 2 * TF_H - 2) / (2 * TF_H - 2), nbw = (g.Wb + 2 * TF_W - 2) / (2 * TF_W - 2);
    const long nb = (long)g.N * nbh * nbw;
    if (nb >= 0x7fffffffL) return PG_EINVAL;
    // workgroups per CU by LDS and registers: 2 for 128 fp32 input channels, else 3
    static const int wgs_env = pg_exp_env("PATCHGAN_TAPNF_WG") ? atoi(pg_exp_env("PATCHGAN_TAPNF_WG")) : 0;
    const int wgs = wgs_env > 0 ? wgs_env : (g.Ca == 128 && !bf) ? 512 : 768;
    const dim3 grid((unsigned)std::min<long>(nb, wgs));
    const int v4 = (g.Cb % 4 == 0 && ld_big % 4 == 0 && aligned16(big)) ? 1 : 0;
    const int sb = (int)small_bytes, bb = (int)big_bytes;
    TimedLaunch timed(st);
#define PG_TAPNF(CBv, KKv, BFv)                                                                                                     \
    hipLaunchKernelGGL((k_s2b_tapnf<CBv, KKv, BFv>), grid, dim3(256), 0, st, small, ld_small, P, bias, big, ld_big, g, act, sb, bb, nbh, nbw, \
                       (int)nb, v4, g.Cb, b0)
#define PG_TAPNF_K(CBv)                                                                                                             \
    if (bf) {                                                                                                                       \
        switch (g.Ca) {                                                                                                             \
            case 32: PG_TAPNF(CBv, 1, true); break;                                                                                 \
            case 64: PG_TAPNF(CBv, 2, true); break;                                                                                 \
            default: PG_TAPNF(CBv, 4, true); break;                                                                                 \
        }                                                                                                                           \
    } else {                                                                                                                        \
        switch (g.Ca) {                                                                                                             \
            case 32: PG_TAPNF(CBv, 2, false); break;                                                                                \
            case 64: PG_TAPNF(CBv, 4, false); break;                                                                                \
            default: PG_TAPNF(CBv, 8, false); break;                                                                                \
        }                                                                                                                           \
    }
    for (int b0 = 0; b0 < g.Cb; b0 += 4) {
        switch (std::min(4, g.Cb - b0)) {
            case 1: PG_TAPNF_K(1) break;
            case 2: PG_TAPNF_K(2) break;
            case 3: PG_TAPNF_K(3) break;
            default: PG_TAPNF_K(4) break;
        }
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    }
#undef PG_TAPNF_K
#undef PG_TAPNF
    return PG_OK;
}
inline size_t bf16x_s2b_tapn_ws(const Geom& g) {
    return pg_bf16x_w_bytes(g.Ca, g.Cb) + (size_t)g.N * g.Hs * g.Ws * 16 * g.Cb * sizeof(float);
}
// packed bf16 weights (unless the caller owns them) followed by the split-K slabs
inline size_t bf16x_ws(const Geom& g, int dir) {
    if (dir == 0 && g.Cb <= 8) dir = 2;
    const pg_bf16x_plan p = pg_bf16x_plan_of(dir, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s, -1);
    return pg_bf16x_w_bytes(g.Ca, g.Cb) + (p.split > 1 ? (size_t)p.split * p.out_elems * sizeof(float) : 0);
}
inline bool bf16x_wgrad_ok(const Geom& g, int algo_full, const Tune& t) {
    if ((algo_full & PG_ALGO_MASK) != PG_ALGO_BF16 || !t.bf16x || force_generic() || (algo_full & PG_IO_MASK) != PG_IO_MASK) return false;
    return pg_bf16x_wgrad_geom_ok(g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s);
}
inline bool aligned_bf_view(const void* p, int ld, bool bf) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld % (bf ? 8 : 4)) == 0;
}

// the whole bf16x call: pack (or reuse) the weights, main kernel, split-K reduce.  Returns PG_EINVAL + 1000 when the call is not
// eligible after all (alignment, workspace): the caller then falls through to the register-staged kernels.
constexpr int BF16X_SKIP = -1000;
int bf16x_run(int dir, const void* in, int ld_in, const float* P, const float* bias, void* out, int ld_out, const Geom& g, int act,
              bool out_bf, void* ws, size_t ws_bytes, hipStream_t st, const pg_conv_extras& x, int ring) {
    const pg_epi_mul mul{x.mul_t, x.mul_ld, x.mul_act};
    if (mul.t && (dir != 1 || (reinterpret_cast<uintptr_t>(mul.t) & 15) || mul.ld % (out_bf ? 8 : 4))) return PG_EINVAL;
    if (x.part && (mul.t || bias || act != PG_ACT_NONE)) return PG_EINVAL;
    if (dir == 0 && g.Cb <= 8) {       // few-channel big: only in 8-channel pixels (16 bytes = one DMA piece per pixel)
        if (ld_in != 8) return BF16X_SKIP;
        dir = 2;
    }
    const int Cin = dir == 0 ? g.Cb : dir == 2 ? 8 : g.Ca, Cout = dir == 1 ? g.Cb : g.Ca;
    const long in_pix = (long)g.N * (dir != 1 ? g.Hb * g.Wb : g.Hs * g.Ws), out_pix = (long)g.N * (dir != 1 ? g.Hs * g.Ws : g.Hb * g.Wb);
    const long in_bytes = tensor_bytes(in_pix, ld_in, Cin, true);
    if (!aligned_bf_view(in, ld_in, true) || !aligned_bf_view(out, ld_out, out_bf) || !aligned16(P) || (bias && !aligned16(bias)) ||
        in_bytes >= FAST_LIMIT)
        return BF16X_SKIP;
    const size_t wb = pg_bf16x_w_bytes(g.Ca, g.Cb);
    void* W = x.u_cache;
    char* rest = (char*)ws;
    size_t avail = ws_bytes;
    if (!W) {
        if (!ws || !aligned16(ws) || ws_bytes < wb) return BF16X_SKIP;
        W = ws;
        rest += wb;
        avail -= wb;
    }
    // small -> big reads the SAME packed copy as big -> small (transposed staging in the kernel): one pack per layer serves both
    // (not the ring-staged variant, PG_TUNE_BF16X_RING: it keeps the per-tap transposed pack; callers that share one cache entry
    // between the two directions of a layer must not set that bit -- engine._ucache checks it)
    pg_bf16x_plan p = pg_bf16x_plan_of(dir, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s, ring);
    // (the window-staged kernel reads weight fragments straight from global memory: K contiguous per output channel = the per-tap
    //  transposed pack for small -> big; pg_conv_prep_batch and pg_bf16x_own_pack() follow the same rule)
    const int bt = (dir == 1 && !(ring > 0) && !p.win) ? 1 : 0;
    if (!(x.u_cache && x.u_valid)) {
        int rc = pg_bf16x_pack(P, W, g.Ca, g.Cb, p.win ? 4 + dir : bt ? 0 : dir, st);
        if (rc != PG_OK) return rc;
    }
    pg_bf16x_clamp(&p, avail);
    const int chunks = x.part ? pg_bf16x_stats_chunks(dir, &p, g.N, g.Hb, g.Wb, g.Hs, g.Ws) : 0;
    if (x.part && (!chunks || !out_bf)) return PG_EINVAL;
    int rc;
    {
        TimedLaunch timed(st);
        if (p.split == 1)
            rc = pg_bf16x_conv(dir, in, ld_in, in_bytes, W, out, ld_out, 0L, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s, &p, bias, act,
                               out_bf ? 1 : 0, st, mul, x.part, chunks, bt);
        else
            rc = pg_bf16x_conv(dir, in, ld_in, in_bytes, W, rest, Cout, p.out_elems, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s, &p,
                               nullptr, 0, 0, st, pg_epi_mul{nullptr, 0, 0}, nullptr, 0, bt);
    }
    if (rc != PG_OK || p.split == 1) return rc;
    return launch_reduce((const float*)rest, p.out_elems, p.split, (float*)out, ld_out, out_pix, Cout, bias, act, st, out_bf ? 1 : 0, mul);
}

}  // namespace

extern "C" {

size_t pg_conv_max_tensor_bytes(void) { return (size_t)FAST_LIMIT; }

size_t pg_conv_workspace_bytes(const pg_conv_geom* g, int op) {
    if (!geom_ok(g)) return 0;
    if (op == 3) {      // pg_conv4x4_bwd_big: its two halves back to back, or V shared + the larger of the two remainders
        size_t bytes = std::max(pg_conv_workspace_bytes(g, 0), pg_conv_workspace_bytes(g, 2));
        const Geom gq = to_geom(g);
        const Tune tw = tune_widest(0);
        if (wino2_b2s_ok(gq, tw) && wino2_wgrad_ok(gq, tw))
            bytes = std::max(bytes, pg_wino2_v_bytes(gq.N, gq.Hs, gq.Ws, gq.Cb) +
                                        std::max(pg_wino2_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb),
                                                 pg_wino2_wgrad_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb)));
        return (bytes + 255) & ~(size_t)255;
    }
    Plan p = (op == 0) ? plan_b2s(g) : (op == 1) ? plan_s2b(g) : plan_wgrad(g);
    size_t bytes = 0;
    int split = p.split;
    if (op == 2 && split < DIRECT_WGRAD_SLICES) split = DIRECT_WGRAD_SLICES;   // the direct algo's slices
    if (split > 1) bytes = (size_t)split * p.out_elems * sizeof(float);
    if (op == 2) bytes += ((size_t)COLSUM_CHUNKS * g->Ca * sizeof(float) + 255) & ~(size_t)255;
    const Geom gq = to_geom(g);
    if (op == 0 && b2s_tapn_ok(gq)) bytes = std::max(bytes, b2s_tapn_ws(gq));
    if (op == 1 && s2b_tapn_ok(gq)) bytes = std::max(bytes, s2b_tapn_ws(gq) + 256);
    // enough for whichever Winograd path a PG_TUNE_* combination selects
    for (int mo1 = 2; mo1 <= 3; ++mo1) {
        const Tune t = tune_widest(mo1);
        if (op == 0 && wino_b2s_ok(gq, t)) bytes = std::max(bytes, pg_wino_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Cb, gq.Ca, mo1));
        if (op == 1 && wino_s2b_ok(gq, t)) bytes = std::max(bytes, pg_wino_ws_bytes(gq.N, gq.Hb, gq.Wb, gq.Ca, gq.Cb, mo1));
    }
    if ((op == 0 || op == 1) && pg_bf16x_geom_ok((op == 0 && gq.Cb <= 8) ? 2 : op, gq.N, gq.Hb, gq.Wb, gq.Hs, gq.Ws, gq.Ca, gq.Cb, gq.s))
        bytes = std::max(bytes, bf16x_ws(gq, op));
    if (op == 1 && gq.Cb <= 8 && pg_bf16x_geom_ok(3, gq.N, gq.Hb, gq.Wb, gq.Hs, gq.Ws, gq.Ca, 16 * gq.Cb, gq.s)) bytes = std::max(bytes, bf16x_s2b_tapn_ws(gq) + 256);
    if (op == 2 && pg_bf16x_wgrad_geom_ok(gq.N, gq.Hb, gq.Wb, gq.Hs, gq.Ws, gq.Ca, gq.Cb, gq.s)) {
        const pg_bf16x_plan wp = pg_bf16x_wgrad_plan(gq.N, gq.Hb, gq.Wb, gq.Hs, gq.Ws, gq.Ca, gq.Cb, gq.s);
        if (wp.split > 1)
            bytes = std::max(bytes, (size_t)wp.split * wp.out_elems * sizeof(float) + (((size_t)COLSUM_CHUNKS * g->Ca * sizeof(float) + 255) & ~(size_t)255));
    }
    const Tune tw = tune_widest(0);
    const size_t colsum = ((size_t)COLSUM_CHUNKS * g->Ca * sizeof(float) + 255) & ~(size_t)255;
    if (op == 0 && wino2_b2s_ok(gq, tw)) bytes = std::max(bytes, pg_wino2_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb));
    if (op == 1 && wino2_s2b_ok(gq, tw)) bytes = std::max(bytes, pg_wino2c_ws_bytes(gq.N, gq.Hb, gq.Wb, gq.Ca, gq.Cb));
    if (op == 2 && wino_wgrad_ok(gq, tw)) bytes = std::max(bytes, pg_wino_wgrad_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb) + colsum);
    if (op == 2 && wino2_wgrad_ok(gq, tw)) bytes = std::max(bytes, pg_wino2_wgrad_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb) + colsum);
    return (bytes + 255) & ~(size_t)255;
}

int pg_conv_time_next(void* ev_start, void* ev_stop) {
    t_ev0 = (hipEvent_t)ev_start;
    t_ev1 = (hipEvent_t)ev_stop;
    t_ev2 = t_ev3 = nullptr;
    return PG_OK;
}

int pg_conv_time_next2(void* ev_start, void* ev_stop, void* ev_start2, void* ev_stop2) {
    t_ev0 = (hipEvent_t)ev_start;
    t_ev1 = (hipEvent_t)ev_stop;
    t_ev2 = (hipEvent_t)ev_start2;
    t_ev3 = (hipEvent_t)ev_stop2;
    return PG_OK;
}

int pg_conv_describe(const pg_conv_geom* g, int op, size_t ws_bytes, int* tile_id, int* split, long* workgroups) {
    const int algo_full = op >> 4;      // op = opcode + 16 * (PG_ALGO_* | PG_TUNE_*): the Winograd codes are reported for PG_ALGO_AUTO only
    const int algo = algo_full & PG_ALGO_MASK;
    const Tune tune = tune_of(algo_full);
    op &= 15;
    if (!geom_ok(g) || op < 0 || op > 2 || algo < PG_ALGO_AUTO || algo > PG_ALGO_BF16) return PG_EINVAL;
    Plan p = (op == 0) ? plan_b2s(g) : (op == 1) ? plan_s2b(g) : plan_wgrad(g);
    // +100: the fast (buffer-load) variant would run for 16-byte-aligned contiguous tensors; +200: its power-of-two
    // pixel-decode instantiation (wgrad only)
    int fastcode = 0;
    if (!force_generic()) {
        const bool p2 = ((g->Hs & (g->Hs - 1)) == 0) && ((g->Ws & (g->Ws - 1)) == 0);
        if (op == 0 && g->Cb % 4 == 0 && g->Cb >= KC) fastcode = 100;
        if (op == 1 && g->Ca % 4 == 0 && g->Ca >= KC) fastcode = 100;
        if (op == 2 && wgrad_mode(g) == 0 && g->Ca % 4 == 0 && g->Cb % 4 == 0 && (p2 || (g->Ws >= 16 && g->Hs >= 2)))
            fastcode = p2 ? 200 : 100;
    }
    size_t reserved = (op == 2) ? (((size_t)COLSUM_CHUNKS * g->Ca * sizeof(float) + 255) & ~(size_t)255) : 0;
    clamp_split(p, ws_bytes, reserved);
    if (tile_id) *tile_id = p.t.id + ((op == 2) ? 10 * wgrad_mode(g) : 0) + fastcode;
    const Geom gq = to_geom(g);
    // 1020 + tile: the LDS-DMA bf16 weight-gradient kernel (k_wgrad_bf16x), both operands bf16 tensors
    if (op == 2 && bf16x_wgrad_ok(gq, algo_full, tune)) {
        pg_bf16x_plan wp = pg_bf16x_wgrad_plan(gq.N, gq.Hb, gq.Wb, gq.Hs, gq.Ws, gq.Ca, gq.Cb, gq.s);
        pg_bf16x_clamp(&wp, ws_bytes);
        if (tile_id) *tile_id = 1020 + wp.tile;
        if (split) *split = wp.split;
        if (workgroups) *workgroups = (long)wp.tiles_m * wp.tiles_n * 16 * wp.split;
        return PG_OK;
    }
    // 1000 + 10 * dir + tile: the LDS-DMA bf16 kernels (k_conv_bf16x) on bf16 tensors
    if (op == 1 && s2b_ca1_ok(gq, algo_full)) {      // 1050: k_s2b_ca1
        if (tile_id) *tile_id = 1050;
        if (split) *split = 1;
        if (workgroups) *workgroups = 2048;
        return PG_OK;
    }
    if (op == 1 && s2b_tapnf_bf_ok(gq, algo_full, tune)) {      // 1070 + Cb: k_s2b_tapnf<Cb, bf16>
        if (tile_id) *tile_id = 1070 + gq.Cb;
        if (split) *split = 1;
        const long nb = (long)gq.N * ((gq.Hb + 2 * TF_H - 2) / (2 * TF_H - 2)) * ((gq.Wb + 2 * TF_W - 2) / (2 * TF_W - 2));
        if (workgroups) *workgroups = std::min<long>(nb, 768);
        return PG_OK;
    }
    if (op == 1 && bf16x_s2b_tapn_ok(gq, algo_full, tune) && ws_bytes >= bf16x_s2b_tapn_ws(gq)) {
        const pg_bf16x_plan bp = pg_bf16x_plan_of(3, gq.N, gq.Hb, gq.Wb, gq.Hs, gq.Ws, gq.Ca, 16 * gq.Cb, gq.s, 0);
        if (tile_id) *tile_id = 1030 + bp.tile;
        if (split) *split = 1;
        if (workgroups) *workgroups = (long)bp.tiles_m * bp.tiles_n;
        return PG_OK;
    }
    if ((op == 0 || op == 1) && bf16x_ok(gq, op, algo_full, tune) && ws_bytes >= pg_bf16x_w_bytes(g->Ca, g->Cb)) {
        const int xdir = (op == 0 && gq.Cb <= 8) ? 2 : op;
        pg_bf16x_plan bp = pg_bf16x_plan_of(xdir, gq.N, gq.Hb, gq.Wb, gq.Hs, gq.Ws, gq.Ca, gq.Cb, gq.s, tune.bf16ring);
        pg_bf16x_clamp(&bp, ws_bytes - pg_bf16x_w_bytes(g->Ca, g->Cb));
        if (tile_id) *tile_id = (xdir == 2) ? 1040 + bp.tile : 1000 + 100 * (bp.win ? 2 : bp.ring) + 10 * op + bp.tile;   // (1200 + ..: k_conv_bf16r)
        if (split) *split = bp.split;
        if (workgroups) *workgroups = (long)bp.tiles_m * bp.tiles_n * bp.ncls * bp.split;
        return PG_OK;
    }
    if (op == 1 && !(algo_full & PG_IO_MASK) && s2b_tapnf_ok(gq)) {      // 1060 + Cb: k_s2b_tapnf<Cb>
        if (tile_id) *tile_id = 1060 + gq.Cb;
        if (split) *split = 1;
        const long nb = (long)gq.N * ((gq.Hb + 2 * TF_H - 2) / (2 * TF_H - 2)) * ((gq.Wb + 2 * TF_W - 2) / (2 * TF_W - 2));
        if (workgroups) *workgroups = std::min<long>(nb, (gq.Ca == 128) ? 512 : 768);
        return PG_OK;
    }
    if ((op == 0 && b2s_tapn_ok(gq) && ws_bytes >= b2s_tapn_ws(gq)) || (op == 1 && s2b_tapn_ok(gq) && ws_bytes >= s2b_tapn_ws(gq))) {
        const long M1 = (op == 0) ? (long)g->N * g->Hb * g->Wb : (long)g->N * g->Hs * g->Ws;
        const int Nc = 16 * ((op == 0) ? g->Ca : g->Cb);
        Tile t = pick_tile(M1, Nc);
        if (tile_id) *tile_id = t.id + 30;
        if (split) *split = 1;
        if (workgroups) *workgroups = ((M1 + t.bm - 1) / t.bm) * ((Nc + t.bn - 1) / t.bn);
        return PG_OK;
    }
    // +40 / +50: under PG_ALGO_AUTO this stride-1 layer runs Winograd F(2x2, 4x4) (k_wino_gemm<2,1,2,2,2> / <1,1,2,2,4>, no
    // split-K); the tile / split
    // reported are those of the implicit-GEMM kernel the other algos use
    if (algo == PG_ALGO_AUTO &&
        ((op == 0 && wino_b2s_ok(gq, tune) && ws_bytes >= pg_wino_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Cb, gq.Ca, tune.mo1)) ||
         (op == 1 && wino_s2b_ok(gq, tune) && ws_bytes >= pg_wino_ws_bytes(gq.N, gq.Hb, gq.Wb, gq.Ca, gq.Cb, tune.mo1)))) {
        const bool st = (op == 0) ? pg_wino_small_tile(gq.N, gq.Hs, gq.Ws, gq.Cb, gq.Ca, tune.mo1) : pg_wino_small_tile(gq.N, gq.Hb, gq.Wb, gq.Ca, gq.Cb, tune.mo1);
        const int mo1 = (op == 0) ? pg_wino_mo(gq.N, gq.Hs, gq.Ws, gq.Cb, gq.Ca, tune.mo1) : pg_wino_mo(gq.N, gq.Hb, gq.Wb, gq.Ca, gq.Cb, tune.mo1);
        if (tile_id) *tile_id += (mo1 == 3) ? 90 : (st ? 50 : 40);     // +90: F(3x3,4x4) variant k_wino_gemm<1,1,2,2,2,3>
    }
    // 81..83: one-shot k_b2s_tapk<Cb> for 1..3 big-side channels
    if (op == 0 && g->Cb <= 5 && !(algo_full & PG_IO_MASK) && !force_generic() && tapk_enabled() &&
        !(b2s_tapn_ok(gq) && ws_bytes >= b2s_tapn_ws(gq))) {
        if (tile_id) *tile_id = 80 + g->Cb;
        if (split) *split = 1;
        const int tmk = g->Cb <= 4 ? 128 : 64;
        if (workgroups) *workgroups = (((long)g->N * g->Hs * g->Ws + tmk - 1) / tmk) * ((g->Ca + 63) / 64);
        return PG_OK;
    }
    // 70 / 71: polyphase Winograd of a stride-2 layer (k_wino_bgemm_s3<2,2,2,2,2> / <1,2,2,2,3>; under PG_TUNE_S3_OFF k_wino_bgemm<2,2,2,2> /
    // <1,2,2,2>, 72 / 73: k_wino_bgemm_mz)
    if (algo == PG_ALGO_AUTO && op == 0 && wino2_b2s_ok(gq, tune) && ws_bytes >= pg_wino2_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb)) {
        const long T = pg_wino2_tiles_b2s(g->N, g->Hs, g->Ws), X = (long)(pg_wino2_mo() + 1) * (pg_wino2_mo() + 1);
        if (tile_id) *tile_id = (T >= 1024 ? 70 : 71) + ((!tune.s3 && pg_wino2_b2s_zb(g->N, g->Hs, g->Ws, g->Ca) > 1) ? 2 : 0);
        if (split) *split = 1;
        if (workgroups) *workgroups = X * ((T + (T >= 1024 ? 127 : 63)) / (T >= 1024 ? 128 : 64)) * ((g->Ca + 127) / 128);
        return PG_OK;
    }
    if (algo == PG_ALGO_AUTO && op == 1 && wino2_s2b_ok(gq, tune) && ws_bytes >= pg_wino2c_ws_bytes(gq.N, gq.Hb, gq.Wb, gq.Ca, gq.Cb)) {
        const long T = pg_wino2_tiles_s2b(g->N, g->Hb, g->Wb), X = (long)(pg_wino2_mo() + 1) * (pg_wino2_mo() + 1);
        if (tile_id) *tile_id = (T >= 1024 ? 70 : 71) + ((!tune.s3 && pg_wino2_s2b_zb(g->N, g->Hb, g->Wb, g->Cb) > 1) ? 2 : 0);
        if (split) *split = 1;
        if (workgroups) *workgroups = X * ((T + (T >= 1024 ? 127 : 63)) / (T >= 1024 ? 128 : 64)) * ((4 * g->Cb + 127) / 128);
        return PG_OK;
    }
    // 61 / 62: polyphase F(2x2, 3x3) weight gradient of a stride-2 layer (k_wino_wgrad_gemm<2,2,2,2> / <1,1,2,2>)
    if (algo == PG_ALGO_AUTO && op == 2 && wino2_wgrad_ok(gq, tune) &&
        ws_bytes >= reserved + pg_wino2_wgrad_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb)) {
        const int sl = pg_wino2_wgrad_slices(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb, tune.s3w);
        if (tile_id) *tile_id = pg_wino2_wgrad_tile64(gq.Ca, gq.Cb, tune.s3w) ? 62 : 61;
        if (split) *split = sl;
        if (workgroups) *workgroups = 16L * ((g->Ca + 127) / 128) * ((4 * g->Cb + 127) / 128) * sl;
        return PG_OK;
    }
    // 60 / 63: Winograd F(4x4, 2x2) weight gradient (k_wino_wgrad_gemm<2,2,2,2> / <1,1,2,2>); split = its K slices
    if (algo == PG_ALGO_AUTO && op == 2 && wino_wgrad_ok(gq, tune) && ws_bytes >= reserved + pg_wino_wgrad_ws_bytes(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb)) {
        const bool t64 = pg_wino_wgrad_tile64(gq.Ca, gq.Cb, tune.s3w);
        const int tt = t64 ? 64 : 128;
        if (tile_id) *tile_id = t64 ? 63 : 60;
        if (split) *split = pg_wino_wgrad_slices(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb, tune.s3w);
        if (workgroups) *workgroups = 25L * ((g->Ca + tt - 1) / tt) * ((g->Cb + tt - 1) / tt) * pg_wino_wgrad_slices(gq.N, gq.Hs, gq.Ws, gq.Ca, gq.Cb, tune.s3w);
        return PG_OK;
    }
    if (split) *split = p.split;
    if (workgroups) *workgroups = (long)p.tiles_m * p.tiles_n * p.ncls * p.split;
    return PG_OK;
}

static int conv_kernel_impl(const pg_conv_geom* g, int op, size_t ws_bytes, char* name, size_t name_len, int* split, double* mfma_flops,
                            double* useful_flops) {
    int code = 0, sp = 1;
    long wgs = 0;
    int rc = pg_conv_describe(g, op, ws_bytes, &code, &sp, &wgs);
    if (rc != PG_OK) return rc;
    const int algo_full = op >> 4, algo = algo_full & PG_ALGO_MASK, oc = op & 15;
    const Tune tune = tune_of(algo_full);
    static const char* const TILE[5] = {"2,2,2,2", "2,1,2,2", "1,1,4,1", "1,2,2,2", "1,1,2,2"};
    const int fast = code / 100, rest = code % 100, tid = rest % 10, mode = rest / 10;
    auto cd = [](long a, long b) { return (a + b - 1) / b; };
    auto fr = [](double a, double b) { return a / b; };       // the same tile counts without the round-up to whole tiles
    const double direct = 2.0 * g->N * g->Hs * g->Ws * 16.0 * g->Ca * g->Cb;
    double fl = direct, fu = direct;      // executed (ragged tiles padded to whole ones) / useful (the algorithm's count on the exact extents)
    char buf[128];
    if (code > 1060 && code <= 1064) {
        snprintf(buf, sizeof buf, "k_s2b_tapnf<%d>", code - 1060);          // one-pass taps-in-N ConvTranspose2d onto <= 4 channels
    } else if (code > 1064 && code <= 1068) {
        snprintf(buf, sizeof buf, "k_s2b_tapnf<4>+k_s2b_tapnf<%d>", code - 1064);      // 5 .. 8 channels: two launches
    } else if (code > 1070 && code <= 1074) {
        snprintf(buf, sizeof buf, "k_s2b_tapnf<%d,bf16>", code - 1070);     // ... from a bf16 tensor
    } else if (code > 1074 && code <= 1078) {
        snprintf(buf, sizeof buf, "k_s2b_tapnf<4,bf16>+k_s2b_tapnf<%d,bf16>", code - 1074);
    } else if (code == 1050) {
        snprintf(buf, sizeof buf, (g->stride == 1 && (size_t)(g->Hs + 4) * (g->Ws + 4) * sizeof(float) <= 48 * 1024)
                                      ? "k_s2b_ca1_s1" : "k_s2b_ca1");      // (the LDS-staged form: stride 1, see s2b_impl)
    } else if (code >= 1020 && code < 1030) {
        snprintf(buf, sizeof buf, "%s", pg_bf16x_wgrad_kernel_name(code - 1020));
    } else if (code >= 1030 && code < 1040) {
        snprintf(buf, sizeof buf, "%s+k_col2im_small2big", pg_bf16x_kernel_name(3, code - 1030, 0));
    } else if (code >= 1040 && code < 1050) {
        snprintf(buf, sizeof buf, "%s", pg_bf16x_kernel_name(2, code - 1040, 0));
    } else if (code >= 1000) {
        snprintf(buf, sizeof buf, "%s", pg_bf16x_kernel_name((code / 10) % 10, code % 10, (code / 100) % 10));
    } else if (algo == PG_ALGO_DIRECT) {
        snprintf(buf, sizeof buf, "%s", oc == 0 ? "k_big2small_direct" : oc == 1 ? "k_small2big_direct" : "k_wgrad_direct");
    } else if (mode == 6) {          // Winograd weight gradients: 60 / 63 stride 1 (F(4x4,2x2)), 61 / 62 polyphase stride 2
        if (tune.s3w) snprintf(buf, sizeof buf, "k_wino_wgrad_gemm_s3<%s>", (tid == 2 || tid == 3) ? "1,1,2,2,2,3" : "2,2,2,2,1,2");
        else snprintf(buf, sizeof buf, "k_wino_wgrad_gemm<%s>", (tid == 2 || tid == 3) ? "1,1,2,2" : "2,2,2,2");
        fl = (g->stride == 1) ? pg_wino_wgrad_flops(g->N, g->Hs, g->Ws, g->Ca, g->Cb)
                              : 2.0 * 16 * g->N * cd(g->Hs, 3) * cd(g->Ws, 3) * g->Ca * 4.0 * g->Cb;
        const int wr = pg_wino_wgrad_r(g->N, g->Hs, g->Ws);
        fu = (g->stride == 1) ? 2.0 * (wr + 3) * (wr + 3) * g->N * fr(g->Hs, wr) * fr(g->Ws, wr) * g->Ca * g->Cb
                              : 2.0 * 16 * g->N * fr(g->Hs, 3) * fr(g->Ws, 3) * g->Ca * 4.0 * g->Cb;
    } else if (mode == 8) {
        snprintf(buf, sizeof buf, (tid <= 4 && tapkp_enabled()) ? "k_b2s_tapkp<%d>" : "k_b2s_tapk<%d>", tid);
        sp = 1;
    } else if (mode == 7) {          // polyphase Winograd of a stride-2 layer
        if (tune.s3)       // split-bf16 form (the default): 128-row tiles at two waves per SIMD, 64-row tiles at three
            snprintf(buf, sizeof buf, "k_wino_bgemm_s3<%s>", (tid & 1) ? "1,2,2,2,3" : "2,2,2,2,2");
        else
            snprintf(buf, sizeof buf, "k_wino_bgemm%s<%s>", tid >= 2 ? "_mz" : "", (tid & 1) ? "1,2,2,2" : "2,2,2,2");
        const int mo = pg_wino2_mo();
        fl = (oc == 0) ? 2.0 * (mo + 1) * (mo + 1) * g->N * cd(g->Hs, mo) * cd(g->Ws, mo) * 4.0 * g->Cb * g->Ca
                       : 2.0 * 4 * (mo + 1) * (mo + 1) * g->N * cd(cd(g->Hb, 2) + 1, mo) * cd(cd(g->Wb, 2) + 1, mo) * (double)g->Ca * g->Cb;
        fu = (oc == 0) ? 2.0 * (mo + 1) * (mo + 1) * g->N * fr(g->Hs, mo) * fr(g->Ws, mo) * 4.0 * g->Cb * g->Ca
                       : 2.0 * 4 * (mo + 1) * (mo + 1) * g->N * fr(fr(g->Hb, 2), mo) * fr(fr(g->Wb, 2), mo) * (double)g->Ca * g->Cb;
        sp = 1;
    } else if (mode == 9 || mode == 4 || mode == 5) {     // stride-1 Winograd forward / data gradient
        const int mo = mode == 9 ? 3 : 2;
        const int dm = tune.dma;
        if (mode == 9 && dm) snprintf(buf, sizeof buf, "k_wino_gemm_dma<3,4,2>");
        else if (mode == 5 && dm == 2) snprintf(buf, sizeof buf, "k_wino_gemm_dma<2,3,3>");
        else if (mode == 9 && pg_wino_row_on())
            snprintf(buf, sizeof buf, (tune.s3r && (oc == 0 ? g->Cb : g->Ca) % 32 == 0) ? "k_wino_gemm_row_s3<2>" : "k_wino_gemm_row<4,1>");
        else snprintf(buf, sizeof buf, "k_wino_gemm<%s>", mode == 9 ? "1,1,2,2,2,3" : mode == 4 ? "2,1,2,2,2,2" : "1,1,2,2,4,2");
        const int ho = oc == 0 ? g->Hs : g->Hb, wo = oc == 0 ? g->Ws : g->Wb;
        fl = 2.0 * (mo + 3) * (mo + 3) * g->N * cd(ho, mo) * cd(wo, mo) * (double)g->Ca * g->Cb;
        fu = 2.0 * (mo + 3) * (mo + 3) * g->N * fr(ho, mo) * fr(wo, mo) * (double)g->Ca * g->Cb;
        sp = 1;
    } else if (mode == 3) {
        snprintf(buf, sizeof buf, "k_b2s_fast<%s,true>+%s", TILE[tid], oc == 0 ? "k_gather_big2small" : "k_col2im_small2big");
    } else if (mode == 1 && oc == 2 && wgrad_tapnp_ok(to_geom(g)) && algo != PG_ALGO_BF16) {
        snprintf(buf, sizeof buf, "k_wgrad_tapnp<%d>", g->Cb);          // persistent taps-in-N weight gradient of the image-facing layers
        sp = wgrad_tapnp_slabs(to_geom(g));
    } else if (mode) {
        snprintf(buf, sizeof buf, "k_wgrad_tapn<%s,%d>", TILE[tid], mode);
    } else {
        const bool half = (algo == PG_ALGO_BF16) && fast;
        // bf16 kernels: the trailing template argument says whether the activation operand(s) are stored as bf16 (PG_IO_* bits)
        const int io = algo_full & PG_IO_MASK;
        const char* hin = (oc == 0 ? (io & PG_IO_BIG_BF16) : oc == 1 ? (io & PG_IO_SMALL_BF16) : io == PG_IO_MASK) ? "true" : "false";
        if (half && oc == 0) snprintf(buf, sizeof buf, "k_b2s_bf16<%s,false,%s>", TILE[tid], hin);
        else if (half && oc == 1) snprintf(buf, sizeof buf, "k_s2b_bf16<%s,%s>", TILE[tid], hin);
        else if (half) snprintf(buf, sizeof buf, "k_wgrad_bf16<%s,%s,%s>", TILE[tid], fast == 2 ? "true" : "false", hin);
        else if (fast && oc == 0) snprintf(buf, sizeof buf, "k_b2s_fast<%s,false>", TILE[tid]);
        else if (fast && oc == 1) snprintf(buf, sizeof buf, "k_s2b_fast<%s>", TILE[tid]);
        else if (fast) snprintf(buf, sizeof buf, "k_wgrad_fast<%s,%s>", TILE[tid], fast == 2 ? "true" : "false");
        else snprintf(buf, sizeof buf, "%s<%s>", oc == 0 ? "k_big2small" : oc == 1 ? "k_small2big" : "k_wgrad", TILE[tid]);
    }
    if (name && name_len) snprintf(name, name_len, "%s", buf);
    if (split) *split = sp;
    if (mfma_flops) *mfma_flops = fl;
    if (useful_flops) *useful_flops = fu;
    return PG_OK;
}

int pg_conv_kernel(const pg_conv_geom* g, int op, size_t ws_bytes, char* name, size_t name_len, int* split, double* mfma_flops) {
    return conv_kernel_impl(g, op, ws_bytes, name, name_len, split, mfma_flops, nullptr);
}

int pg_conv_kernel_flops(const pg_conv_geom* g, int op, size_t ws_bytes, double* executed, double* useful) {
    return conv_kernel_impl(g, op, ws_bytes, nullptr, 0, nullptr, executed, useful);
}

// the col2im half of the taps-folded-into-N small -> big paths
static int launch_col2im(const float* D, const float* bias, float* big, int ld_big, const Geom& g, int act, hipStream_t st) {
    const size_t lds = (size_t)(CT_H + 2) * (CT_W + 2) * 16 * g.Cb * sizeof(float);
    if (g.s == 2) {                    // (a D row is 16 * Cb floats: always whole float4s)
        if (lds <= 64 * 1024 && aligned16(D)) {
            const int tiles_h = (g.Hs + CT_H) / CT_H, tiles_w = (g.Ws + CT_W) / CT_W;      // big rows reach one small row past Hs - 1
            hipLaunchKernelGGL(k_col2im_s2_lds, dim3((unsigned)(g.N * tiles_h * tiles_w)), dim3(256), lds, st, D, bias, big, ld_big, g, act,
                               tiles_h, tiles_w);
            return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
        }
    }
    const long total = (long)g.N * g.Hb * g.Wb * g.Cb;
    hipLaunchKernelGGL(k_col2im_small2big, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, st, D, bias, big, ld_big, g, act);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

static const pg_conv_extras NO_EXTRAS = {nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, 0};

static int b2s_impl(const float* big, int ld_big, const float* P, const float* bias, float* small, int ld_small,
                    const pg_conv_geom* gg, int act, int algo, void* ws, size_t ws_bytes, void* stream, const pg_conv_extras* xp) {
    const pg_conv_extras& x = xp ? *xp : NO_EXTRAS;
    double* part = x.part;
    if ((x.u_cache && !aligned16(x.u_cache)) || (x.v_keep && !aligned16(x.v_keep)) || x.v_pre) return PG_EINVAL;
    if (!geom_ok(gg) || !big || !P || !small || ld_big < gg->Cb || ld_small < gg->Ca) return PG_EINVAL;
    if (act < PG_ACT_NONE || act > PG_ACT_SIGMOID) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    Geom g = to_geom(gg);
    const Tune tune = tune_of(algo);
    const int io = algo & PG_IO_MASK;                 // bf16 activation storage: PG_IO_BIG_BF16 = input, PG_IO_SMALL_BF16 = output
    algo &= PG_ALGO_MASK;
    if (io && algo != PG_ALGO_BF16) return PG_EINVAL;
    if (algo == PG_ALGO_DIRECT) {
        const long total = (long)g.N * g.Hs * g.Ws * g.Ca;
        int blocks = (int)std::min<long>((total + 255) / 256, 65536);
        hipLaunchKernelGGL(k_big2small_direct, dim3(blocks), dim3(256), 0, st, big, ld_big, P, bias, small, ld_small, g,
                           act);
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    if (!ws) ws_bytes = 0;
    if (algo == PG_ALGO_AUTO && wino_b2s_ok(g, tune) && aligned16(P) && aligned16(ws) &&
        ws_bytes >= pg_wino_ws_bytes(g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1) &&
        pg_wino_eligible(g.N, g.Hb, g.Wb, g.Cb, g.Hs, g.Ws, g.Ca, ld_big, big, tune.mo1)) {
        if (part) return PG_EINVAL;
        if (x.v_keep && !(wino_wgrad_ok(g, tune) && pg_wino_wgrad_v_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb, tune.mo1))) return PG_EINVAL;
        int rc = pg_wino_prepare(big, ld_big, P, 0, g.N, g.Hb, g.Wb, g.Cb, g.Hs, g.Ws, g.Ca, 1, ws, st, tune.mo1, x.u_cache, x.u_valid,
                                 x.v_keep);
        if (rc != PG_OK) return rc;
        const int nsl = pg_wino_gemm_rows(g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1, tune.dma, small, ld_small, bias, pg_epi_mul{nullptr, 0, 0})
                            ? 1 : pg_wino_gemm_slices(g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1);
        {
            TimedLaunch timed(st);
            rc = pg_wino_gemm(bias, small, ld_small, g.N, g.Cb, g.Hs, g.Ws, g.Ca, act, ws, st, tune.mo1, tune.dma, x.u_cache,
                              pg_epi_mul{nullptr, 0, 0}, x.v_keep, tune.s3r);
        }
        if (rc != PG_OK || nsl == 1) return rc;
        const long pix = (long)g.N * g.Hs * g.Ws;
        return launch_reduce(pg_wino_gemm_slabs(ws, g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1), pix * g.Ca, nsl, small, ld_small, pix, g.Ca, bias,
                             act, st);
    }
    if (algo == PG_ALGO_AUTO && wino2_b2s_ok(g, tune) && (ld_big % 4 == 0) && (ld_small % 4 == 0) && aligned16(big) && aligned16(P) &&
        aligned16(small) && aligned16(ws) && (!bias || aligned16(bias)) &&
        ws_bytes >= pg_wino2_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb)) {
        hipEvent_t e0 = t_ev0, e1 = t_ev1;
        t_ev0 = nullptr;
        t_ev1 = nullptr;
        if (part && pg_wino2_b2s_stats_chunks(g.N, g.Hs, g.Ws, g.Ca) == 0) return PG_EINVAL;
        if (x.v_keep && pg_wino2_mo() != 3) return PG_EINVAL;
        return pg_wino2_b2s(big, ld_big, P, bias, small, ld_small, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, act, ws, st, e0, e1, nullptr,
                            part, x.v_keep, x.u_cache, x.u_valid, tune.s3);
    }
    if (bf16x_ok(g, 0, algo | io, tune) && !x.v_keep) {
        const int rc = bf16x_run(0, big, ld_big, P, bias, small, ld_small, g, act, io & PG_IO_SMALL_BF16, ws, ws_bytes, st, x, tune.bf16ring);
        if (rc != BF16X_SKIP) return rc;
    }
    // only the Winograd paths (and, for the partial sums, the persistent image-facing kernel below) have operands to hand over
    // (the pg_conv_*_bytes / _chunks queries said 0)
    if (x.v_keep || x.u_cache) return PG_EINVAL;
    if (part && !(algo == PG_ALGO_AUTO && !io && tapk_enabled() && tapkp_enabled() && tapkp_stats_chunks(g) > 0 && !bias && act == PG_ACT_NONE))
        return PG_EINVAL;
    if (!io && b2s_tapn_ok(g) && (ld_big % 4 == 0) && aligned16(big) && aligned16(P) && aligned16(ws) &&
        ws_bytes >= b2s_tapn_ws(g) && tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb) < FAST_LIMIT) {
        // D[big pixel][(tap, a)] = big . P^T (row GEMM over the pixels), then gather the 16 taps per output pixel
        float* D = (float*)ws;
        const long Mb = (long)g.N * g.Hb * g.Wb;
        const int Nc = 16 * g.Ca;
        Geom g1{g.N, g.Hb, g.Wb, g.Hb, g.Wb, Nc, g.Cb, 1};
        Tile t = pick_tile(Mb, Nc);
        dim3 grid((unsigned)((Mb + t.bm - 1) / t.bm), (Nc + t.bn - 1) / t.bn, 1);
        {
            TimedLaunch timed(st);
            PG_DISPATCH_B2SF(true, t.id, grid, st, big, ld_big, P, D, Nc, 0L, g1, g.Cb / KC, (const float*)nullptr, 0,
                             (int)tensor_bytes(Mb, ld_big, g.Cb), (int)(16L * g.Ca * g.Cb * 4));
        }
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        const long total = (long)g.N * g.Hs * g.Ws * g.Ca;
        hipLaunchKernelGGL(k_gather_big2small, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, st, D, bias,
                           small, ld_small, g, act);
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    if (g.Cb <= 5 && !io && !force_generic() && tapk_enabled()) {     // (6..8 channels: measured slower than the generic kernel)
        // K = 16*Cb <= 48: one-shot kernel (with 4 channels the pipelined generic kernel is as fast: 46 TFLOP/s both)
        const int vec4 = (g.Cb == 4) && (ld_big % 4 == 0) && aligned16(big);
        const int vec_out = (g.Ca % 4 == 0) && (ld_small % 4 == 0) && aligned16(small) && (!bias || aligned16(bias));
        const int tmk = g.Cb <= 4 ? 128 : 64;
        dim3 grid((unsigned)(((long)g.N * g.Hs * g.Ws + tmk - 1) / tmk), (g.Ca + 63) / 64, 1);
        TimedLaunch timed(st);
        const long big_b = tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb), out_b = tensor_bytes((long)g.N * g.Hs * g.Ws, ld_small, g.Ca);
        if (g.Cb <= 4 && tapkp_enabled() && vec_out && (act == PG_ACT_NONE || act == PG_ACT_LEAKY) && big_b < FAST_LIMIT && out_b < FAST_LIMIT &&
            (g.Cb != 4 || vec4) && (long)g.N * g.Hs * g.Ws < 0x3fffffL * 64) {
            // persistent form: 2 workgroups per CU (two LDS buffers of 9 .. 35 KB, two accumulator sets), each walking tiles blockIdx.x, + gridDim.x, ...
            const int ntiles = (int)grid.x;
            static const int pwg = pg_exp_env("PATCHGAN_TAPKP_WG") ? atoi(pg_exp_env("PATCHGAN_TAPKP_WG")) : 512;
            const bool wide3 = (g.Cb == 3) && (ld_big % 4 == 0) && (ld_big >= 4) && aligned16(big);
            dim3 pgrid((unsigned)std::min<long>(ntiles, std::max<long>(1, pwg / (long)grid.y)), grid.y, 1);
            if (part) {             // (act == none, no bias, whole tiles per sample: checked above); one workgroup per CU (88 .. 104 KB of LDS)
                pgrid.x = (unsigned)std::min<long>(ntiles, std::max<long>(1, 256 / (long)grid.y));
                switch (g.Cb) {
                    case 1: hipLaunchKernelGGL((k_b2s_tapkp<1, 0, false, true>), pgrid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, (int)big_b, (int)out_b, ntiles, part); break;
                    case 2: hipLaunchKernelGGL((k_b2s_tapkp<2, 0, false, true>), pgrid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, (int)big_b, (int)out_b, ntiles, part); break;
                    case 3:
                        if (wide3) hipLaunchKernelGGL((k_b2s_tapkp<3, 0, true, true>), pgrid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, (int)big_b, (int)out_b, ntiles, part);
                        else hipLaunchKernelGGL((k_b2s_tapkp<3, 0, false, true>), pgrid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, (int)big_b, (int)out_b, ntiles, part);
                        break;
                    default: hipLaunchKernelGGL((k_b2s_tapkp<4, 0, true, true>), pgrid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, (int)big_b, (int)out_b, ntiles, part); break;
                }
                return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
            }
#define PG_TAPKP(CB_, WIDE_)                                                                                                            \
    do {                                                                                                                                \
        if (act == PG_ACT_LEAKY)                                                                                                        \
            hipLaunchKernelGGL((k_b2s_tapkp<CB_, 1, WIDE_>), pgrid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, (int)big_b, \
                               (int)out_b, ntiles);                                                                                \
        else                                                                                                                            \
            hipLaunchKernelGGL((k_b2s_tapkp<CB_, 0, WIDE_>), pgrid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, (int)big_b, \
                               (int)out_b, ntiles);                                                                                \
    } while (0)
            switch (g.Cb) {
                case 1: PG_TAPKP(1, false); break;
                case 2: PG_TAPKP(2, false); break;
                case 3: if (wide3) PG_TAPKP(3, true); else PG_TAPKP(3, false); break;
                default: PG_TAPKP(4, true); break;
            }
#undef PG_TAPKP
            return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
        }
        if (part) return PG_EINVAL;        // (a view the persistent kernel does not take: unaligned output, tensor beyond 32-bit offsets)
        switch (g.Cb) {
            case 1: hipLaunchKernelGGL(k_b2s_tapk<1>, grid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, act, vec4, vec_out); break;
            case 2: hipLaunchKernelGGL(k_b2s_tapk<2>, grid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, act, vec4, vec_out); break;
            case 3: hipLaunchKernelGGL(k_b2s_tapk<3>, grid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, act, vec4, vec_out); break;
            case 5: hipLaunchKernelGGL((k_b2s_tapk<5, 1>), grid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, act, 0, vec_out); break;
            case 6: hipLaunchKernelGGL((k_b2s_tapk<6, 1>), grid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, act, 0, vec_out); break;
            case 7: hipLaunchKernelGGL((k_b2s_tapk<7, 1>), grid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, act, 0, vec_out); break;
            case 8: hipLaunchKernelGGL((k_b2s_tapk<8, 1>), grid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, act, 0, vec_out); break;
            default: hipLaunchKernelGGL(k_b2s_tapk<4>, grid, dim3(256), 0, st, big, ld_big, P, small, ld_small, g, bias, act, vec4, vec_out); break;
        }
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    Plan p = plan_b2s(gg);
    clamp_split(p, ws_bytes, 0);
    const bool in_bf = io & PG_IO_BIG_BF16, out_bf = io & PG_IO_SMALL_BF16;
    const int veck = (g.Cb % 4 == 0) && (ld_big % 4 == 0) && aligned_io(big, in_bf) && aligned16(P);
    dim3 grid(p.tiles_m, p.tiles_n, p.split);
    const long big_bytes = tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb, in_bf), p_bytes = 16L * g.Ca * g.Cb * 4;
    const bool fast = veck && g.Cb >= KC && big_bytes < FAST_LIMIT && p_bytes < FAST_P_LIMIT && !force_generic();
    if (io && !fast) return PG_EINVAL;               // bf16 tensors only on the fast bf16 kernels
    if (p.split == 1) {
        TimedLaunch timed(st);
        if (fast && algo == PG_ALGO_BF16 && in_bf) {
            PG_DISPATCH_B2SH(true, p.t.id, grid, st, big, ld_big, P, small, ld_small, 0L, g, p.cps, bias, act, (int)big_bytes,
                             (int)p_bytes, (int)out_bf);
        } else if (fast && algo == PG_ALGO_BF16) {
            PG_DISPATCH_B2SH(false, p.t.id, grid, st, big, ld_big, P, small, ld_small, 0L, g, p.cps, bias, act, (int)big_bytes,
                             (int)p_bytes, (int)out_bf);
        } else if (fast) {
            PG_DISPATCH_B2SF(false, p.t.id, grid, st, big, ld_big, P, small, ld_small, 0L, g, p.cps, bias, act,
                             (int)big_bytes, (int)p_bytes);
        } else {
            PG_DISPATCH_TILE(k_big2small, p.t.id, grid, st, big, ld_big, P, small, ld_small, 0L, g, p.cps, veck, bias,
                             act);
        }
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    float* slabs = (float*)ws;
    {
        TimedLaunch timed(st);
        if (fast && algo == PG_ALGO_BF16 && in_bf) {
            PG_DISPATCH_B2SH(true, p.t.id, grid, st, big, ld_big, P, slabs, g.Ca, p.out_elems, g, p.cps, (const float*)nullptr, 0,
                             (int)big_bytes, (int)p_bytes, 0);
        } else if (fast && algo == PG_ALGO_BF16) {
            PG_DISPATCH_B2SH(false, p.t.id, grid, st, big, ld_big, P, slabs, g.Ca, p.out_elems, g, p.cps, (const float*)nullptr, 0,
                             (int)big_bytes, (int)p_bytes, 0);
        } else if (fast) {
            PG_DISPATCH_B2SF(false, p.t.id, grid, st, big, ld_big, P, slabs, g.Ca, p.out_elems, g, p.cps,
                             (const float*)nullptr, 0, (int)big_bytes, (int)p_bytes);
        } else {
            PG_DISPATCH_TILE(k_big2small, p.t.id, grid, st, big, ld_big, P, slabs, g.Ca, p.out_elems, g, p.cps, veck,
                             (const float*)nullptr, 0);
        }
    }
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    return launch_reduce(slabs, p.out_elems, p.split, small, ld_small, (long)g.N * g.Hs * g.Ws, g.Ca, bias, act, st, out_bf);
}

int pg_conv4x4_big2small(const float* big, int ld_big, const float* P, const float* bias, float* small,
                         int ld_small, const pg_conv_geom* gg, int act, int algo, void* ws, size_t ws_bytes,
                         void* stream) {
    return b2s_impl(big, ld_big, P, bias, small, ld_small, gg, act, algo, ws, ws_bytes, stream, nullptr);
}

int pg_conv4x4_big2small_x(const float* big, int ld_big, const float* P, const float* bias, float* small,
                           int ld_small, const pg_conv_geom* gg, int act, int algo, void* ws, size_t ws_bytes,
                           void* stream, const pg_conv_extras* x) {
    return b2s_impl(big, ld_big, P, bias, small, ld_small, gg, act, algo, ws, ws_bytes, stream, x);
}

static int s2b_impl(const float* small, int ld_small, const float* P, const float* bias, float* big, int ld_big,
                    const pg_conv_geom* gg, int act, int algo, void* ws, size_t ws_bytes, void* stream, const pg_conv_extras* xp) {
    const pg_conv_extras& x = xp ? *xp : NO_EXTRAS;
    double* part = x.part;
    if ((x.u_cache && !aligned16(x.u_cache)) || x.v_keep || x.v_pre) return PG_EINVAL;
    const pg_epi_mul mul{x.mul_t, x.mul_ld, x.mul_act};
    if (mul.t && (mul.ld < gg->Cb || mul.act < PG_ACT_NONE || mul.act > PG_ACT_SIGMOID || part)) return PG_EINVAL;
    if (!geom_ok(gg) || !big || !P || !small || ld_big < gg->Cb || ld_small < gg->Ca) return PG_EINVAL;
    if (act < PG_ACT_NONE || act > PG_ACT_SIGMOID) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    Geom g = to_geom(gg);
    const Tune tune = tune_of(algo);
    const int io = algo & PG_IO_MASK;                 // PG_IO_SMALL_BF16 = input, PG_IO_BIG_BF16 = output
    algo &= PG_ALGO_MASK;
    if (io && algo != PG_ALGO_BF16) return PG_EINVAL;
    if (algo == PG_ALGO_DIRECT) {
        if (mul.t) return PG_EINVAL;
        const long total = (long)g.N * g.Hb * g.Wb * g.Cb;
        int blocks = (int)std::min<long>((total + 255) / 256, 65536);
        hipLaunchKernelGGL(k_small2big_direct, dim3(blocks), dim3(256), 0, st, small, ld_small, P, bias, big, ld_big, g,
                           act);
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    if (!ws) ws_bytes = 0;
    if (s2b_ca1_ok(g, algo | io)) {
        const bool out_bf = io & PG_IO_BIG_BF16;
        const bool al = aligned16(P) && (!bias || aligned16(bias)) && aligned_bf_view(big, ld_big, out_bf) &&
                        (!mul.t || aligned_bf_view(mul.t, mul.ld, out_bf));
        if (al && !part && !x.u_cache) {
            const long total = (long)g.N * g.Hb * g.Wb;
            // pixels per workgroup: ~1500 workgroups, whole trips of 2 pixels per thread row (256 / (Cb / 4) pixel rows per workgroup)
            const int trip = 2 * (256 / (g.Cb / 4));
            static const int ca1_wgs = pg_exp_env("PATCHGAN_CA1_WGS") ? atoi(pg_exp_env("PATCHGAN_CA1_WGS")) : 1536;
            const int ppb = (int)std::max<long>(trip, ((total / ca1_wgs + trip - 1) / trip) * trip);
            TimedLaunch timed(st);
            const size_t xs_bytes = (size_t)(g.Hs + 4) * (g.Ws + 4) * sizeof(float);
            static const bool no_s1 = pg_exp_env("PATCHGAN_NO_CA1S1") != nullptr;
            if (!no_s1 && g.s == 1 && xs_bytes <= 48 * 1024 && g.Hb == g.Hs + 1 && g.Wb == g.Ws + 1) {
                // per-sample form: ~1536 workgroups in all, whole pixel rows of the workgroup (256 / (Cb / 4) pixels per trip)
                const int npr = 256 / (g.Cb / 4), hwb = g.Hb * g.Wb;
                const long per = std::max<long>(1, 1536 / g.N);
                static const int minpix = pg_exp_env("PATCHGAN_CA1S1_MINPIX") ? atoi(pg_exp_env("PATCHGAN_CA1S1_MINPIX")) : 0;
                const int ppb1 = std::max(minpix, (int)(((hwb + per - 1) / per + npr - 1) / npr * npr));
                hipLaunchKernelGGL(k_s2b_ca1_s1, dim3((unsigned)((hwb + ppb1 - 1) / ppb1), g.N), dim3(256), xs_bytes, st, small, ld_small, P, bias,
                                   big, ld_big, g, act, out_bf ? 1 : 0, mul, ppb1);
                return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
            }
            hipLaunchKernelGGL(k_s2b_ca1, dim3((unsigned)((total + ppb - 1) / ppb)), dim3(256), 0, st, small, ld_small, P, bias, big, ld_big, g,
                               act, out_bf ? 1 : 0, mul, ppb);
            return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
        }
    }
    if (algo == PG_ALGO_AUTO && wino_s2b_ok(g, tune) && aligned16(P) && aligned16(ws) &&
        ws_bytes >= pg_wino_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1) &&
        pg_wino_eligible(g.N, g.Hs, g.Ws, g.Ca, g.Hb, g.Wb, g.Cb, ld_small, small, tune.mo1)) {
        if (part) return PG_EINVAL;
        if (mul.t && (!aligned16(mul.t) || mul.ld % 4)) return PG_EINVAL;
        int rc = pg_wino_prepare(small, ld_small, P, 1, g.N, g.Hs, g.Ws, g.Ca, g.Hb, g.Wb, g.Cb, 2, ws, st, tune.mo1, x.u_cache, x.u_valid);
        if (rc != PG_OK) return rc;
        const int nsl = pg_wino_gemm_rows(g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1, tune.dma, big, ld_big, bias, mul)
                            ? 1 : pg_wino_gemm_slices(g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1);
        {
            TimedLaunch timed(st);
            rc = pg_wino_gemm(bias, big, ld_big, g.N, g.Ca, g.Hb, g.Wb, g.Cb, act, ws, st, tune.mo1, tune.dma, x.u_cache, mul, nullptr, tune.s3r);
        }
        if (rc != PG_OK || nsl == 1) return rc;
        const long pix = (long)g.N * g.Hb * g.Wb;
        return launch_reduce(pg_wino_gemm_slabs(ws, g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1), pix * g.Cb, nsl, big, ld_big, pix, g.Cb, bias, act,
                             st, 0, mul);
    }
    if (algo == PG_ALGO_AUTO && wino2_s2b_ok(g, tune) && (ld_big % 4 == 0) && (ld_small % 4 == 0) && aligned16(big) && aligned16(P) &&
        aligned16(small) && aligned16(ws) && (!bias || aligned16(bias)) &&
        ws_bytes >= pg_wino2c_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb)) {
        hipEvent_t e0 = t_ev0, e1 = t_ev1;
        t_ev0 = nullptr;
        t_ev1 = nullptr;
        if (part && pg_wino2_s2b_stats_chunks(g.N, g.Hb, g.Wb, g.Cb) == 0) return PG_EINVAL;
        if (mul.t && (!aligned16(mul.t) || mul.ld % 4)) return PG_EINVAL;
        return pg_wino2_s2b(small, ld_small, P, bias, big, ld_big, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, act, ws, st, e0, e1, part,
                            x.u_cache, x.u_valid, mul, tune.s3);
    }
    if (s2b_tapnf_bf_ok(g, algo | io, tune) && !part && !x.u_cache && !mul.t && (ld_small % 8 == 0) && aligned16(small) &&
        tensor_bytes((long)g.N * g.Hs * g.Ws, ld_small, g.Ca, true) < FAST_LIMIT && tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb) < FAST_LIMIT)
        return launch_tapnf(true, small, ld_small, P, bias, big, ld_big, g, act, tensor_bytes((long)g.N * g.Hs * g.Ws, ld_small, g.Ca, true),
                            tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb), st);
    if (bf16x_s2b_tapn_ok(g, algo | io, tune) && !part && !x.u_cache && !mul.t && ws && aligned16(ws) && ws_bytes >= bf16x_s2b_tapn_ws(g) &&
        aligned_bf_view(small, ld_small, true) && aligned16(P) &&
        tensor_bytes((long)g.N * g.Hs * g.Ws, ld_small, g.Ca, true) < FAST_LIMIT) {
        // D[small pixel][(tap, b)] = small . W' (bf16 row GEMM), then col2im: each big pixel sums the taps that reach it
        const int Nc = 16 * g.Cb;
        const size_t wb = pg_bf16x_w_bytes(g.Ca, g.Cb);
        float* D = (float*)((char*)ws + wb);
        int rc = pg_bf16x_pack(P, ws, g.Ca, g.Cb, 1, st);
        if (rc != PG_OK) return rc;
        const pg_bf16x_plan bp = pg_bf16x_plan_of(3, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, Nc, g.s, 0);
        {
            TimedLaunch timed(st);
            rc = pg_bf16x_conv(3, small, ld_small, tensor_bytes((long)g.N * g.Hs * g.Ws, ld_small, g.Ca, true), ws, D, Nc, 0L, g.N, g.Hb,
                               g.Wb, g.Hs, g.Ws, g.Ca, Nc, g.s, &bp, nullptr, 0, 0, st);
        }
        if (rc != PG_OK) return rc;
        return launch_col2im(D, bias, big, ld_big, g, act, st);
    }
    if (bf16x_ok(g, 1, algo | io, tune)) {
        const int rc = bf16x_run(1, small, ld_small, P, bias, big, ld_big, g, act, io & PG_IO_BIG_BF16, ws, ws_bytes, st, x, tune.bf16ring);
        if (rc != BF16X_SKIP) return rc;
    }
    if (part || x.u_cache || mul.t) return PG_EINVAL;
    if (!io && s2b_tapnf_ok(g) && (ld_small % 4 == 0) && aligned16(small) &&
        tensor_bytes((long)g.N * g.Hs * g.Ws, ld_small, g.Ca) < FAST_LIMIT && tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb) < FAST_LIMIT)
        return launch_tapnf(false, small, ld_small, P, bias, big, ld_big, g, act, tensor_bytes((long)g.N * g.Hs * g.Ws, ld_small, g.Ca),
                            tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb), st);
    if (!io && s2b_tapn_ok(g) && (ld_small % 4 == 0) && aligned16(small) && aligned16(P) && aligned16(ws) &&
        ws_bytes >= s2b_tapn_ws(g) && tensor_bytes((long)g.N * g.Hs * g.Ws, ld_small, g.Ca) < FAST_LIMIT) {
        // D[small pixel][(tap, b)] = small . W' (row GEMM), then col2im: each big pixel sums the taps that reach it
        const int Nc = 16 * g.Cb;
        float* Wp = (float*)ws;
        float* D = Wp + (((size_t)Nc * g.Ca + 63) & ~(size_t)63);
        const float* W = P;
        if (g.Cb > 1) {
            hipLaunchKernelGGL(k_pack_taps_b, dim3((Nc * g.Ca + 255) / 256), dim3(256), 0, st, P, Wp, g.Ca, g.Cb);
            if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
            W = Wp;
        }
        const long Ms = (long)g.N * g.Hs * g.Ws;
        Geom g1{g.N, g.Hs, g.Ws, g.Hs, g.Ws, Nc, g.Ca, 1};
        Tile t = pick_tile(Ms, Nc);
        dim3 grid((unsigned)((Ms + t.bm - 1) / t.bm), (Nc + t.bn - 1) / t.bn, 1);
        {
            TimedLaunch timed(st);
            PG_DISPATCH_B2SF(true, t.id, grid, st, small, ld_small, W, D, Nc, 0L, g1, g.Ca / KC, (const float*)nullptr, 0,
                             (int)tensor_bytes(Ms, ld_small, g.Ca), (int)((long)Nc * g.Ca * 4));
        }
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        return launch_col2im(D, bias, big, ld_big, g, act, st);
    }
    Plan p = plan_s2b(gg);
    clamp_split(p, ws_bytes, 0);
    const bool in_bf = io & PG_IO_SMALL_BF16, out_bf = io & PG_IO_BIG_BF16;
    const int veck = (g.Ca % 4 == 0) && (ld_small % 4 == 0) && aligned_io(small, in_bf);
    const int vecn = (g.Cb % 4 == 0) && aligned16(P);
    dim3 grid(p.tiles_m, p.tiles_n, p.ncls * p.split);
    const long small_bytes = tensor_bytes((long)g.N * g.Hs * g.Ws, ld_small, g.Ca, in_bf), p_bytes = 16L * g.Ca * g.Cb * 4;
    const bool fast = veck && g.Ca >= KC && small_bytes < FAST_LIMIT && p_bytes < FAST_P_LIMIT && !force_generic();
    if (io && !fast) return PG_EINVAL;
    if (p.split == 1) {
        TimedLaunch timed(st);
        if (fast && algo == PG_ALGO_BF16 && in_bf) {
            PG_DISPATCH_S2BH(true, p.t.id, grid, st, small, ld_small, P, big, ld_big, 0L, g, p.cps, bias, act,
                             (int)small_bytes, (int)p_bytes, (int)out_bf);
        } else if (fast && algo == PG_ALGO_BF16) {
            PG_DISPATCH_S2BH(false, p.t.id, grid, st, small, ld_small, P, big, ld_big, 0L, g, p.cps, bias, act,
                             (int)small_bytes, (int)p_bytes, (int)out_bf);
        } else if (fast) {
            PG_DISPATCH_TILE(k_s2b_fast, p.t.id, grid, st, small, ld_small, P, big, ld_big, 0L, g, p.cps, bias, act,
                             (int)small_bytes, (int)p_bytes);
        } else {
            PG_DISPATCH_TILE(k_small2big, p.t.id, grid, st, small, ld_small, P, big, ld_big, 0L, g, p.cps, veck, vecn,
                             bias, act);
        }
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    float* slabs = (float*)ws;
    {
        TimedLaunch timed(st);
        if (fast && algo == PG_ALGO_BF16 && in_bf) {
            PG_DISPATCH_S2BH(true, p.t.id, grid, st, small, ld_small, P, slabs, g.Cb, p.out_elems, g, p.cps,
                             (const float*)nullptr, 0, (int)small_bytes, (int)p_bytes, 0);
        } else if (fast && algo == PG_ALGO_BF16) {
            PG_DISPATCH_S2BH(false, p.t.id, grid, st, small, ld_small, P, slabs, g.Cb, p.out_elems, g, p.cps,
                             (const float*)nullptr, 0, (int)small_bytes, (int)p_bytes, 0);
        } else if (fast) {
            PG_DISPATCH_TILE(k_s2b_fast, p.t.id, grid, st, small, ld_small, P, slabs, g.Cb, p.out_elems, g, p.cps,
                             (const float*)nullptr, 0, (int)small_bytes, (int)p_bytes);
        } else {
            PG_DISPATCH_TILE(k_small2big, p.t.id, grid, st, small, ld_small, P, slabs, g.Cb, p.out_elems, g, p.cps, veck,
                             vecn, (const float*)nullptr, 0);
        }
    }
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    return launch_reduce(slabs, p.out_elems, p.split, big, ld_big, (long)g.N * g.Hb * g.Wb, g.Cb, bias, act, st, out_bf);
}

int pg_conv4x4_small2big(const float* small, int ld_small, const float* P, const float* bias, float* big,
                         int ld_big, const pg_conv_geom* gg, int act, int algo, void* ws, size_t ws_bytes,
                         void* stream) {
    return s2b_impl(small, ld_small, P, bias, big, ld_big, gg, act, algo, ws, ws_bytes, stream, nullptr);
}

int pg_conv4x4_small2big_x(const float* small, int ld_small, const float* P, const float* bias, float* big,
                           int ld_big, const pg_conv_geom* gg, int act, int algo, void* ws, size_t ws_bytes,
                           void* stream, const pg_conv_extras* x) {
    return s2b_impl(small, ld_small, P, bias, big, ld_big, gg, act, algo, ws, ws_bytes, stream, x);
}

size_t pg_conv_u_bytes(const pg_conv_geom* gg, int op, int algo, size_t ws_bytes) {
    if (!geom_ok(gg) || (op != 0 && op != 1)) return 0;
    const Geom g = to_geom(gg);
    const Tune tune = tune_of(algo);
    if (bf16x_ok(g, op, algo, tune)) return pg_bf16x_w_bytes(g.Ca, g.Cb);
    if ((algo & PG_ALGO_MASK) != PG_ALGO_AUTO) return 0;
    if (op == 0) {
        if (wino_b2s_ok(g, tune) && ws_bytes >= pg_wino_ws_bytes(g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1))
            return pg_wino_u_bytes(g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1);
        if (wino2_b2s_ok(g, tune) && ws_bytes >= pg_wino2_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb)) return pg_wino2_u_bytes(g.Ca, g.Cb);
        return 0;
    }
    if (wino_s2b_ok(g, tune) && ws_bytes >= pg_wino_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1))
        return pg_wino_u_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1);
    if (wino2_s2b_ok(g, tune) && ws_bytes >= pg_wino2c_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb)) return pg_wino2_u_bytes(g.Ca, g.Cb);
    return 0;
}

int pg_conv_prep_batch(int n, const pg_conv_prep_item* items, void* stream) {
    if (n < 0 || (n > 0 && !items)) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    std::vector<pg_wino_prep> wv;
    std::vector<pg_bf16x_pack_item> bv;
    for (int i = 0; i < n; ++i) {      // mirrors pg_conv_u_bytes / the dispatch of b2s_impl and s2b_impl
        const pg_conv_prep_item& it = items[i];
        if (!geom_ok(&it.g) || (it.op != 0 && it.op != 1) || !it.P || !it.u || !aligned16(it.P) || !aligned16(it.u)) return PG_EINVAL;
        const Geom g = to_geom(&it.g);
        const Tune tune = tune_of(it.algo);
        const int op = it.op;
        if ((it.algo & PG_ALGO_MASK) == PG_ALGO_AUTO) {
            if (op == 0 && wino_b2s_ok(g, tune) && it.ws_bytes >= pg_wino_ws_bytes(g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1))
                wv.push_back(pg_wino_prep{it.P, (float*)it.u, g.Ca, g.Cb, 0, pg_wino_mo(g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1), 0});
            else if (op == 0 && wino2_b2s_ok(g, tune) && it.ws_bytes >= pg_wino2_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb))
                wv.push_back(pg_wino_prep{it.P, (float*)it.u, g.Ca, g.Cb, 1, 0, 0});
            else if (op == 1 && wino_s2b_ok(g, tune) && it.ws_bytes >= pg_wino_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1))
                wv.push_back(pg_wino_prep{it.P, (float*)it.u, g.Cb, g.Ca, 0, pg_wino_mo(g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1), 1});
            else if (op == 1 && wino2_s2b_ok(g, tune) && it.ws_bytes >= pg_wino2c_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb))
                wv.push_back(pg_wino_prep{it.P, (float*)it.u, g.Ca, g.Cb, 2, 0, 0});
            else
                return PG_EINVAL;
        } else if (bf16x_ok(g, op, it.algo, tune)) {
            // big -> small: [tap][a][b] (8-channel-pixel form for a few-channel big); small -> big: the same pack (transposed staging in
            // the kernel) unless the ring-staged variant is pinned (bf16x_run)
            const bool own = g.Cb > 8 && pg_bf16x_plan_of(op, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s, tune.bf16ring).win;
            const int dir = own ? 4 + op : (op == 0) ? (g.Cb <= 8 ? 2 : 0) : (tune.bf16ring > 0 ? 1 : 0);
            bv.push_back(pg_bf16x_pack_item{it.P, it.u, g.Ca, g.Cb, dir});
        } else {
            return PG_EINVAL;
        }
    }
    for (size_t o = 0; o < wv.size(); o += PG_WINO_PREP_MAX) {
        const int rc = pg_wino_prep_batch((int)std::min<size_t>(PG_WINO_PREP_MAX, wv.size() - o), wv.data() + o, st);
        if (rc != PG_OK) return rc;
    }
    for (size_t o = 0; o < bv.size(); o += PG_BF16X_PACK_MAX) {
        const int rc = pg_bf16x_pack_batch((int)std::min<size_t>(PG_BF16X_PACK_MAX, bv.size() - o), bv.data() + o, st);
        if (rc != PG_OK) return rc;
    }
    return PG_OK;
}

int pg_conv_mul_ok(const pg_conv_geom* gg, int algo, size_t ws_bytes) {
    if (!geom_ok(gg)) return 0;
    const Geom g = to_geom(gg);
    const Tune tune = tune_of(algo);
    // mirrors the dispatch of s2b_impl for 16-byte-aligned tensors
    if (s2b_ca1_ok(g, algo)) return 1;
    if ((algo & PG_ALGO_MASK) == PG_ALGO_AUTO) {
        if (wino_s2b_ok(g, tune) && ws_bytes >= pg_wino_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1)) return 1;
        if (wino2_s2b_ok(g, tune) && ws_bytes >= pg_wino2c_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb)) return 1;
        return 0;
    }
    if (bf16x_s2b_tapn_ok(g, algo, tune) && ws_bytes >= bf16x_s2b_tapn_ws(g)) return 0;
    return bf16x_ok(g, 1, algo, tune) && ws_bytes >= bf16x_ws(g, 1) ? 1 : 0;
}

size_t pg_conv_v_bytes(const pg_conv_geom* gg, int algo, size_t ws_bytes) {
    if (!geom_ok(gg) || (algo & PG_ALGO_MASK) != PG_ALGO_AUTO) return 0;
    const Geom g = to_geom(gg);
    const Tune tune = tune_of(algo);
    // stride-1 layer: forward on F(3x3,4x4), weight gradient on F(4x4,3x3) -- one transformed input
    if (wino_b2s_ok(g, tune) && wino_wgrad_ok(g, tune) && ws_bytes >= pg_wino_ws_bytes(g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1) &&
        ws_bytes >= (((size_t)COLSUM_CHUNKS * g.Ca * sizeof(float) + 255) & ~(size_t)255) + pg_wino_wgrad_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb))
        return pg_wino_wgrad_v_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb, tune.mo1);
    if (pg_wino2_mo() != 3) return 0;
    const size_t colsum = ((size_t)COLSUM_CHUNKS * g.Ca * sizeof(float) + 255) & ~(size_t)255;
    if (wino2_b2s_ok(g, tune) && ws_bytes >= pg_wino2_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb) && wino2_wgrad_ok(g, tune) &&
        ws_bytes >= colsum + pg_wino2_wgrad_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb))
        return pg_wino2_v_bytes(g.N, g.Hs, g.Ws, g.Cb);
    return 0;
}

int pg_conv_stats_chunks(const pg_conv_geom* gg, int op, int algo, size_t ws_bytes) {
    if (!geom_ok(gg) || (op != 0 && op != 1)) return 0;
    const Geom g = to_geom(gg);
    const Tune tune = tune_of(algo);
    if ((algo & PG_IO_MASK) == PG_IO_MASK && bf16x_ok(g, op, algo, tune) && !(op == 0 && g.Cb <= 8) && ws_bytes >= bf16x_ws(g, op)) {
        pg_bf16x_plan bp = pg_bf16x_plan_of(op, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s, tune.bf16ring);
        pg_bf16x_clamp(&bp, ws_bytes - pg_bf16x_w_bytes(g.Ca, g.Cb));
        return pg_bf16x_stats_chunks(op, &bp, g.N, g.Hb, g.Wb, g.Hs, g.Ws);
    }
    if ((algo & PG_ALGO_MASK) != PG_ALGO_AUTO) return 0;
    // mirrors the dispatch of b2s_impl / s2b_impl for 16-byte-aligned tensors: the stride-1 Winograd path comes first
    if (op == 0) {
        if (wino_b2s_ok(g, tune) && ws_bytes >= pg_wino_ws_bytes(g.N, g.Hs, g.Ws, g.Cb, g.Ca, tune.mo1)) return 0;
        if (wino2_b2s_ok(g, tune) && ws_bytes >= pg_wino2_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb))
            return pg_wino2_b2s_stats_chunks(g.N, g.Hs, g.Ws, g.Ca);
        if (tapk_enabled() && tapkp_enabled()) return tapkp_stats_chunks(g);      // image-facing layer (enc0): k_b2s_tapkp<.., STATS>
        return 0;
    }
    if (wino_s2b_ok(g, tune) && ws_bytes >= pg_wino_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb, tune.mo1)) return 0;
    if (wino2_s2b_ok(g, tune) && ws_bytes >= pg_wino2c_ws_bytes(g.N, g.Hb, g.Wb, g.Ca, g.Cb))
        return pg_wino2_s2b_stats_chunks(g.N, g.Hb, g.Wb, g.Cb);
    return 0;
}

static int wgrad_impl(const float* small, int ld_small, const float* big, int ld_big, float* dP, float* dbias,
                      const pg_conv_geom* gg, int algo, void* ws, size_t ws_bytes, void* stream, const float* v_pre) {
    if (!geom_ok(gg) || !big || !dP || !small || ld_big < gg->Cb || ld_small < gg->Ca) return PG_EINVAL;
    if (v_pre && !aligned16(v_pre)) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    Geom g = to_geom(gg);
    const Tune tune = tune_of(algo);
    const int io = algo & PG_IO_MASK;                 // both activation operands bf16, or neither
    algo &= PG_ALGO_MASK;
    if (io && (algo != PG_ALGO_BF16 || io != PG_IO_MASK)) return PG_EINVAL;
    if (io && dbias && !((g.Ca % 4 == 0) && (ld_small % 4 == 0) && aligned_io(small, true))) return PG_EINVAL;
    if (!ws) ws_bytes = 0;
    const long Kp = (long)g.N * g.Hs * g.Ws;
    size_t reserved = 0;
    if (dbias) {
        reserved = ((size_t)COLSUM_CHUNKS * g.Ca * sizeof(float) + 255) & ~(size_t)255;
        if (ws_bytes < reserved) return PG_EWORKSPACE;
        float* part = (float*)ws;
        int chunks = (int)std::min<long>(COLSUM_CHUNKS, Kp);
        long rpc = (Kp + chunks - 1) / chunks;
        chunks = (int)((Kp + rpc - 1) / rpc);
        if (io)
            hipLaunchKernelGGL(k_colsum_partial4<true>, dim3(chunks, (g.Ca + 63) / 64), dim3(256), 0, st, small, ld_small, Kp, g.Ca,
                               rpc, part);
        else if ((g.Ca % 4 == 0) && (ld_small % 4 == 0) && aligned16(small) && aligned16(part))
            hipLaunchKernelGGL(k_colsum_partial4<false>, dim3(chunks, (g.Ca + 63) / 64), dim3(256), 0, st, small, ld_small, Kp, g.Ca,
                               rpc, part);
        else
            hipLaunchKernelGGL(k_colsum_partial, dim3(chunks, (g.Ca + 63) / 64), dim3(256), 0, st, small, ld_small, Kp, g.Ca,
                               rpc, part);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        int rc = launch_reduce(part, g.Ca, chunks, dbias, g.Ca, 1, g.Ca, nullptr, 0, st);
        if (rc != PG_OK) return rc;
    }
    const long per = 16L * g.Ca * g.Cb;
    if (algo == PG_ALGO_DIRECT) {
        long smax = (long)((ws_bytes - reserved) / (sizeof(float) * (size_t)per));
        int slices = (int)std::min<long>(std::min<long>(DIRECT_WGRAD_SLICES, smax), Kp);
        if (slices < 1) slices = 1;
        long pps = (Kp + slices - 1) / slices;
        slices = (int)((Kp + pps - 1) / pps);
        float* dst = slices == 1 ? dP : (float*)((char*)ws + reserved);
        hipLaunchKernelGGL(k_wgrad_direct, dim3((int)((per + 255) / 256), slices), dim3(256), 0, st, small, ld_small, big,
                           ld_big, dst, per, g, pps);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        if (slices == 1) return PG_OK;
        return launch_reduce(dst, per, slices, dP, g.Cb, 16L * g.Ca, g.Cb, nullptr, 0, st);
    }
    if (algo == PG_ALGO_AUTO && wino_wgrad_ok(g, tune) && (ld_small % 4 == 0) && (ld_big % 4 == 0) && aligned16(small) &&
        aligned16(big) && aligned16(ws) && ws_bytes >= reserved + pg_wino_wgrad_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb)) {
        hipEvent_t e0 = t_ev0, e1 = t_ev1;
        t_ev0 = nullptr;
        t_ev1 = nullptr;
        if (v_pre && !pg_wino_wgrad_v_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb, tune.mo1)) return PG_EINVAL;
        return pg_wino_wgrad(small, ld_small, big, ld_big, dP, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, (char*)ws + reserved, st,
                             e0, e1, v_pre, tune.s3w);
    }
    if (algo == PG_ALGO_AUTO && wino2_wgrad_ok(g, tune) && (ld_small % 4 == 0) && (ld_big % 4 == 0) && aligned16(small) &&
        aligned16(big) && aligned16(ws) && aligned16(dP) && ws_bytes >= reserved + pg_wino2_wgrad_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb)) {
        hipEvent_t e0 = t_ev0, e1 = t_ev1;
        t_ev0 = nullptr;
        t_ev1 = nullptr;
        return pg_wino2_wgrad(small, ld_small, big, ld_big, dP, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, (char*)ws + reserved, st,
                              e0, e1, v_pre, tune.s3w);
    }
    if (v_pre) return PG_EINVAL;     // pg_conv_v_bytes said 0 for this call: there is no transformed operand to reuse
    if (bf16x_wgrad_ok(g, algo | io, tune) && (g.Cb > 8 || ld_big == 8) && aligned_bf_view(small, ld_small, true) &&
        aligned_bf_view(big, ld_big, true) && aligned16(dP) && (ws_bytes <= reserved || aligned16(ws))) {
        const long small_bytes = tensor_bytes(Kp, ld_small, g.Ca, true);
        const long big_bytes = tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb > 8 ? g.Cb : 8, true);
        if (small_bytes < FAST_LIMIT && big_bytes < FAST_LIMIT) {
            pg_bf16x_plan wp = pg_bf16x_wgrad_plan(g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s);
            pg_bf16x_clamp(&wp, ws_bytes > reserved ? ws_bytes - reserved : 0);
            float* dst = wp.split == 1 ? dP : (float*)((char*)ws + reserved);
            int rc;
            {
                TimedLaunch timed(st);
                rc = pg_bf16x_wgrad(small, ld_small, small_bytes, big, ld_big, big_bytes, dst, wp.split == 1 ? 0L : wp.out_elems, g.N, g.Hb,
                                    g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, g.s, &wp, st);
            }
            if (rc != PG_OK || wp.split == 1) return rc;
            return launch_reduce(dst, wp.out_elems, wp.split, dP, g.Cb, 16L * g.Ca, g.Cb, nullptr, 0, st);
        }
    }
    Plan p = plan_wgrad(gg);
    clamp_split(p, ws_bytes, reserved);
    const bool in_bf = io != 0;
    const int vecm = (g.Ca % 4 == 0) && (ld_small % 4 == 0) && aligned_io(small, in_bf);
    const int vecn = (g.Cb % 4 == 0) && (ld_big % 4 == 0) && aligned_io(big, in_bf);
    const int mode = wgrad_mode(gg);
    if (io && mode != 0) return PG_EINVAL;
    float* dst = p.split == 1 ? dP : (float*)((char*)ws + reserved);
    TimedLaunch* timed = new (alloca(sizeof(TimedLaunch))) TimedLaunch(st);
    if (mode == 0) {
        dim3 grid(p.tiles_m * p.tiles_n, 16, p.split);
        const long small_bytes = tensor_bytes(Kp, ld_small, g.Ca, in_bf);
        const long big_bytes = tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb, in_bf);
        const bool pow2 = ((g.Hs & (g.Hs - 1)) == 0) && ((g.Ws & (g.Ws - 1)) == 0);
        const bool fast = vecm && vecn && small_bytes < FAST_LIMIT && big_bytes < FAST_LIMIT && !force_generic() &&
                          (pow2 || (g.Ws >= 16 && g.Hs >= 2));
        if (io && !fast) {
            timed->~TimedLaunch();
            return PG_EINVAL;
        }
        if (fast && algo == PG_ALGO_BF16 && pow2 && in_bf) {
            PG_DISPATCH_WGH(true, true, p.t.id, grid, st, small, ld_small, big, ld_big, dst, p.out_elems, g, p.cps, p.tiles_n,
                            (int)small_bytes, (int)big_bytes);
        } else if (fast && algo == PG_ALGO_BF16 && in_bf) {
            PG_DISPATCH_WGH(false, true, p.t.id, grid, st, small, ld_small, big, ld_big, dst, p.out_elems, g, p.cps, p.tiles_n,
                            (int)small_bytes, (int)big_bytes);
        } else if (fast && algo == PG_ALGO_BF16 && pow2) {
            PG_DISPATCH_WGH(true, false, p.t.id, grid, st, small, ld_small, big, ld_big, dst, p.out_elems, g, p.cps, p.tiles_n,
                            (int)small_bytes, (int)big_bytes);
        } else if (fast && algo == PG_ALGO_BF16) {
            PG_DISPATCH_WGH(false, false, p.t.id, grid, st, small, ld_small, big, ld_big, dst, p.out_elems, g, p.cps, p.tiles_n,
                            (int)small_bytes, (int)big_bytes);
        } else if (fast && pow2) {
            PG_DISPATCH_WGF(true, p.t.id, grid, st, small, ld_small, big, ld_big, dst, p.out_elems, g, p.cps, p.tiles_n,
                            (int)small_bytes, (int)big_bytes);
        } else if (fast) {
            PG_DISPATCH_WGF(false, p.t.id, grid, st, small, ld_small, big, ld_big, dst, p.out_elems, g, p.cps, p.tiles_n,
                            (int)small_bytes, (int)big_bytes);
        } else {
            PG_DISPATCH_TILE(k_wgrad, p.t.id, grid, st, small, ld_small, big, ld_big, dst, p.out_elems, g, p.cps,
                             p.tiles_n, vecm, vecn);
        }
    } else if (mode == 1 && wgrad_tapnp_ok(g) && vecm && aligned16(ws) &&
               (g.Cb != 4 || ((ld_big % 4 == 0) && aligned16(big))) && tensor_bytes(Kp, ld_small, g.Ca) < FAST_LIMIT &&
               tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb) < FAST_LIMIT &&
               ws_bytes >= reserved + (size_t)wgrad_tapnp_slabs(g) * p.out_elems * sizeof(float)) {
        // persistent form: one slab per workgroup column, reduced in workgroup order
        const int ntiles = (int)((Kp + 127) / 128), G = wgrad_tapnp_slabs(g);
        const int small_b = (int)tensor_bytes(Kp, ld_small, g.Ca), big_b = (int)tensor_bytes((long)g.N * g.Hb * g.Wb, ld_big, g.Cb);
        const bool wide3 = (g.Cb == 3) && (ld_big % 4 == 0) && (ld_big >= 4) && aligned16(big);
        float* slabs = (float*)((char*)ws + reserved);
        if (g.Cb <= 2) {
            dim3 grid(G, (g.Ca + 127) / 128, 1);
            if (g.Cb == 1) hipLaunchKernelGGL((k_wgrad_tapnp<1, false, 4>), grid, dim3(256), 0, st, small, ld_small, big, ld_big, slabs, p.out_elems, g, small_b, big_b, ntiles);
            else hipLaunchKernelGGL((k_wgrad_tapnp<2, false, 4>), grid, dim3(256), 0, st, small, ld_small, big, ld_big, slabs, p.out_elems, g, small_b, big_b, ntiles);
        } else {
            dim3 grid(G, (g.Ca + 63) / 64, 1);
            if (g.Cb == 4) hipLaunchKernelGGL((k_wgrad_tapnp<4, true, 2>), grid, dim3(256), 0, st, small, ld_small, big, ld_big, slabs, p.out_elems, g, small_b, big_b, ntiles);
            else if (wide3) hipLaunchKernelGGL((k_wgrad_tapnp<3, true, 2>), grid, dim3(256), 0, st, small, ld_small, big, ld_big, slabs, p.out_elems, g, small_b, big_b, ntiles);
            else hipLaunchKernelGGL((k_wgrad_tapnp<3, false, 2>), grid, dim3(256), 0, st, small, ld_small, big, ld_big, slabs, p.out_elems, g, small_b, big_b, ntiles);
        }
        timed->~TimedLaunch();
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        return launch_reduce(slabs, p.out_elems, G, dP, g.Cb, 16L * g.Ca, g.Cb, nullptr, 0, st);
    } else if (mode == 1) {
        dim3 grid(p.tiles_m * p.tiles_n, 1, p.split);
        const int vecy = (g.Cb == 4) && (ld_big % 4 == 0) && aligned16(big);
        PG_DISPATCH_TAPN(1, p.t.id, grid, st, small, ld_small, big, ld_big, dst, p.out_elems, g, p.cps, p.tiles_n, vecm,
                         vecy);
    } else {
        dim3 grid(p.tiles_m * p.tiles_n, 1, p.split);
        PG_DISPATCH_TAPN(2, p.t.id, grid, st, big, ld_big, small, ld_small, dst, p.out_elems, g, p.cps, p.tiles_n, vecn, 0);
    }
    timed->~TimedLaunch();
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    if (p.split == 1) return PG_OK;
    return launch_reduce(dst, p.out_elems, p.split, dP, g.Cb, 16L * g.Ca, g.Cb, nullptr, 0, st);
}

int pg_conv4x4_wgrad(const float* small, int ld_small, const float* big, int ld_big, float* dP, float* dbias,
                     const pg_conv_geom* gg, int algo, void* ws, size_t ws_bytes, void* stream) {
    return wgrad_impl(small, ld_small, big, ld_big, dP, dbias, gg, algo, ws, ws_bytes, stream, nullptr);
}

int pg_conv4x4_wgrad_x(const float* small, int ld_small, const float* big, int ld_big, float* dP, float* dbias,
                       const pg_conv_geom* gg, int algo, void* ws, size_t ws_bytes, void* stream, const pg_conv_extras* x) {
    if (x && (x->part || x->v_keep || x->u_cache)) return PG_EINVAL;
    return wgrad_impl(small, ld_small, big, ld_big, dP, dbias, gg, algo, ws, ws_bytes, stream, x ? x->v_pre : nullptr);
}

int pg_conv4x4_bwd_big(const float* small, int ld_small, const float* big, int ld_big, const float* P, float* dP,
                       float* dsmall, int ld_dsmall, const pg_conv_geom* gg, int algo, void* ws, size_t ws_bytes, void* stream) {
    return pg_conv4x4_bwd_big_x(small, ld_small, big, ld_big, P, dP, dsmall, ld_dsmall, gg, algo, ws, ws_bytes, stream, nullptr);
}

int pg_conv4x4_bwd_big_x(const float* small, int ld_small, const float* big, int ld_big, const float* P, float* dP,
                         float* dsmall, int ld_dsmall, const pg_conv_geom* gg, int algo, void* ws, size_t ws_bytes, void* stream,
                         const pg_conv_extras* x) {
    if (!geom_ok(gg) || !small || !big || !P || !dP || !dsmall) return PG_EINVAL;
    if (x && (x->part || x->v_keep || x->v_pre || x->mul_t)) return PG_EINVAL;      // only the data gradient's u_cache / u_valid
    float* const Uext = x ? x->u_cache : nullptr;
    const int u_valid = x ? x->u_valid : 0;
    if (Uext && !aligned16(Uext)) return PG_EINVAL;
    if (ld_small < gg->Ca || ld_big < gg->Cb || ld_dsmall < gg->Ca) return PG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const Geom g = to_geom(gg);
    const Tune tune = tune_of(algo);
    if (!ws) ws_bytes = 0;
    const size_t vb = pg_wino2_v_bytes(g.N, g.Hs, g.Ws, g.Cb);
    const bool share = (algo & PG_ALGO_MASK) == PG_ALGO_AUTO && pg_wino2_mo() == 3 && wino2_b2s_ok(g, tune) && wino2_wgrad_ok(g, tune) &&
                       (ld_small % 4 == 0) && (ld_big % 4 == 0) && (ld_dsmall % 4 == 0) && aligned16(small) && aligned16(big) &&
                       aligned16(P) && aligned16(dP) && aligned16(dsmall) && aligned16(ws) &&
                       ws_bytes >= vb + std::max(pg_wino2_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb),
                                                 pg_wino2_wgrad_ws_bytes(g.N, g.Hs, g.Ws, g.Ca, g.Cb));
    hipEvent_t e0 = t_ev0, e1 = t_ev1, e2 = t_ev2, e3 = t_ev3;
    if (!share) {       // the two halves as separate calls (each consumes one armed event pair)
        t_ev2 = t_ev3 = nullptr;
        int rc = pg_conv4x4_wgrad(small, ld_small, big, ld_big, dP, nullptr, gg, algo, ws, ws_bytes, stream);
        if (rc != PG_OK) return rc;
        t_ev0 = e2;
        t_ev1 = e3;
        if (Uext) {
            pg_conv_extras xu{};
            xu.u_cache = Uext;
            xu.u_valid = u_valid;
            return pg_conv4x4_big2small_x(big, ld_big, P, nullptr, dsmall, ld_dsmall, gg, PG_ACT_NONE, algo, ws, ws_bytes, stream, &xu);
        }
        return pg_conv4x4_big2small(big, ld_big, P, nullptr, dsmall, ld_dsmall, gg, PG_ACT_NONE, algo, ws, ws_bytes, stream);
    }
    t_ev0 = t_ev1 = t_ev2 = t_ev3 = nullptr;
    // V(big) once; the weight-gradient GEMM and the data-gradient GEMM both read it (ws: V | DY S, then V | U M)
    float* V = (float*)ws;
    void* rest = (char*)ws + vb;
    int rc = pg_wino2_v(big, ld_big, V, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Cb, st);
    if (rc != PG_OK) return rc;
    rc = pg_wino2_wgrad(small, ld_small, big, ld_big, dP, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, rest, st, e0, e1, V, tune.s3w);
    if (rc != PG_OK) return rc;
    return pg_wino2_b2s(big, ld_big, P, nullptr, dsmall, ld_dsmall, g.N, g.Hb, g.Wb, g.Hs, g.Ws, g.Ca, g.Cb, PG_ACT_NONE, rest, st, e2,
                        e3, V, nullptr, nullptr, Uext, u_valid, tune.s3);
}

}  // extern "C"
