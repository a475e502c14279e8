// conv_wino.hip -- Winograd F(2x2, 4x4) for the stride-1 4x4 convolutions of the PatchGAN discriminator.
//
// The layer d_n of reference disc.py:37 (Conv2d k4 s1 p1, 256 -> 512 channels on a 32x32 map at cfg2) carries 37 % of
// all FLOPs of a G+D step (64.5 GFLOP forward per 16 images, run on 16 + 32 images forward and again as data-gradient).
// A stride-1 4x4 correlation   out[y,x,co] = sum_{i,j,ci} in[y-pad+i, x-pad+j, ci] * Wt[i][j][co][ci]
// is computed per 2x2 output tile from a 5x5 input tile as
//        Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A
// (Toom-Cook points 0, 1, -1, 2, inf): 25 multiplies per tile and channel pair instead of 2*2*16 = 64, i.e. 2.56x fewer
// MFMA FLOPs, exact in exact arithmetic; in fp32 the measured error is 1.5e-6 relative (direct fp32: 1e-7), inside the
// 1e-5 per-layer parity bound.  Three kernels:
//   k_wino_u     U[xi][co][ci]   = (G g G^T)[xi]           weights, 25*Co*Ci floats, once per call
//   k_wino_v     V[xi][tile][ci] = (B^T d B)[xi]           input tiles (zero padded), HBM-bound
//   k_wino_gemm  25 row GEMMs M_xi = V_xi U_xi^T on v_mfma_f32_32x32x2_f32 with the output transform A^T M A, bias and
//                activation fused: a workgroup owns 128 tiles x 64 channels, loops xi = 0..24, accumulates each M_xi in
//                one set of MFMA accumulators and folds it into the four output accumulators (2x2 positions) with the
//                A^T coefficients, so M never goes to memory.
// Forward uses (in, out, pad, Wt) = (big, small, 1, P[i*4+j][co][ci]); the data-gradient uses (small, big, 2,
// P[(3-i)*4+(3-j)][ci][co]) -- the same correlation with the kernel flipped and the channel roles swapped.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "patchgan_hip.h"
#include "pg_common.h"
#include "conv_wino.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;
constexpr int LDK = KC + 4;


// Toom-Cook matrices for F(2, 4), points {0, 1, -1, 2, inf}
__device__ __constant__ float c_BT[5][5] = {{2, -1, -2, 1, 0}, {0, -2, -1, 1, 0}, {0, 2, -3, 1, 0}, {0, -1, 0, 1, 0},
                                            {0, 2, -1, -2, 1}};
__device__ __constant__ float c_G[5][4] = {{0.5f, 0, 0, 0},
                                           {-0.5f, -0.5f, -0.5f, -0.5f},
                                           {-1.f / 6, 1.f / 6, -1.f / 6, 1.f / 6},
                                           {1.f / 6, 1.f / 3, 2.f / 3, 4.f / 3},
                                           {0, 0, 0, 1}};

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ int voff(int elem_off, bool ok) {
    return (int)(((unsigned)elem_off << 2) | (ok ? 0u : 0x80000000u));
}
__device__ __noinline__ float act_slow(float v, int act) { return pg_act(v, act); }
__device__ __forceinline__ float act_epi(float v, int act) {
    if (act == PG_ACT_NONE) return v;
    if (act == PG_ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    if (act == PG_ACT_RELU) return v > 0.f ? v : 0.f;
    return act_slow(v, act);
}

// F(3x3, 4x4) on the points {0, 1, -1, 1/2, -2, inf}: 36 instead of 144 multiplies per 3x3 output tile and channel pair (4x;
// F(2x2, 4x4) above: 2.56x) at the same fp32 error in simulation (4.5e-6 vs 4.6e-6: the reciprocal pair 1/2, -2 keeps the
// transform entries small).  MO = 2 / 3 selects the variant in the three kernels below; NP = MO + 3 points.
__device__ __constant__ float c_BT6[6][6] = {{1, -1.5f, -2, 1.5f, 1, 0}, {0, -1, 0.5f, 2.5f, 1, 0}, {0, 1, -2.5f, 0.5f, 1, 0},
                                             {0, -2, -1, 2, 1, 0}, {0, 0.5f, -1, -0.5f, 1, 0}, {0, 1, -1.5f, -2, 1.5f, 1}};
__device__ __constant__ float c_G34[6][4] = {{1, 0, 0, 0},
                                             {1.f / 3, 1.f / 3, 1.f / 3, 1.f / 3},
                                             {-1.f / 3, 1.f / 3, -1.f / 3, 1.f / 3},
                                             {-16.f / 15, -8.f / 15, -4.f / 15, -2.f / 15},
                                             {1.f / 15, -2.f / 15, 4.f / 15, -8.f / 15},
                                             {0, 0, 0, 1}};
__device__ __constant__ float c_AT34[3][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, -2, 0}, {0, 1, 1, 0.25f, 4, 1}};
__device__ __constant__ float c_AT24[2][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 2, 1}};
template <int MO> __device__ __forceinline__ float s1_bt(int a, int i) { return MO == 2 ? c_BT[a][i] : c_BT6[a][i]; }
template <int MO> __device__ __forceinline__ float s1_g(int a, int k) { return MO == 2 ? c_G[a][k] : c_G34[a][k]; }
template <int MO> __device__ __forceinline__ float s1_at(int k, int i) { return MO == 2 ? c_AT24[k][i] : c_AT34[k][i]; }

// U[xi][co][ci] = sum_{k,l} G[xi_i][k] G[xi_j][l] w(k, l, co, ci);  flip = 0: w = P[(k*4+l)][co][ci] (forward),
// flip = 1: w = P[((3-k)*4 + (3-l))][ci][co] (data gradient).  One thread per (co, ci).
template <int MO>
__device__ __forceinline__ void wino_u_body(const float* __restrict__ P, float* __restrict__ U, int Co, int Ci, int flip, long block) {
    constexpr int NP = MO + 3;
    const long idx = block * 256 + threadIdx.x;
    if (idx >= (long)Co * Ci) return;
    const int ci = (int)(idx % Ci), co = (int)(idx / Ci);
    float g[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 4; ++l)
            g[k][l] = flip ? P[((long)((3 - k) * 4 + (3 - l)) * Ci + ci) * Co + co]     // P[tap][a = ci][b = co]
                           : P[((long)(k * 4 + l) * Co + co) * Ci + ci];                // P[tap][a = co][b = ci]
    float t[NP][4];
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += s1_g<MO>(a, k) * g[k][l];
            t[a][l] = s;
        }
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            float s = 0.f;
#pragma unroll
            for (int l = 0; l < 4; ++l) s += t[a][l] * s1_g<MO>(b, l);
            U[((long)(a * NP + b) * Co + co) * Ci + ci] = s;
        }
}
template <int MO>
__global__ __launch_bounds__(256) void k_wino_u(const float* __restrict__ P, float* __restrict__ U, int Co, int Ci, int flip) {
    wino_u_body<MO>(P, U, Co, Ci, flip, blockIdx.x);
}

// V[xi][tile][ci] = (B^T d B)[xi], d[i][j] = in[n, 2*Ti - pad + i, 2*Tj - pad + j, ci] (0 outside).  One thread per
// (tile, 4 channels).
template <int MO>
__global__ __launch_bounds__(256) void k_wino_v(const float* __restrict__ in, int ld_in, float* __restrict__ V, int N,
                                                int Hin, int Win, int Ci, int TH, int TW, int pad) {
    constexpr int NP = MO + 3;
    const int cq = Ci >> 2;
    const long T = (long)N * TH * TW;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= T * cq) return;
    const int c0 = (int)(idx % cq) << 2;
    const long tile = idx / cq;
    const int n = (int)(tile / (TH * TW));
    const int rem = (int)(tile - (long)n * TH * TW);
    const int ti = rem / TW, tj = rem - ti * TW;
    const int y0 = MO * ti - pad, x0 = MO * tj - pad;
    f32x4 d[NP][NP];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int y = y0 + i, x = x0 + j;
            const bool ok = (unsigned)y < (unsigned)Hin && (unsigned)x < (unsigned)Win;
            d[i][j] = ok ? *reinterpret_cast<const f32x4*>(in + ((long)(n * Hin + y) * Win + x) * ld_in + c0) : z;
        }
    f32x4 t[NP][NP];
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            f32x4 s = z;
#pragma unroll
            for (int i = 0; i < NP; ++i) s += s1_bt<MO>(a, i) * d[i][j];
            t[a][j] = s;
        }
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            f32x4 s = z;
#pragma unroll
            for (int j = 0; j < NP; ++j) s += t[a][j] * s1_bt<MO>(b, j);
            *reinterpret_cast<f32x4*>(V + ((long)(a * NP + b) * T + tile) * Ci + c0) = s;
        }
}

// 25 row GEMMs + fused output transform.  Workgroup = 4 waves (2 x 2), tile = 128 Winograd tiles x 64 output channels,
// each wave 64 x 32 (two 32x32 MFMA tiles).  Requires Ci % 32 == 0.
// MUL: the epilogue also multiplies by f'(t) (pg_epi_mul) -- a separate instantiation, with unconditional (clamped) loads of t: a
// conditional load inside the fully unrolled epilogue sends the accumulator arrays to scratch memory
template <int MR, int NR, int WM, int WN, int WPE, int MO, bool MUL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wino_gemm(const float* __restrict__ V, const float* __restrict__ U,
                                                   const float* __restrict__ bias, float* __restrict__ out, int ld_out,
                                                   int T, int Ci, int Co, int TH, int TW, int Hout, int Wout, int act,
                                                   int v_bytes, int u_bytes, int tiles_n, pg_epi_mul mul, int nsl, long slab_stride) {
    // nsl > 1: the input-channel (K) range is split over nsl workgroups per output tile; workgroup `slice` writes its partial OUTPUTS
    // (the output transform is linear) to out + slice * slab_stride, the caller reduces the slabs in order (bias / act / mul there)
    static_assert(NR == 1 && WM * WN == 4, "one 32-column MFMA tile per wave");
    constexpr int NP = MO + 3, NXI = NP * NP, NY = MO * MO;      // points, products, outputs per tile
    // K chunk: 64 floats for the F(3x3,4x4) instance (its workgroup tile is only 64 x 64: twice the MFMAs per barrier pair)
    constexpr int KCL = (MO == 3) ? 64 : 32, LDL = KCL + 4, QR = KCL / 4, RP = 256 / QR;   // quads per row, rows per pass
    constexpr int BM = 32 * MR * WM, BN = 32 * NR * WN, AI = BM / RP, BI = BN / RP;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDL];
    float* As = smem;
    float* Bs = smem + BM * LDL;
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, u_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    // 1-D grid: work index w = tile block * tiles_n + channel block, each XCD a contiguous run of w -- the tiles_n workgroups
    // that read one slab of V sit next to each other on one XCD (one fabric fetch of V instead of tiles_n), and every XCD walks
    // xi in step, so the current U[xi] slab (Co*Ci floats) stays in its L2
    int w = pg_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int slice = w % nsl;
    const int m0 = (w / nsl) * BM;
    const int nch = Ci / KCL / nsl, total = NXI * nch, ch0 = slice * nch;      // this slice's chunks [ch0, ch0 + nch) of every xi
    out += slice * slab_stride;
    const int kq = tid % QR, r0 = tid / QR;

    int a_off[AI], b_off[BI];
    bool a_ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + RP * i;
        a_ok[i] = m < T;
        a_off[i] = min(m, T - 1) * Ci + kq * 4;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int co = n0 + r0 + RP * i;
        b_off[i] = (co < Co) ? co * Ci + kq * 4 : 0x10000000;
    }
    const int v_xi = T * Ci, u_xi = Co * Ci;     // elements per xi slab

    f32x4 ra[AI], rb[BI];
    int ld_xi = 0, ld_ch = 0;                    // (xi, chunk) of the NEXT load
    auto issue_loads = [&](bool on) {
        const int av = ld_xi * v_xi + (ch0 + ld_ch) * KCL, bu = ld_xi * u_xi + (ch0 + ld_ch) * KCL;
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[i] = bload4(rV, voff(a_off[i] + av, on && a_ok[i]));
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[i] = bload4(rU, voff(b_off[i] + bu, on));
        const bool wrap = ld_ch + 1 >= nch;
        ld_ch = wrap ? 0 : ld_ch + 1;
        ld_xi = wrap ? ld_xi + 1 : ld_xi;
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(r0 + RP * i) * LDL + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(r0 + RP * i) * LDL + kq * 4]) = rb[i];
    };

    f32x16 accm[MR], accy[NY][MR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accm[i][r] = 0.f;
#pragma unroll
            for (int p = 0; p < NY; ++p) accy[p][i][r] = 0.f;
        }

    issue_loads(true);
    store_chunk();
    __syncthreads();
    int xi = 0, ch = 0;
    for (int it = 0; it < total; ++it) {
        const bool more = it + 1 < total;
        issue_loads(more);
        __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
        for (int kk = 0; kk < KCL / 8; ++kk) {
            f32x4 af[MR], bf;
#pragma unroll
            for (int i = 0; i < MR; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MR + i) * 32 + lrow) * LDL + kk * 8 + lh * 4]);
            bf = *reinterpret_cast<const f32x4*>(&Bs[(wn * 32 + lrow) * LDL + kk * 8 + lh * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MR; ++i) accm[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[e], accm[i], 0, 0, 0);
        }
        if (ch == nch - 1) {
            // fold M_xi into the outputs: Y[al][be] += AT[al][xi_i] * AT[be][xi_j] * M_xi
            const int xa = xi / NP, xb = xi - xa * NP;
            float ca[MO], cb[MO];
#pragma unroll
            for (int k = 0; k < MO; ++k) {
                ca[k] = s1_at<MO>(k, xa);
                cb[k] = s1_at<MO>(k, xb);
            }
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mv = accm[i][r];
#pragma unroll
                    for (int al = 0; al < MO; ++al)
#pragma unroll
                        for (int be = 0; be < MO; ++be) accy[al * MO + be][i][r] += (ca[al] * cb[be]) * mv;
                    accm[i][r] = 0.f;
                }
        }
        const bool wrap = ch + 1 >= nch;
        ch = wrap ? 0 : ch + 1;
        xi = wrap ? xi + 1 : xi;
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    const int col = n0 + wn * 32 + lrow;
    const float bv = (bias != nullptr && col < Co) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int m = m0 + (wm * MR + i) * 32 + row;
            float tm[MUL ? NY : 1];
            if constexpr (MUL) {               // every lane loads (clamped address), the store below stays conditional
                const int mc = min(m, T - 1), colc = min(col, Co - 1);
                const int n = mc / (TH * TW);
                const int rem = mc - n * (TH * TW);
                const int ti = rem / TW, tj = rem - ti * TW;
#pragma unroll
                for (int p = 0; p < NY; ++p) {
                    const int y = min(MO * ti + p / MO, Hout - 1), x = min(MO * tj + p % MO, Wout - 1);
                    tm[p] = pg_act_grad_sel(((const float*)mul.t)[((long)(n * Hout + y) * Wout + x) * mul.ld + colc], mul.act);
                }
            }
            if (m < T && col < Co) {
                const int n = m / (TH * TW);
                const int rem = m - n * (TH * TW);
                const int ti = rem / TW, tj = rem - ti * TW;
#pragma unroll
                for (int p = 0; p < NY; ++p) {
                    const int y = MO * ti + p / MO, x = MO * tj + p % MO;
                    if (y < Hout && x < Wout) {
                        float v = act_epi(accy[p][i][r] + bv, act);
                        if constexpr (MUL) v *= tm[p];
                        out[((long)(n * Hout + y) * Wout + x) * ld_out + col] = v;
                    }
                }
            }
        }
}


// F(3x3,4x4) with the output transform cut in two (round 2, second half).  k_wino_gemm<..,3> keeps all nine output sets of a tile
// in registers (220 VGPRs, 2 waves per SIMD, 0.53-0.56 of the MFMA peak).  Here a workgroup owns ONE ROW i of the 6 x 6 products of its
// 64 tiles x 64 channels and folds only the column transform, T_i[c] = sum_j M_ij * AT[c][j]: three accumulator sets + M, 4 waves per
// SIMD, six times the workgroups (no K split needed at batch 16).  T (18 planes, half of what an unfused M would be) goes through
// memory once; k_wino_t_out finishes Y[r][c] = sum_i AT[r][i] * T_i[c] with bias, activation and the data-gradient multiplier.
template <int WPE, int MR = 1>      // MR: 32-row MFMA tiles per wave (64 * MR tiles per workgroup)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wino_gemm_row(
    const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ Tout, int T, int Ci, int Co, int v_bytes, int u_bytes,
    int tiles_n) {
    constexpr int NP = 6, KCL = 64, LDL = KCL + 4, QR = KCL / 4, RP = 256 / QR, BM = 64 * MR, BN = 64, AI = BM / RP, BI = BN / RP;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDL];
    float* As = smem;
    float* Bs = smem + BM * LDL;
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, u_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lh = lane >> 5;
    // work index = (row * tile blocks + tile block) * tiles_n + channel block, each XCD a contiguous run of it: the tiles_n workgroups
    // that share the V slabs of one (tile block, row) are neighbours, and everything resident on an XCD works on the SAME row, i.e.
    // walks the same six U slabs in step (512 KB each: L2-resident; with the row index fastest an XCD cycled through all 36)
    int w = pg_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int tiles_m = (T + BM - 1) / BM;
    const int xrow = w / tiles_m;
    const int m0 = (w % tiles_m) * BM;
    const int nch = Ci / KCL, total = NP * nch;
    const int kq = tid % QR, r0 = tid / QR;
    int a_off[AI], b_off[BI];
    bool a_ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + RP * i;
        a_ok[i] = m < T;
        a_off[i] = min(m, T - 1) * Ci + kq * 4;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int co = n0 + r0 + RP * i;
        b_off[i] = (co < Co) ? co * Ci + kq * 4 : 0x10000000;
    }
    const int v_xi = T * Ci, u_xi = Co * Ci;
    f32x4 ra[AI], rb[BI];
    int ld_j = 0, ld_ch = 0;
    auto issue_loads = [&](bool on) {
        const int xi = xrow * NP + ld_j;
        const int av = xi * v_xi + ld_ch * KCL, bu = xi * u_xi + ld_ch * KCL;
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[i] = bload4(rV, voff(a_off[i] + av, on && a_ok[i]));
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[i] = bload4(rU, voff(b_off[i] + bu, on));
        const bool wrap = ld_ch + 1 >= nch;
        ld_ch = wrap ? 0 : ld_ch + 1;
        ld_j = wrap ? ld_j + 1 : ld_j;
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(r0 + RP * i) * LDL + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(r0 + RP * i) * LDL + kq * 4]) = rb[i];
    };
    f32x16 accm[MR], acct[3][MR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accm[i][r] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) acct[c][i][r] = 0.f;
        }
    issue_loads(true);
    store_chunk();
    __syncthreads();
    int j = 0, ch = 0;
    for (int it = 0; it < total; ++it) {
        const bool more = it + 1 < total;
        issue_loads(more);
        __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
        for (int kk = 0; kk < KCL / 8; ++kk) {
            f32x4 af[MR];
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MR + i) * 32 + lrow) * LDL + kk * 8 + lh * 4]);
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[(wn * 32 + lrow) * LDL + kk * 8 + lh * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MR; ++i) accm[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[e], accm[i], 0, 0, 0);
        }
        if (ch == nch - 1) {
            const float c0 = c_AT34[0][j], c1 = c_AT34[1][j], c2 = c_AT34[2][j];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mv = accm[i][r];
                    acct[0][i][r] += c0 * mv;
                    acct[1][i][r] += c1 * mv;
                    acct[2][i][r] += c2 * mv;
                    accm[i][r] = 0.f;
                }
        }
        const bool wrap = ch + 1 >= nch;
        ch = wrap ? 0 : ch + 1;
        j = wrap ? j + 1 : j;
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }
    const int col = n0 + wn * 32 + lrow;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int m = m0 + (wm * MR + i) * 32 + row;
            if (m < T && col < Co) {
#pragma unroll
                for (int c = 0; c < 3; ++c) Tout[((long)(xrow * 3 + c) * T + m) * Co + col] = acct[c][i][r];
            }
        }
}

// Y[r][c] = sum_i AT[r][i] * T_i[c] per (tile, four channels), + bias, activation, optional multiplier f'(t); one thread per item
template <bool MUL>
__global__ __launch_bounds__(256) void k_wino_t_out(const float* __restrict__ Tt, const float* __restrict__ bias, float* __restrict__ out,
                                                    int ld_out, int T, int Co, int TH, int TW, int Hout, int Wout, int act, pg_epi_mul mul) {
    const int cq = Co >> 2;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)T * cq) return;
    const int c0 = (int)(idx % cq) << 2;
    const int m = (int)(idx / cq);
    const int n = m / (TH * TW);
    const int rem = m - n * (TH * TW);
    const int ti = rem / TW, tj = rem - ti * TW;
    f32x4 t[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) t[i][c] = *reinterpret_cast<const f32x4*>(Tt + ((long)(i * 3 + c) * T + m) * Co + c0);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) bv = *reinterpret_cast<const f32x4*>(bias + c0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int y = 3 * ti + r;
        if (y >= Hout) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int x = 3 * tj + c;
            if (x >= Wout) continue;
            f32x4 v = c_AT34[r][0] * t[0][c];
#pragma unroll
            for (int i = 1; i < 6; ++i) v += c_AT34[r][i] * t[i][c];
            const long pix = (long)(n * Hout + y) * Wout + x;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = act_epi(v[e] + bv[e], act);
            if constexpr (MUL) {
                const f32x4 tm = *reinterpret_cast<const f32x4*>((const float*)mul.t + pix * mul.ld + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= pg_act_grad_sel(tm[e], mul.act);
            }
            *reinterpret_cast<f32x4*>(out + pix * ld_out + c0) = o;
        }
    }
}

// The same batched GEMM + fused output transform for the 64-tile x 64-channel workgroup tile, with the operand tiles brought
// in by LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS without staging registers) into a ring of NSTAGE stages of
// 128 rows x 32 k, so the loads of chunk c + NSTAGE - 1 are in flight while chunk c is multiplied.  The register-staged
// kernel above prefetches ONE chunk ahead.  Built to test whether the F(3x3,4x4) instance (nine output accumulator sets, 2 waves
// per SIMD, 0.53 MFMA-pipe utilisation) is load-latency bound: it is not -- this kernel is as fast as the register-staged one
// (opt-in, PATCHGAN_WINO_DMA=1).  One barrier per chunk (stage c + NSTAGE - 1 was last read in iteration c - 1).
// LDS rows are 128 B linear with the 16-byte slot index XOR-swizzled by (row >> 1) & 7 on the SOURCE address (the DMA
// destination is wave-uniform base + lane * 16), so the ds_read_b128 fragment reads stay conflict-free without padding.
// Waves 0,1 stage the 64 V rows, waves 2,3 the 64 U rows; 4 DMA instructions per wave and chunk.
template <int MO, int NSTAGE, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wino_gemm_dma(
    const float* __restrict__ V, const float* __restrict__ U, const float* __restrict__ bias, float* __restrict__ out, int ld_out,
    int T, int Ci, int Co, int TH, int TW, int Hout, int Wout, int act, int v_bytes, int u_bytes, int tiles_n) {
    constexpr int NP = MO + 3, NXI = NP * NP, NY = MO * MO;
    constexpr int KD = 32, STAGE = 128 * KD;                      // floats per stage: 64 A rows then 64 B rows
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
    __shared__ __attribute__((aligned(1024))) float smem[NSTAGE * STAGE];
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, u_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably uniform: the DMA's LDS base goes to M0 directly
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lh = lane >> 5;
    const int w = pg_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (w / tiles_n) * 64, n0 = (w % tiles_n) * 64;
    const int nch = Ci / KD, total = NXI * nch;
    const bool isA = wave < 2;

    int voff[4];                                                   // byte offset of this lane's 16 bytes in piece jj, xi = chunk = 0
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int row = (wave & 1) * 32 + jj * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        const int g = (isA ? m0 : n0) + row, lim = isA ? T : Co;
        voff[jj] = (int)(((unsigned)(min(g, lim - 1) * Ci + slot * 4) << 2) | (g < lim ? 0u : 0x80000000u));
    }
    const int xi_bytes = (isA ? T : Co) * Ci * 4;                  // bytes per xi slab of this wave's operand
    float* const dst0 = smem + (isA ? 0 : 64 * KD) + (wave & 1) * 32 * KD;
    int ld_ch = 0, ld_off = 0;                                     // chunk / byte offset of the NEXT chunk to load
    auto dma = [&](int stage, bool on) {
        const unsigned kill = on ? 0u : 0x80000000u;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            float* dst = dst0 + stage * STAGE + jj * 8 * KD;
            const int off = (int)((unsigned)(voff[jj] + ld_off) | kill);
            if (isA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, dst, 16, off, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rU, dst, 16, off, 0, 0, 0);
        }
        const bool wrap = ld_ch + 1 >= nch;
        ld_off += wrap ? xi_bytes - (nch - 1) * KD * 4 : KD * 4;
        ld_ch = wrap ? 0 : ld_ch + 1;
    };

    // two accumulators for M_xi (even / odd k steps): consecutive MFMAs of the wave are independent, their sum is folded
    f32x16 accm, accm2, accy[NY];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accm[r] = 0.f;
        accm2[r] = 0.f;
#pragma unroll
        for (int p = 0; p < NY; ++p) accy[p][r] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) dma(s, s < total);
    const int ra_ = wm * 32 + lrow, rb_ = wn * 32 + lrow;
    const int sa = (ra_ >> 1) & 7, sb = (rb_ >> 1) & 7;
    int xi = 0, ch = 0, stage = 0;
    for (int it = 0; it < total; ++it) {
        // this wave's pieces of chunk `it` have landed when at most the (NSTAGE - 2) younger chunks are outstanding
        if (NSTAGE == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nstage = (stage == 0) ? NSTAGE - 1 : stage - 1;   // == (it + NSTAGE - 1) % NSTAGE: read last in iteration it - 1
        dma(nstage, it + NSTAGE - 1 < total);
        const float* As = smem + stage * STAGE;
        const float* Bs = As + 64 * KD;
        f32x4 af[KD / 8], bf[KD / 8];                              // all eight fragment reads first: none waits behind an MFMA
#pragma unroll
        for (int kk = 0; kk < KD / 8; ++kk) {
            af[kk] = *reinterpret_cast<const f32x4*>(&As[ra_ * KD + (((kk * 2 + lh) ^ sa) << 2)]);
            bf[kk] = *reinterpret_cast<const f32x4*>(&Bs[rb_ * KD + (((kk * 2 + lh) ^ sb) << 2)]);
        }
        __builtin_amdgcn_sched_barrier(0);                         // (hipcc otherwise sinks each read pair in front of its MFMAs)
#pragma unroll
        for (int kk = 0; kk < KD / 8; ++kk)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                accm = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][e], bf[kk][e], accm, 0, 0, 0);
                accm2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][e + 1], bf[kk][e + 1], accm2, 0, 0, 0);
            }
        if (ch == nch - 1) {
            const int xa = xi / NP, xb = xi - xa * NP;
            float ca[MO], cb[MO];
#pragma unroll
            for (int k = 0; k < MO; ++k) {
                ca[k] = s1_at<MO>(k, xa);
                cb[k] = s1_at<MO>(k, xb);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float mv = accm[r] + accm2[r];
                accm2[r] = 0.f;
#pragma unroll
                for (int al = 0; al < MO; ++al)
#pragma unroll
                    for (int be = 0; be < MO; ++be) accy[al * MO + be][r] += (ca[al] * cb[be]) * mv;
                accm[r] = 0.f;
            }
        }
        const bool wrap = ch + 1 >= nch;
        ch = wrap ? 0 : ch + 1;
        xi = wrap ? xi + 1 : xi;
        stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the killed tail DMAs (zeros into unused stages) drain before exit

    const int col = n0 + wn * 32 + lrow;
    const float bv = (bias != nullptr && col < Co) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = m0 + wm * 32 + row;
        if (m < T && col < Co) {
            const int n = m / (TH * TW);
            const int rem = m - n * (TH * TW);
            const int ti = rem / TW, tj = rem - ti * TW;
#pragma unroll
            for (int p = 0; p < NY; ++p) {
                const int y = MO * ti + p / MO, x = MO * tj + p % MO;
                if (y < Hout && x < Wout) out[((long)(n * Hout + y) * Wout + x) * ld_out + col] = act_epi(accy[p][r] + bv, act);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layer, F(4x4, 2x2):  dW[kh][kw][a][b] = sum_pix dy[pix][a] * x[pix + (kh-1, kw-1)][b]  is, per
// 2x2 tile of dy and its 5x5 window of x, the 4x4 correlation of the window with the tile:
//        dW = A4^T [ sum_tiles (G2 dy G2^T) (.) (B^T x B) ] A4        (same points, same B^T as above)
//   k_wino_dy          DY[xi][tile][a] = (G2 dy G2^T)[xi]            (zero beyond the last row / column)
//   k_wino_v           V[xi][tile][b]                                 (the forward's input transform)
//   k_wino_wgrad_gemm  S[slice][xi][a][b] = sum_{tile in slice} DY[xi][tile][a] * V[xi][tile][b]     (MFMA, split over tiles)
//   k_wino_wgrad_out   dP[kh*4+kw][a][b] = sum_slice sum_xi A4T[kh][xi_i] A4T[kw][xi_j] S[slice][xi][a][b]   (fixed order)
__device__ __constant__ float c_G2[5][2] = {{0.5f, 0.f}, {-0.5f, -0.5f}, {-1.f / 6, 1.f / 6}, {1.f / 6, 1.f / 3}, {0.f, 1.f}};
__device__ __constant__ float c_A4T[4][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 2, 0}, {0, 1, 1, 4, 0}, {0, 1, -1, 8, 1}};

// F(4x4, 3x3): 3x3 tiles of dy against 6x6 windows of x on the six points of the forward's F(3x3,4x4) (0, 1, -1, 1/2, -2, inf: the
// same B^T, i.e. the same k_wino_v<3>): 36 instead of 144 multiplies per tile and channel pair, 1.56x fewer than F(4x4,2x2), transformed
// operands 0.64x the size; simulated fp32 error 3e-6 (F(4x4,2x2): 4e-6).  G rows = (1, p, p^2) / N_p with the Lagrange denominators of c_G34.
__device__ __constant__ float c_G43[6][3] = {{1.f, 0.f, 0.f},           {1.f / 3, 1.f / 3, 1.f / 3},       {-1.f / 3, 1.f / 3, -1.f / 3},
                                             {-16.f / 15, -8.f / 15, -4.f / 15}, {1.f / 15, -2.f / 15, 4.f / 15}, {0.f, 0.f, 1.f}};
__device__ __constant__ float c_A4T6[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, -2, 0}, {0, 1, 1, 0.25f, 4, 0}, {0, 1, -1, 0.125f, -8, 1}};
template <int R> __device__ __forceinline__ float wg_g(int a, int u) {
    if constexpr (R == 2) return c_G2[a][u];
    else return c_G43[a][u];
}
template <int R> __device__ __forceinline__ float wg_at(int k, int i) {
    if constexpr (R == 2) return c_A4T[k][i];
    else return c_A4T6[k][i];
}

template <int R>      // R x R tiles of dy (2: F(4x4,2x2), 3: F(4x4,3x3)); NP = R + 3 points
__global__ __launch_bounds__(256) void k_wino_dy(const float* __restrict__ dy, int ld, float* __restrict__ DY, int N, int Hs,
                                                 int Ws, int Ca, int TH, int TW) {
    constexpr int NP = R + 3;
    const int cq = Ca >> 2;
    const long T = (long)N * TH * TW;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= T * cq) return;
    const int c0 = (int)(idx % cq) << 2;
    const long tile = idx / cq;
    const int n = (int)(tile / (TH * TW));
    const int rem = (int)(tile - (long)n * TH * TW);
    const int ti = rem / TW, tj = rem - ti * TW;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[R][R];
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int v = 0; v < R; ++v) {
            const int y = R * ti + u, x = R * tj + v;
            d[u][v] = (y < Hs && x < Ws) ? *reinterpret_cast<const f32x4*>(dy + ((long)(n * Hs + y) * Ws + x) * ld + c0) : z;
        }
    f32x4 t[NP][R];
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int v = 0; v < R; ++v) {
            f32x4 sum = wg_g<R>(a, 0) * d[0][v];
#pragma unroll
            for (int u = 1; u < R; ++u) sum += wg_g<R>(a, u) * d[u][v];
            t[a][v] = sum;
        }
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            f32x4 sum = t[a][0] * wg_g<R>(b, 0);
#pragma unroll
            for (int v = 1; v < R; ++v) sum += t[a][v] * wg_g<R>(b, v);
            *reinterpret_cast<f32x4*>(DY + ((long)(a * NP + b) * T + tile) * Ca + c0) = sum;
        }
}

// 1-D grid of tilesA * tilesB * NX * slices workgroups; work index = ((slice * NX + xi) * tilesA + tile_a) * tilesB + tile_b,
// each XCD a contiguous run of it: the tilesA * tilesB workgroups that share the DY[xi] / V[xi] slabs of one K slice run
// together on one XCD.  Both operands are K-major (K = tiles): rows of Ca resp. Cb floats
template <int MR, int NR, int WM, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wino_wgrad_gemm(const float* __restrict__ DY, const float* __restrict__ V,
                                                         float* __restrict__ S, int T, int Ca, int Cb, int chunks_per_slice,
                                                         int tilesA, int tilesB, int NX, int dy_bytes, int v_bytes) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int AQ = BM / 4, AROWS = 256 / AQ, AI = KC / AROWS;
    constexpr int BQ = BN / 4, BROWS = 256 / BQ, BI = KC / BROWS;
    __shared__ __attribute__((aligned(16))) float smem[KC * LDA + KC * LDB];
    float* As = smem;
    float* Bs = smem + KC * LDA;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)DY, 0, dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, v_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = pg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_b = w % tilesB;
    w /= tilesB;
    const int tile_a = w % tilesA;
    w /= tilesA;
    const int xi = w % NX, slice = w / NX;
    const int m0 = tile_a * BM, n0 = tile_b * BN;
    const int nchunks = (T + KC - 1) / KC;
    const int c_begin = slice * chunks_per_slice;
    const int c_end = min(nchunks, c_begin + chunks_per_slice);
    const int aq = tid % AQ, arow0 = tid / AQ;
    const int bq = tid % BQ, brow0 = tid / BQ;
    const bool a_in = m0 + aq * 4 < Ca, b_in = n0 + bq * 4 < Cb;
    const int a_base = xi * T * Ca + m0 + aq * 4, b_base = xi * T * Cb + n0 + bq * 4;

    f32x4 ra[AI], rb[BI];
    auto issue_loads = [&](int c, bool on) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int k = c * KC + arow0 + AROWS * i;
            ra[i] = bload4(rA, voff(a_base + k * Ca, on && a_in && k < T));
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int k = c * KC + brow0 + BROWS * i;
            rb[i] = bload4(rB, voff(b_base + k * Cb, on && b_in && k < T));
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
        issue_loads(c_begin, true);
        store_chunk();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = (c + 1 < c_end);
        issue_loads(c + 1, more);
        __builtin_amdgcn_sched_barrier(0x386);
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
                for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }
    float* o = S + ((long)slice * NX + xi) * Ca * Cb;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int a = m0 + (wm * MR + i) * 32 + row;
                if (a < Ca && col < Cb) o[(long)a * Cb + col] = acc[i][j][r];
            }
        }
}

// one thread per (a, b): sums the slices in order, then the NP^2 -> 16 output transform
template <int R>
__global__ void k_wino_wgrad_out(const float* __restrict__ S, int slices, float* __restrict__ dP, int Ca, int Cb) {
    constexpr int NP = R + 3;
    const long ab = (long)Ca * Cb;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= ab) return;
    float m[NP][NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            float v = 0.f;
            for (int s = 0; s < slices; ++s) v += S[((long)s * NP * NP + i * NP + j) * ab + idx];
            m[i][j] = v;
        }
    float t[4][NP];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) v += wg_at<R>(k, i) * m[i][j];
            t[k][j] = v;
        }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < NP; ++j) v += t[k][j] * wg_at<R>(l, j);
            dP[(long)(k * 4 + l) * ab + idx] = v;
        }
}

// ------------------------------------------------------------------------------------------------------------------------
// Stride-2 layers, polyphase Winograd F(MO x MO, 2x2), MO = 3 (points 0, 1, -1, inf: 16 instead of 36 multiplies per tile,
// phase and channel pair; error ~3e-6, the level of the stride-1 path) or MO = 4 (points 0, 1, -1/2, -2, inf: 25 instead
// of 64, error ~6e-6; experiment switch, the size heuristics are tuned for MO = 3).
//
// big -> small (Conv2d forward, ConvTranspose2d data gradient):
//   small[p][q] = sum_{r,s in {0,1}} sum_{u,v in {0,1}} X_rs[p+u][q+v] * w[2u+r][2v+s],   X_rs[i][j] = big[2i+r-1][2j+s-1]
// -- four 2x2-kernel stride-1 correlations over the four parity phases of the input, one GEMM with K = 4*Cb:
//   k_wino2_u    U[xi][a][ph*Cb + b] = (G g_ph G^T)[xi]
//   k_wino2_v    V[xi][tile][ph*Cb + b] = (B^T X_ph B)[xi]
//   k_wino_bgemm M[xi][tile][a] = sum_k V[xi][tile][k] U[xi][a][k]          (NP^2 row GEMMs, grid.z = xi)
//   k_wino2_out  small[MO*ti+al][MO*tj+be][a] = act(bias + sum_xi AT[al][xi_i] AT[be][xi_j] M[xi][tile][a])
// small -> big (ConvTranspose2d forward, Conv2d data gradient): each output parity class (r, s) of `big` is a 2x2-kernel
// stride-1 correlation of `small`:
//   big[2i+r][2j+s] = sum_{t,t' in {0,1}} small[i+r-1+t][j+s-1+t'] * w[kh(r,t)][kw(s,t')],  kh(0,t) = 3-2t, kh(1,t) = 2-2t
// class r needs the window small[i+r-1 .. ]: with windows starting at MO*t - 1 all four classes share ONE transformed
// input per tile -- class r then produces the outputs i = MO*t - r + al, i.e. the classes' output tiles are staggered by
// one pixel instead of their windows -- and the four classes are the column blocks of one GEMM with N = 4*Cb:
//   k_wino2c_u   U[xi][cls*Cb + b][a] = (G g_class G^T)[xi]        k_wino2c_v   V[xi][tile][a] = (B^T d B)[xi]
//   k_wino_bgemm M[xi][tile][cls*Cb + b] = sum_a V[xi][tile][a] U[xi][cls*Cb + b][a]
//   k_wino2c_out big[2(MO*ti-r+al)+r][2(MO*tj-s+be)+s][b] = act(bias + sum_xi AT[al][xi_i] AT[be][xi_j] M[xi][tile][cls*Cb+b])
// Pays where the channel counts are large against the tile count (the transformed tensors make a round trip through HBM).
__device__ __constant__ float c_BT3[4][4] = {{-1, 0, 1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, -1, 0, 1}};
__device__ __constant__ float c_G3[4][2] = {{-1.f, 0.f}, {0.5f, 0.5f}, {0.5f, -0.5f}, {0.f, 1.f}};
__device__ __constant__ float c_A3T[3][4] = {{1, 1, 1, 0}, {0, 1, -1, 0}, {0, 1, 1, 1}};
// MO = 4 on the points {0, 1, -1/2, -2, inf}: the most accurate of the 5-point sets drawn from {0, +-1, +-1/2, +-2}
// (simulated fp32 error 6.4e-6; {0, 1, -1, 2}: 1.1e-5; F(3x3,2x2) on {0, 1, -1}: 2.5e-6)
__device__ __constant__ float c_BT4p[5][5] = {{-1, -1.5f, 1.5f, 1, 0}, {0, 1, 2.5f, 1, 0}, {0, -2, 1, 1, 0}, {0, -0.5f, -0.5f, 1, 0},
                                              {0, -1, -1.5f, 1.5f, 1}};
__device__ __constant__ float c_G2p[5][2] = {{-1.f, 0.f}, {2.f / 9, 2.f / 9}, {8.f / 9, -4.f / 9}, {-1.f / 9, 2.f / 9}, {0.f, 1.f}};
__device__ __constant__ float c_A4Tp[4][5] = {{1, 1, 1, 1, 0}, {0, 1, -0.5f, -2, 0}, {0, 1, 0.25f, 4, 0}, {0, 1, -0.125f, -8, 1}};
template <int MO> __device__ __forceinline__ float w_bt(int a, int i) { return MO == 4 ? c_BT4p[a][i] : c_BT3[a][i]; }
template <int MO> __device__ __forceinline__ float w_g(int a, int u) { return MO == 4 ? c_G2p[a][u] : c_G3[a][u]; }
template <int MO> __device__ __forceinline__ float w_at(int k, int i) { return MO == 4 ? c_A4Tp[k][i] : c_A3T[k][i]; }

template <int MO>
__device__ __forceinline__ void wino2_u_body(const float* __restrict__ P, float* __restrict__ U, int Ca, int Cb, long block) {
    constexpr int NP = MO + 1;
    const long idx = block * 256 + threadIdx.x;
    if (idx >= (long)Ca * Cb) return;
    const int b = (int)(idx % Cb), a = (int)(idx / Cb);
    float w[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 4; ++l) w[k][l] = P[((long)(k * 4 + l) * Ca + a) * Cb + b];
    const int K = 4 * Cb;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int sph = 0; sph < 2; ++sph) {
            float t[NP][2];
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int v = 0; v < 2; ++v) t[i][v] = w_g<MO>(i, 0) * w[r][2 * v + sph] + w_g<MO>(i, 1) * w[2 + r][2 * v + sph];
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    U[((long)(i * NP + j) * Ca + a) * K + (r * 2 + sph) * Cb + b] = t[i][0] * w_g<MO>(j, 0) + t[i][1] * w_g<MO>(j, 1);
        }
}
template <int MO>
__global__ __launch_bounds__(256) void k_wino2_u(const float* __restrict__ P, float* __restrict__ U, int Ca, int Cb) {
    wino2_u_body<MO>(P, U, Ca, Cb, blockIdx.x);
}

// shared by both directions: window origin (oy, ox) + step `st` pixels between window entries; writes V[(xi*nb + bi)][tile][c]
template <int MO>
__device__ __forceinline__ void wino2_v_tile(const float* __restrict__ src, int ld, int n, int H, int W, int oy, int ox, int st,
                                             int c0, float* __restrict__ V, long zstride, long vbase) {
    constexpr int NP = MO + 1;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[NP][NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int y = oy + st * i, x = ox + st * j;
            const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            d[i][j] = ok ? *reinterpret_cast<const f32x4*>(src + ((long)(n * H + y) * W + x) * ld + c0) : z;
        }
    f32x4 t[NP][NP];
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            f32x4 v = z;
#pragma unroll
            for (int i = 0; i < NP; ++i) v += w_bt<MO>(a, i) * d[i][j];
            t[a][j] = v;
        }
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            f32x4 v = z;
#pragma unroll
            for (int j = 0; j < NP; ++j) v += t[a][j] * w_bt<MO>(b, j);
            *reinterpret_cast<f32x4*>(V + (long)(a * NP + b) * zstride + vbase) = v;
        }
}

template <int MO>
__global__ __launch_bounds__(256) void k_wino2_v(const float* __restrict__ big, int ld, float* __restrict__ V, int N, int Hb,
                                                 int Wb, int Cb, int TH, int TW) {
    const int cq = Cb >> 2;
    const long T = (long)N * TH * TW;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= T * 4 * cq) return;
    const int c0 = (int)(idx % cq) << 2;
    long rr = idx / cq;
    const int ph = (int)(rr & 3);
    const long tile = rr >> 2;
    const int n = (int)(tile / (TH * TW));
    const int rem = (int)(tile - (long)n * TH * TW);
    const int ti = rem / TW, tj = rem - ti * TW;
    const long K = 4L * Cb;
    // X_ph[MO*ti + i][MO*tj + j] = big[2*(MO*ti + i) + r - 1][2*(MO*tj + j) + s - 1]
    wino2_v_tile<MO>(big, ld, n, Hb, Wb, 2 * MO * ti + (ph >> 1) - 1, 2 * MO * tj + (ph & 1) - 1, 2, c0, V, T * K,
                     tile * K + ph * Cb + c0);
}

#ifdef PG_TRACE_R
// diagnostics build only (make trace): phase stamps of k_wino_bgemm, as in conv_bf16.hip
__device__ unsigned long long* pg_trace_buf_w = nullptr;
#endif
// NZ independent row GEMMs: C[z][m][n] = sum_k A[z][m][k] * B[z][n][k]; rows of A and B are K contiguous floats (K % 32 == 0)
template <int MR, int NR, int WM, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wino_bgemm(const float* __restrict__ A, const float* __restrict__ B,
                                                    float* __restrict__ C, int Mrows, int Ncols, int K, int a_bytes,
                                                    int b_bytes, int tiles_m, int tiles_n) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32, AI = BM / 32, BI = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    float* As = smem;
    float* Bs = smem + BM * LDK;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    // 1-D grid, work index = (z * tiles_m + tm) * tiles_n + tn, each XCD a contiguous run: the tiles_n workgroups that read
    // one 128-row slab of A[z] are neighbours on one XCD, and an XCD stays on one batch z (one B[z]) for a long stretch
    int w = pg_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int m0 = (w % tiles_m) * BM, z = w / tiles_m;
    const int nch = K / KC;
    const int kq = tid & 7, r0 = tid >> 3;
    int a_off[AI], b_off[BI];
    bool a_ok[AI], b_ok[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < Mrows;
        a_off[i] = (z * Mrows + min(m, Mrows - 1)) * K + kq * 4;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_ok[i] = n < Ncols;
        b_off[i] = (z * Ncols + min(n, Ncols - 1)) * K + kq * 4;
    }
    f32x4 ra[AI], rb[BI];
    auto issue_loads = [&](int c, bool on) {
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[i] = bload4(rA, voff(a_off[i] + c * KC, on && a_ok[i]));
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[i] = bload4(rB, voff(b_off[i] + c * KC, on && b_ok[i]));
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
#ifdef PG_TRACE_R
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tr_w = 0, tr_m = 0;
#endif
    issue_loads(0, true);
    store_chunk();
    __syncthreads();
#ifdef PG_TRACE_R
    const unsigned long long tr_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int c = 0; c < nch; ++c) {
        const bool more = c + 1 < nch;
#ifdef PG_TRACE_R
        const unsigned long long tr_b = __builtin_amdgcn_s_memrealtime();
#endif
        issue_loads(c + 1, more);
        __builtin_amdgcn_sched_barrier(0x386);
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
#ifdef PG_TRACE_R
        const unsigned long long tr_c = __builtin_amdgcn_s_memrealtime();
        tr_m += tr_c - tr_b;
#endif
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
#ifdef PG_TRACE_R
        tr_w += __builtin_amdgcn_s_memrealtime() - tr_c;
#endif
    }
#ifdef PG_TRACE_R
    const unsigned long long tr_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    float* o = C + (long)z * Mrows * Ncols;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + (wm * MR + i) * 32 + row;
                if (m < Mrows && col < Ncols) o[(long)m * Ncols + col] = acc[i][j][r];
            }
        }
#ifdef PG_TRACE_R
    if (pg_trace_buf_w != nullptr && lane == 0 && blockIdx.x < 16384) {
        unsigned long long* const ob = pg_trace_buf_w + ((size_t)blockIdx.x * 4 + wave) * 8;
        ob[0] = tr_t0;
        ob[1] = tr_t1;
        ob[2] = tr_w;            // barriers + LDS staging between chunks
        ob[3] = tr_m;            // issuing a chunk's loads, LDS reads and MFMAs
        ob[4] = tr_t2;
        ob[5] = __builtin_amdgcn_s_memrealtime();
        ob[6] = 0;
        ob[7] = (unsigned long long)nch;
    }
#endif
}

// ---- split-bf16 form of the same batched row GEMM (the default; PG_TUNE_S3_OFF selects the fp32-MFMA kernel above) ----
// fp32 products on the bf16 matrix pipe: every operand value a is split, while it is staged into LDS, into three bf16 pieces
// a = a1 + a2 + a3 (a1 = RNE_bf16(a), a2 = RNE_bf16(a - a1), a3 = a - a1 - a2: exact, 8 + 8 + 8 significand bits cover fp32's 24), and the six
// products a_i b_j with i + j <= 4 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (a product of two bf16 values is exact in fp32; the
// dropped terms a2 b3 + a3 b2 + a3 b3 are <= 2^-25 |a b|, below the rounding of the fp32 accumulation itself).  Six MFMAs of 32 cycles per 16 k
// instead of eight fp32 MFMAs of 64: 2.67x fewer matrix-pipe cycles for fp32-grade results (measured per kernel against float64: the same
// 1e-6-level errors as the fp32-MFMA kernel, tests/test_bench_layers_gpu.py).  Operands stay fp32 in HBM: the split costs VALU work per
// staged value, not bytes.  Inf / NaN operands: a - RNE_bf16(a) is NaN for an infinite a, so an infinity in an operand gives NaN where
// the fp32 kernel gives +-inf or NaN -- both poison the step.
typedef __bf16 s3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s3_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void s3_split(const f32x4 v, s3_bf16x4& h, s3_bf16x4& m, s3_bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 a = (__bf16)v[e];
        const float r1 = v[e] - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        h[e] = a;
        m[e] = b;
        l[e] = (__bf16)r2;
    }
}

// Structure: K chunks of 16, two LDS buffers, ONE barrier per chunk.  While the MFMAs of chunk c run out of buffer c & 1, the same wave
// splits chunk c + 1 (loaded two chunks earlier into one of two register sets) and writes it into the other buffer: the split's VALU work and
// the LDS writes sit between the MFMAs of the wave's own instruction stream instead of in a phase of their own (a two-phase form with
// 32-wide chunks measured 0.65 us of MFMAs and 1.6-1.9 us of waiting + splitting + writing per chunk: EXPERIMENTS.md, round 6).
// LDS row = [a1: 16 k | a2: 16 k | a3: 16 k | pad] = 112 bytes = 7 x 16: conflict-free ds_read_b128 fragments of 8 consecutive k.
constexpr int S3_KC = 16, S3_LDR = 3 * S3_KC + 8;
// PIPE: the fragments double-buffered in REGISTERS -- the MFMAs of chunk c run on fragments read during chunk c - 1, while this iteration reads
// chunk c + 1's fragments and splits / stores chunk c + 2, so nothing an MFMA waits for was issued in its own iteration -- with the instruction
// mix laid out by sched_group_barrier (one MFMA, a load / LDS read, a slice of the split, an LDS store).  Same products in the same order per
// accumulator: bit-identical to the plain form (tools/s3_probe.hip modes 7 / 8: 4-6 % faster back to back, not inside the step: off by default,
// PATCHGAN_S3_PIPE=1 selects it).
template <int MR, int NR, int WM, int WN, int WPE, int VAR = 1, bool PIPE = false>      // VAR: order of the six products (1 = largest first, the shipped one; others: PATCHGAN_S3_VAR, experiment)
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wino_bgemm_s3(const float* __restrict__ A, const float* __restrict__ B,
                                                    float* __restrict__ C, int Mrows, int Ncols, int K, int a_bytes,
                                                    int b_bytes, int tiles_m, int tiles_n) {
    // (WM x WN waves; four staging lanes per 16-wide row piece, RP rows per pass.  <2,2,4,2>: a 256 x 128 tile on EIGHT waves -- the B rows
    //  staged once for twice the MFMAs, 86 KB of LDS, one workgroup per CU)
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32, RP = WM * WN * 16, AI = BM / RP, BI = BN / RP, BUF = (BM + BN) * S3_LDR;
    static_assert(AI * RP == BM && BI * RP == BN, "whole staging passes");
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * BUF];
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = pg_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int m0 = (w % tiles_m) * BM, z = w / tiles_m;
    const int nch = K / S3_KC;
    const int kq = tid & 3, r0 = tid >> 2;      // four lanes per 16-wide row piece, 64 rows per pass
    int a_off[AI], b_off[BI];
    bool a_ok[AI], b_ok[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + RP * i;
        a_ok[i] = m < Mrows;
        a_off[i] = (z * Mrows + min(m, Mrows - 1)) * K + kq * 4;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int n = n0 + r0 + RP * i;
        b_ok[i] = n < Ncols;
        b_off[i] = (z * Ncols + min(n, Ncols - 1)) * K + kq * 4;
    }
    auto issue_loads = [&](f32x4 (&ra)[AI], f32x4 (&rb)[BI], int c) {
        const bool on = c < nch;
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[i] = bload4(rA, voff(a_off[i] + c * S3_KC, on && a_ok[i]));
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[i] = bload4(rB, voff(b_off[i] + c * S3_KC, on && b_ok[i]));
    };
    auto stage = [&](const f32x4 (&ra)[AI], const f32x4 (&rb)[BI], __bf16* buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            s3_bf16x4 h, m, l;
            s3_split(ra[i], h, m, l);
            __bf16* row = &buf[(r0 + RP * i) * S3_LDR + kq * 4];
            *reinterpret_cast<s3_bf16x4*>(row) = h;
            *reinterpret_cast<s3_bf16x4*>(row + S3_KC) = m;
            *reinterpret_cast<s3_bf16x4*>(row + 2 * S3_KC) = l;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            s3_bf16x4 h, m, l;
            s3_split(rb[i], h, m, l);
            __bf16* row = &buf[(BM + r0 + RP * i) * S3_LDR + kq * 4];
            *reinterpret_cast<s3_bf16x4*>(row) = h;
            *reinterpret_cast<s3_bf16x4*>(row + S3_KC) = m;
            *reinterpret_cast<s3_bf16x4*>(row + 2 * S3_KC) = l;
        }
    };
    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](const __bf16* buf) {
        s3_bf16x8 bf[NR][3], af[MR][3];
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bf[j][p] = *reinterpret_cast<const s3_bf16x8*>(&buf[(BM + (wn * NR + j) * 32 + lrow) * S3_LDR + p * S3_KC + lh * 8]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                af[i][p] = *reinterpret_cast<const s3_bf16x8*>(&buf[((wm * MR + i) * 32 + lrow) * S3_LDR + p * S3_KC + lh * 8]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
#define S3_MM(pa, pb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0)
                if constexpr (VAR == 1) {          // largest terms first
                    S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
                } else if constexpr (VAR == 2) {   // eight products, smallest first
                    S3_MM(2, 1); S3_MM(1, 2); S3_MM(2, 0); S3_MM(0, 2); S3_MM(1, 1); S3_MM(1, 0); S3_MM(0, 1); S3_MM(0, 0);
                } else if constexpr (VAR == 3) {   // a-major order
                    S3_MM(2, 0); S3_MM(1, 1); S3_MM(1, 0); S3_MM(0, 2); S3_MM(0, 1); S3_MM(0, 0);
                } else {                           // smallest terms first
                    S3_MM(2, 0); S3_MM(0, 2); S3_MM(1, 1); S3_MM(1, 0); S3_MM(0, 1); S3_MM(0, 0);
                }
#undef S3_MM
            }
    };
    __bf16* const buf0 = smem;
    __bf16* const buf1 = smem + BUF;
    f32x4 ra0[AI], rb0[BI], ra1[AI], rb1[BI];
    if constexpr (PIPE) {
        static_assert(VAR == 1, "the pipelined form ships the largest-first order only");
        s3_bf16x8 fa0[MR][3], fb0[NR][3], fa1[MR][3], fb1[NR][3];
        auto rd = [&](s3_bf16x8 (&fa)[MR][3], s3_bf16x8 (&fb)[NR][3], const __bf16* buf) {
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fb[j][p] = *reinterpret_cast<const s3_bf16x8*>(&buf[(BM + (wn * NR + j) * 32 + lrow) * S3_LDR + p * S3_KC + lh * 8]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fa[i][p] = *reinterpret_cast<const s3_bf16x8*>(&buf[((wm * MR + i) * 32 + lrow) * S3_LDR + p * S3_KC + lh * 8]);
        };
        auto mm = [&](const s3_bf16x8 (&fa)[MR][3], const s3_bf16x8 (&fb)[NR][3]) {      // per accumulator the order of VAR 1
#define S3_MM(pa, pb)                                                                       \
    _Pragma("unroll") for (int i = 0; i < MR; ++i) _Pragma("unroll") for (int j = 0; j < NR; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa], fb[j][pb], acc[i][j], 0, 0, 0)
            S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
#undef S3_MM
        };
        auto mix = [&]() {
            constexpr int NM = 6 * MR * NR, NRD = 3 * (MR + NR), NWR = 3 * (AI + BI), NLD = AI + BI;
#pragma unroll
            for (int t = 0; t < NM; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
                if (t < NLD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // the global loads first
                if (t < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // the next chunk's fragments early
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                      // a slice of the split
                if (t >= NM - NWR) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // the staged pieces late
            }
        };
        issue_loads(ra0, rb0, 0);
        issue_loads(ra1, rb1, 1);
        stage(ra0, rb0, buf0);
        issue_loads(ra0, rb0, 2);
        __syncthreads();
        rd(fa0, fb0, buf0);
        stage(ra1, rb1, buf1);
        issue_loads(ra1, rb1, 3);
        __syncthreads();
        // top of step c: fa0 / fb0 = chunk c; buf1 = chunk c + 1; set 0 = chunk c + 2, set 1 = chunk c + 3 (in flight); buf0 is free
        for (int c = 0; c < nch; c += 2) {
            mm(fa0, fb0);
            rd(fa1, fb1, buf1);
            stage(ra0, rb0, buf0);
            issue_loads(ra0, rb0, c + 4);
            mix();
            __syncthreads();
            mm(fa1, fb1);
            rd(fa0, fb0, buf0);
            stage(ra1, rb1, buf1);
            issue_loads(ra1, rb1, c + 5);
            mix();
            __syncthreads();
        }
    } else {
    issue_loads(ra0, rb0, 0);
    issue_loads(ra1, rb1, 1);
    stage(ra0, rb0, buf0);
    issue_loads(ra0, rb0, 2);
    __syncthreads();
    // invariant at the top of step c: buf[c & 1] = chunk c; set (c + 1) & 1 = chunk c + 1; set c & 1 = chunk c + 2 (both in flight / landed)
    // (K % 32 == 0: an even number of chunks; loads beyond the last chunk are masked to zero and staged into a buffer nobody reads, so the
    //  body has no branch and the compiler interleaves the split with the MFMAs)
    for (int c = 0; c < nch; c += 2) {
        stage(ra1, rb1, buf1);
        compute(buf0);
        issue_loads(ra1, rb1, c + 3);
        __syncthreads();
        stage(ra0, rb0, buf0);
        compute(buf1);
        issue_loads(ra0, rb0, c + 4);
        __syncthreads();
    }
    }
    float* o = C + (long)z * Mrows * Ncols;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + (wm * MR + i) * 32 + row;
                if (m < Mrows && col < Ncols) o[(long)m * Ncols + col] = acc[i][j][r];
            }
        }
}

// ---- split-bf16 form of the weight-gradient GEMM (k_wino_wgrad_gemm above): S[slice][xi][a][b] = sum_{tile in slice} DY[xi][tile][a] V[xi][tile][b] ----
// Both operands are K-major here (K = the Winograd tile index; a row of DY / V holds one tile's channels), while a bf16 MFMA operand is 8
// consecutive K of one channel: the transposition happens in registers on the way to LDS.  A staging thread owns a 4 (tiles) x 4 (channels)
// block -- four 16-byte loads from four consecutive tile rows -- and writes, per channel, the three bf16 pieces of its 4 consecutive tiles:
// the same twelve 8-byte LDS stores per 16 values as k_wino_bgemm_s3, the same [channel][piece][16 k] rows, the same fragment reads and the
// same six products per fp32 product (largest first).  KS k-steps of 16 tiles per barrier: 1 for the 128 x 128 tile (256 staging threads =
// 128 + 128 channel rows x 4 tile groups), 2 for the 64 x 64 tile.  Tiles beyond T (the last chunk, the slice's end) are masked to zero.
template <int MR, int NR, int WM, int WN, int KS, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wino_wgrad_gemm_s3(const float* __restrict__ DY, const float* __restrict__ V,
                                                         float* __restrict__ S, int T, int Ca, int Cb, int chunks_per_slice,
                                                         int tilesA, int tilesB, int NX, int dy_bytes, int v_bytes) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32, KCH = 16 * KS, ROW = KS * 48 + 8, BUF = (BM + BN) * ROW;
    static_assert((BM + BN) * KS == 256, "one 4 x 4 block per staging thread");
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = pg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_b = w % tilesB;
    w /= tilesB;
    const int tile_a = w % tilesA;
    w /= tilesA;
    const int xi = w % NX, slice = w / NX;
    const int m0 = tile_a * BM, n0 = tile_b * BN;
    // chunks_per_slice counts 32-tile chunks (the fp32 kernel's KC) so that both kernels cut K into the same slices
    const int k_begin = slice * chunks_per_slice * KC, k_end = min(T, k_begin + chunks_per_slice * KC);
    const int nch = (max(k_end - k_begin, 0) + KCH - 1) / KCH;
    // staging role (wave-uniform, so that the buffer descriptor stays in scalar registers): items 0 .. BM * KS - 1 stage DY, the rest V
    const bool is_a = __builtin_amdgcn_readfirstlane(tid >> 6) < BM * KS / 64;
    const int it = is_a ? tid : tid - BM * KS, BX = is_a ? BM : BN;
    const int kg = it & 3, mq = (it >> 2) % (BX / 4), ks = it / BX;
    const int ld = is_a ? Ca : Cb, c0 = (is_a ? m0 : n0) + mq * 4;
    const bool c_in = c0 < ld;
    const __amdgpu_buffer_rsrc_t rS = is_a ? __builtin_amdgcn_make_buffer_rsrc((void*)DY, 0, dy_bytes, 0x00020000)
                                           : __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, v_bytes, 0x00020000);
    const int base = xi * T * ld + c0;                           // + tile * ld
    const int krow = ks * 16 + kg * 4;                           // first of this thread's four tiles within a chunk
    const int wr_off = ((is_a ? 0 : BM) + mq * 4) * ROW + ks * 48 + kg * 4;
    auto issue_loads = [&](f32x4 (&r)[4], int c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k_begin + c * KCH + krow + j;
            r[j] = bload4(rS, voff(base + k * ld, c < nch && c_in && k < k_end));
        }
    };
    auto stage = [&](const f32x4 (&r)[4], __bf16* buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 v = {r[0][e], r[1][e], r[2][e], r[3][e]};          // channel c0 + e of four consecutive tiles
            s3_bf16x4 h, m, l;
            s3_split(v, h, m, l);
            __bf16* row = &buf[wr_off + e * ROW];
            *reinterpret_cast<s3_bf16x4*>(row) = h;
            *reinterpret_cast<s3_bf16x4*>(row + 16) = m;
            *reinterpret_cast<s3_bf16x4*>(row + 32) = l;
        }
    };
    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](const __bf16* buf) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            s3_bf16x8 bf[NR][3], af[MR][3];
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bf[j][p] = *reinterpret_cast<const s3_bf16x8*>(&buf[(BM + (wn * NR + j) * 32 + lrow) * ROW + s * 48 + p * 16 + lh * 8]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[i][p] = *reinterpret_cast<const s3_bf16x8*>(&buf[((wm * MR + i) * 32 + lrow) * ROW + s * 48 + p * 16 + lh * 8]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
#define S3_MM(pa, pb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0)
                    S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
#undef S3_MM
                }
        }
    };
    __bf16* const buf0 = smem;
    __bf16* const buf1 = smem + BUF;
    f32x4 r0[4], r1[4];
    issue_loads(r0, 0);
    issue_loads(r1, 1);
    stage(r0, buf0);
    issue_loads(r0, 2);
    __syncthreads();
    // as k_wino_bgemm_s3: buf[c & 1] = chunk c, the two register sets hold chunks c + 1 and c + 2; an odd chunk count runs one masked chunk
    for (int c = 0; c < nch; c += 2) {
        stage(r1, buf1);
        compute(buf0);
        issue_loads(r1, c + 3);
        __syncthreads();
        stage(r0, buf0);
        compute(buf1);
        issue_loads(r0, c + 4);
        __syncthreads();
    }
    float* o = S + ((long)slice * NX + xi) * Ca * Cb;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int col = n0 + (wn * NR + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int a = m0 + (wm * MR + i) * 32 + row;
                if (a < Ca && col < Cb) o[(long)a * Cb + col] = acc[i][j][r];
            }
        }
}

// ---- split-bf16 form of k_wino_gemm_row (F(3x3,4x4), output transform cut in two) ----
// The same work split -- a workgroup owns ONE ROW i of the 6 x 6 products of its tiles x channels and folds the column transform
// T_i[c] = sum_j M_ij AT[c][j] -- on the bf16 matrix pipe: 128 tiles x 64 channels per workgroup (a wave: 64 x 32 = two accumulator sets for M_ij
// + six for T_i), operands split into three bf16 pieces while they are staged (s3_split), 16-wide K chunks, two LDS buffers and two register
// sets of loads as in k_wino_bgemm_s3, one flattened (j, chunk) loop.  Ci % 32 == 0 (an even number of chunks per j).
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wino_gemm_row_s3(
    const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ Tout, int T, int Ci, int Co, int v_bytes, int u_bytes,
    int tiles_n) {
    constexpr int NP = 6, BM = 128, BN = 64, MR = 2, BUF = (BM + BN) * S3_LDR;
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * BUF];
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, u_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = pg_xcd_remap(blockIdx.x, gridDim.x);             // (work order as k_wino_gemm_row: an XCD stays on one row of products)
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int tiles_m = (T + BM - 1) / BM;
    const int xrow = w / tiles_m;
    const int m0 = (w % tiles_m) * BM;
    const int nch = Ci / S3_KC, total = NP * nch;
    const int kq = tid & 3, r0 = tid >> 2;
    int a_off[2], b_off;
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + r0 + 64 * i;
        a_ok[i] = m < T;
        a_off[i] = min(m, T - 1) * Ci + kq * 4;
    }
    {
        const int co = n0 + r0;
        b_off = (co < Co) ? co * Ci + kq * 4 : 0x10000000;
    }
    const int v_xi = T * Ci, u_xi = Co * Ci;
    int ld_j = 0, ld_ch = 0;
    auto issue_loads = [&](f32x4 (&ra)[2], f32x4& rb) {         // the next (j, chunk) in order; beyond the last one: masked
        const bool on = ld_j < NP;
        const int xi = xrow * NP + ld_j;
        const int av = xi * v_xi + ld_ch * S3_KC, bu = xi * u_xi + ld_ch * S3_KC;
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = bload4(rV, voff(a_off[i] + av, on && a_ok[i]));
        rb = bload4(rU, voff(b_off + bu, on));
        const bool wrap = ld_ch + 1 >= nch;
        ld_ch = wrap ? 0 : ld_ch + 1;
        ld_j = wrap ? ld_j + 1 : ld_j;
    };
    auto stage = [&](const f32x4 (&ra)[2], const f32x4& rb, __bf16* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            s3_bf16x4 h, m, l;
            s3_split(ra[i], h, m, l);
            __bf16* row = &buf[(r0 + 64 * i) * S3_LDR + kq * 4];
            *reinterpret_cast<s3_bf16x4*>(row) = h;
            *reinterpret_cast<s3_bf16x4*>(row + S3_KC) = m;
            *reinterpret_cast<s3_bf16x4*>(row + 2 * S3_KC) = l;
        }
        s3_bf16x4 h, m, l;
        s3_split(rb, h, m, l);
        __bf16* row = &buf[(BM + r0) * S3_LDR + kq * 4];
        *reinterpret_cast<s3_bf16x4*>(row) = h;
        *reinterpret_cast<s3_bf16x4*>(row + S3_KC) = m;
        *reinterpret_cast<s3_bf16x4*>(row + 2 * S3_KC) = l;
    };
    f32x16 accm[MR], acct[3][MR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accm[i][r] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) acct[c][i][r] = 0.f;
        }
    int j = 0, ch = 0;
    auto compute = [&](const __bf16* buf) {
        s3_bf16x8 bf[3], af[MR][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const s3_bf16x8*>(&buf[(BM + wn * 32 + lrow) * S3_LDR + p * S3_KC + lh * 8]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                af[i][p] = *reinterpret_cast<const s3_bf16x8*>(&buf[((wm * MR + i) * 32 + lrow) * S3_LDR + p * S3_KC + lh * 8]);
#pragma unroll
        for (int i = 0; i < MR; ++i) {
#define S3_MM(pa, pb) accm[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[pb], accm[i], 0, 0, 0)
            S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
#undef S3_MM
        }
        if (ch == nch - 1) {            // the last chunk of product (xrow, j): fold it into the three column-transformed sets
            const float c0 = c_AT34[0][j], c1 = c_AT34[1][j], c2 = c_AT34[2][j];
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mv = accm[i][r];
                    acct[0][i][r] += c0 * mv;
                    acct[1][i][r] += c1 * mv;
                    acct[2][i][r] += c2 * mv;
                    accm[i][r] = 0.f;
                }
        }
        const bool wrap = ch + 1 >= nch;
        ch = wrap ? 0 : ch + 1;
        j = wrap ? j + 1 : j;
    };
    __bf16* const buf0 = smem;
    __bf16* const buf1 = smem + BUF;
    f32x4 ra0[2], rb0, ra1[2], rb1;
    issue_loads(ra0, rb0);
    issue_loads(ra1, rb1);
    stage(ra0, rb0, buf0);
    issue_loads(ra0, rb0);
    __syncthreads();
    for (int it = 0; it < total; it += 2) {
        stage(ra1, rb1, buf1);
        compute(buf0);
        issue_loads(ra1, rb1);
        __syncthreads();
        stage(ra0, rb0, buf0);
        compute(buf1);
        issue_loads(ra0, rb0);
        __syncthreads();
    }
    const int col = n0 + wn * 32 + lrow;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int m = m0 + (wm * MR + i) * 32 + row;
            if (m < T && col < Co) {
#pragma unroll
                for (int c = 0; c < 3; ++c) Tout[((long)(xrow * 3 + c) * T + m) * Co + col] = acct[c][i][r];
            }
        }
}

// Multi-batch variant: a workgroup owns one (m, n) tile position and runs `zb` consecutive batches z through ONE flattened (z, chunk) loop: the
// loads of the next batch's first chunk are in flight during the last MFMAs of the current one and the tile stores are
// fire-and-forget, so the short K loops of these layers (4..32 chunks) do not pay a prologue and an epilogue each.
// Two waves per SIMD: without the attribute hipcc took 196 VGPRs + 64 AGPRs = one wave per SIMD (one workgroup per CU, nothing to overlap
// its barriers with); at two the same 196 registers fit twice (cfg2 step 9.07 -> 8.93 ms on one box); at three / four the kernel spills
// (116 / 260 bytes of scratch: 9.29 / 9.21 ms).
template <int MR, int NR, int WM, int WN, int WPE = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_wino_bgemm_mz(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int Mrows, int Ncols, int K, int zb, int a_bytes,
                                                       int b_bytes, int tiles_m, int tiles_n) {
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32, AI = BM / 32, BI = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    float* As = smem;
    float* Bs = smem + BM * LDK;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = pg_xcd_remap(blockIdx.x, gridDim.x);          // work order as in k_wino_bgemm, z = group of zb batches
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int m0 = (w % tiles_m) * BM, z0 = (w / tiles_m) * zb;
    const int nch = K / KC, total = zb * nch;
    const int kq = tid & 7, r0 = tid >> 3;
    int a_off[AI], b_off[BI];
    bool a_ok[AI], b_ok[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < Mrows;
        a_off[i] = (z0 * Mrows + min(m, Mrows - 1)) * K + kq * 4;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_ok[i] = n < Ncols;
        b_off[i] = (z0 * Ncols + min(n, Ncols - 1)) * K + kq * 4;
    }
    const int a_zs = Mrows * K, b_zs = Ncols * K;       // elements per batch
    f32x4 ra[AI], rb[BI];
    int ld_z = 0, ld_c = 0;                             // (batch, chunk) of the NEXT load
    auto issue_loads = [&](bool on) {
        const int ao = ld_z * a_zs + ld_c * KC, bo = ld_z * b_zs + ld_c * KC;
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[i] = bload4(rA, voff(a_off[i] + ao, on && a_ok[i]));
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[i] = bload4(rB, voff(b_off[i] + bo, on && b_ok[i]));
        const bool wrap = ld_c + 1 >= nch;
        ld_c = wrap ? 0 : ld_c + 1;
        ld_z = wrap ? ld_z + 1 : ld_z;
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
    issue_loads(true);
    store_chunk();
    __syncthreads();
    int z = 0, c = 0;
    for (int it = 0; it < total; ++it) {
        const bool more = it + 1 < total;
        issue_loads(more);
        __builtin_amdgcn_sched_barrier(0x386);
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
        if (c == nch - 1) {      // batch z0 + z is complete: store its tile, start the next from zero
            float* o = C + (long)(z0 + z) * Mrows * Ncols;
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int col = n0 + (wn * NR + j) * 32 + lrow;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int m = m0 + (wm * MR + i) * 32 + row;
                        if (m < Mrows && col < Ncols) o[(long)m * Ncols + col] = acc[i][j][r];
                        acc[i][j][r] = 0.f;
                    }
                }
        }
        const bool wrap = c + 1 >= nch;
        c = wrap ? 0 : c + 1;
        z = wrap ? z + 1 : z;
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }
}

// shared output transform: M values at M[xi*zstride + mbase], outputs at (oy + st*k, ox + st*l) of an H x W image.
// STATS: also returns the sum and the sum of squares (fp64) of the values it stored, per channel of the quad
template <int MO, bool STATS = false>
__device__ __forceinline__ void wino2_out_tile(const float* __restrict__ M, long zstride, long mbase, const float* __restrict__ bias,
                                               float* __restrict__ out, int ld_out, int n, int H, int W, int oy, int ox, int st,
                                               int c0, int act, double* s1 = nullptr, double* s2 = nullptr,
                                               pg_epi_mul mul = pg_epi_mul{nullptr, 0, 0}) {
    constexpr int NP = MO + 1;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 t[MO][NP];
#pragma unroll
    for (int k = 0; k < MO; ++k)
#pragma unroll
        for (int j = 0; j < NP; ++j) t[k][j] = z;
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(M + (long)(i * NP + j) * zstride + mbase);
#pragma unroll
            for (int k = 0; k < MO; ++k) t[k][j] += w_at<MO>(k, i) * m;
        }
    f32x4 bv = z;
    if (bias != nullptr) bv = *reinterpret_cast<const f32x4*>(bias + c0);
#pragma unroll
    for (int k = 0; k < MO; ++k)
#pragma unroll
        for (int l = 0; l < MO; ++l) {
            const int y = oy + st * k, x = ox + st * l;
            if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) continue;
            f32x4 v = bv;
#pragma unroll
            for (int j = 0; j < NP; ++j) v += t[k][j] * w_at<MO>(l, j);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_epi(v[e], act);
            if (mul.t) {        // data gradient: times the previous layer's activation derivative, expressed through its output t
                const f32x4 tv = *reinterpret_cast<const f32x4*>((const float*)mul.t + ((long)(n * H + y) * W + x) * mul.ld + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= pg_act_grad_from_out(tv[e], mul.act);
            }
            if (STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double d = (double)v[e];
                    s1[e] += d;
                    s2[e] += d * d;
                }
            }
            *reinterpret_cast<f32x4*>(out + ((long)(n * H + y) * W + x) * ld_out + c0) = v;
        }
}

// InstanceNorm statistics from the producer (the conv epilogue emits the sums, the separate statistics pass over
// y disappears).  The STATS forms of the two output-transform kernels run one grid row per sample (blockIdx.y = n) so that no
// workgroup straddles two samples; a workgroup covers 256 / cq consecutive (tile[, class]) units x cq channel quads, sums its
// units per channel in unit order through LDS (fixed order: deterministic) and writes part[(n * chunks + blockIdx.x) * C + c] =
// (sum, sum of squares) in fp64 -- the layout k_in_merge (norm_act.hip) reads.  Requires 256 % cq == 0.
template <int CQ_DUMMY = 0>
__device__ __forceinline__ void stats_block_reduce(const double* s1, const double* s2, int cq, int c0, bool active, double* __restrict__ part,
                                                   int n, int chunks, int C) {
    __shared__ double red[8][256];
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[e][tid] = active ? s1[e] : 0.0;
        red[4 + e][tid] = active ? s2[e] : 0.0;
    }
    __syncthreads();
    if (tid < cq) {
        double a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = red[e][tid];
        for (int u = tid + cq; u < 256; u += cq)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += red[e][u];
        double* o = part + (((long)n * chunks + blockIdx.x) * C + tid * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[2 * e] = a[e];
            o[2 * e + 1] = a[4 + e];
        }
    }
    (void)c0;
}

template <int MO>
__global__ __launch_bounds__(256) void k_wino2_out(const float* __restrict__ M, const float* __restrict__ bias,
                                                   float* __restrict__ out, int ld_out, int N, int Hs, int Ws, int Ca, int TH,
                                                   int TW, int act) {
    const int cq = Ca >> 2;
    const long T = (long)N * TH * TW;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= T * cq) return;
    const int c0 = (int)(idx % cq) << 2;
    const long tile = idx / cq;
    const int n = (int)(tile / (TH * TW));
    const int rem = (int)(tile - (long)n * TH * TW);
    const int ti = rem / TW, tj = rem - ti * TW;
    wino2_out_tile<MO>(M, T * Ca, tile * Ca + c0, bias, out, ld_out, n, Hs, Ws, MO * ti, MO * tj, 1, c0, act);
}

// grid (chunks, N): the same transform + per-sample partial sums of the output (see stats_block_reduce)
template <int MO>
__global__ __launch_bounds__(256) void k_wino2_out_stats(const float* __restrict__ M, const float* __restrict__ bias,
                                                         float* __restrict__ out, int ld_out, int N, int Hs, int Ws, int Ca, int TH,
                                                         int TW, int act, double* __restrict__ part) {
    const int cq = Ca >> 2, TT = TH * TW, n = blockIdx.y;
    const long T = (long)N * TT;
    const int local = blockIdx.x * 256 + threadIdx.x;
    const bool active = local < TT * cq;
    const int c0 = (local % cq) << 2;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (active) {
        const int rem = local / cq;
        const long tile = (long)n * TT + rem;
        const int ti = rem / TW, tj = rem - ti * TW;
        wino2_out_tile<MO, true>(M, T * Ca, tile * Ca + c0, bias, out, ld_out, n, Hs, Ws, MO * ti, MO * tj, 1, c0, act, s1, s2);
    }
    stats_block_reduce(s1, s2, cq, c0, active, part, n, gridDim.x, Ca);
}

// P is [tap][a][b] (b fastest), U is [.][b][a] (a fastest): a block transposes a 16 (a) x 16 (b) tile of the 16 taps through
// LDS so that both the reads and the writes are 64-byte runs.  grid (ceil(Cb/16), ceil(Ca/16)), 256 threads (32 x 32 tiles with
// 1024 threads and 66 KB of LDS gave 128 workgroups on a 512 x 256 layer: half the chip idle, 12 us for 22 MB).
template <int MO>
__device__ __forceinline__ void wino2c_u_body(const float* __restrict__ P, float* __restrict__ U, int Ca, int Cb, int bx, int by,
                                              float (*tile)[16][17]) {
    constexpr int NP = MO + 1;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int b0 = bx * 16, a0 = by * 16;
    {   // read: b fastest
        const int a = a0 + ty, b = b0 + tx;
        const bool ok = a < Ca && b < Cb;
#pragma unroll
        for (int t = 0; t < 16; ++t) tile[t][ty][tx] = ok ? P[((long)t * Ca + a) * Cb + b] : 0.f;
    }
    __syncthreads();
    const int a = a0 + tx, b = b0 + ty;     // write: a fastest
    if (a >= Ca || b >= Cb) return;
    float w[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 4; ++l) w[k][l] = tile[k * 4 + l][tx][ty];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
            float t[NP][2];      // g[t][t'] = w[kh(r,t)][kw(sc,t')]
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int kw = (sc == 0) ? 3 - 2 * v : 2 - 2 * v;
                    const float g0 = w[(r == 0) ? 3 : 2][kw], g1 = w[(r == 0) ? 1 : 0][kw];
                    t[i][v] = w_g<MO>(i, 0) * g0 + w_g<MO>(i, 1) * g1;
                }
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    U[(((long)(i * NP + j) * 4 + r * 2 + sc) * Cb + b) * Ca + a] = t[i][0] * w_g<MO>(j, 0) + t[i][1] * w_g<MO>(j, 1);
        }
}
template <int MO>
__global__ __launch_bounds__(256) void k_wino2c_u(const float* __restrict__ P, float* __restrict__ U, int Ca, int Cb) {
    __shared__ float tile[16][16][17];
    wino2c_u_body<MO>(P, U, Ca, Cb, blockIdx.x, blockIdx.y, tile);
}

// Several layers' weight transforms in ONE launch (pg_conv_prep_batch): block ranges [block0, next block0) per item, the item's own
// kernel body on its local block index.  Items are few (one per layer and direction of a network): a linear scan, uniform per block.
struct WinoPrepItem {
    const float* P;
    float* U;
    int Ca, Cb, kind, flip, gx, block0;      // kind: 2 / 3 = stride-1 F(2x2,4x4) / F(3x3,4x4) (Ca = Co, Cb = Ci), 10 + MO = big -> small, 20 + MO = small -> big
};
struct WinoPrepBatch {
    int n;
    WinoPrepItem it[PG_WINO_PREP_MAX];
};
__global__ __launch_bounds__(256) void k_wino_prep_batch(const WinoPrepBatch b) {
    __shared__ float tile[16][16][17];
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.it[i + 1].block0) ++i;
    const WinoPrepItem& t = b.it[i];
    const int blk = blockIdx.x - t.block0;
    switch (t.kind) {
        case 2: wino_u_body<2>(t.P, t.U, t.Ca, t.Cb, t.flip, blk); break;
        case 3: wino_u_body<3>(t.P, t.U, t.Ca, t.Cb, t.flip, blk); break;
        case 13: wino2_u_body<3>(t.P, t.U, t.Ca, t.Cb, blk); break;
        case 14: wino2_u_body<4>(t.P, t.U, t.Ca, t.Cb, blk); break;
        case 23: wino2c_u_body<3>(t.P, t.U, t.Ca, t.Cb, blk % t.gx, blk / t.gx, tile); break;
        case 24: wino2c_u_body<4>(t.P, t.U, t.Ca, t.Cb, blk % t.gx, blk / t.gx, tile); break;
        default: break;
    }
}

template <int MO>
__global__ __launch_bounds__(256) void k_wino2c_v(const float* __restrict__ small, int ld, float* __restrict__ V, int N, int Hs,
                                                  int Ws, int Ca, int TH, int TW) {
    const int cq = Ca >> 2;
    const long T = (long)N * TH * TW;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= T * cq) return;
    const int c0 = (int)(idx % cq) << 2;
    const long tile = idx / cq;
    const int n = (int)(tile / (TH * TW));
    const int rem = (int)(tile - (long)n * TH * TW);
    const int ti = rem / TW, tj = rem - ti * TW;
    wino2_v_tile<MO>(small, ld, n, Hs, Ws, MO * ti - 1, MO * tj - 1, 1, c0, V, T * Ca, tile * Ca + c0);
}

template <int MO>
__global__ __launch_bounds__(256) void k_wino2c_out(const float* __restrict__ M, const float* __restrict__ bias,
                                                    float* __restrict__ out, int ld_out, int N, int Hb, int Wb, int Cb, int TH,
                                                    int TW, int act, pg_epi_mul mul) {
    const int cq = Cb >> 2;
    const long T = (long)N * TH * TW;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= T * 4 * cq) return;
    const int c0 = (int)(idx % cq) << 2;
    long rr = idx / cq;
    const int cls = (int)(rr & 3);
    const long tile = rr >> 2;
    const int r = cls >> 1, sc = cls & 1;
    const int n = (int)(tile / (TH * TW));
    const int rem = (int)(tile - (long)n * TH * TW);
    const int ti = rem / TW, tj = rem - ti * TW;
    // class plane index i = MO*ti - r + al  ->  big row 2*i + r = 2*MO*ti - r + 2*al
    wino2_out_tile<MO>(M, 4L * T * Cb, tile * 4L * Cb + cls * Cb + c0, bias, out, ld_out, n, Hb, Wb, 2 * MO * ti - r,
                       2 * MO * tj - sc, 2, c0, act, nullptr, nullptr, mul);
}

// grid (chunks, N): the same transform + per-sample partial sums of the output (see stats_block_reduce)
template <int MO>
__global__ __launch_bounds__(256) void k_wino2c_out_stats(const float* __restrict__ M, const float* __restrict__ bias,
                                                          float* __restrict__ out, int ld_out, int N, int Hb, int Wb, int Cb, int TH,
                                                          int TW, int act, double* __restrict__ part) {
    const int cq = Cb >> 2, TT = TH * TW, n = blockIdx.y;
    const long T = (long)N * TT;
    const int local = blockIdx.x * 256 + threadIdx.x;
    const bool active = local < TT * 4 * cq;
    const int c0 = (local % cq) << 2;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (active) {
        const int rr = local / cq, cls = rr & 3, rem = rr >> 2;
        const int r = cls >> 1, sc = cls & 1;
        const long tile = (long)n * TT + rem;
        const int ti = rem / TW, tj = rem - ti * TW;
        wino2_out_tile<MO, true>(M, 4L * T * Cb, tile * 4L * Cb + cls * Cb + c0, bias, out, ld_out, n, Hb, Wb, 2 * MO * ti - r,
                                 2 * MO * tj - sc, 2, c0, act, s1, s2);
    }
    stats_block_reduce(s1, s2, cq, c0, active, part, n, gridDim.x, Cb);
}

// Weight gradient of the stride-2 layers, polyphase F(2x2, 3x3): per phase (r, s) the taps (2u+r, 2v+s), u, v in {0,1}, are
//   dW[2u+r][2v+s][a][b] = sum_pix dy[p][q][a] * X_rs[p+u][q+v][b]
// -- per 3x3 tile of dy and its 4x4 window of the phase (the forward's window: V is the forward's k_wino2_v<3>) a 2x2
// correlation with a 3x3 "kernel" dy: 16 instead of 36 multiplies.
//   k_wino2_dy         DY[xi][tile][a] = (G23 dy G23^T)[xi]
//   k_wino_wgrad_gemm  S[slice][xi][a][ph*Cb + b] = sum_{tile in slice} DY[xi][tile][a] * V[xi][tile][ph*Cb + b]
//   k_wino2_wgrad_out  dP[(2u+r)*4 + 2v+s][a][b] = sum_slice sum_xi A2T[u][xi_i] A2T[v][xi_j] S[slice][xi][a][ph*Cb + b]
__device__ __constant__ float c_G23[4][3] = {{-1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
__device__ __constant__ float c_A2T[2][4] = {{1, 1, 1, 0}, {0, 1, -1, 1}};

__global__ __launch_bounds__(256) void k_wino2_dy(const float* __restrict__ dy, int ld, float* __restrict__ DY, int N, int Hs,
                                                  int Ws, int Ca, int TH, int TW) {
    const int cq = Ca >> 2;
    const long T = (long)N * TH * TW;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= T * cq) return;
    const int c0 = (int)(idx % cq) << 2;
    const long tile = idx / cq;
    const int n = (int)(tile / (TH * TW));
    const int rem = (int)(tile - (long)n * TH * TW);
    const int ti = rem / TW, tj = rem - ti * TW;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int y = 3 * ti + u, x = 3 * tj + v;
            d[u][v] = (y < Hs && x < Ws) ? *reinterpret_cast<const f32x4*>(dy + ((long)(n * Hs + y) * Ws + x) * ld + c0) : z;
        }
    f32x4 t[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int v = 0; v < 3; ++v) t[a][v] = c_G23[a][0] * d[0][v] + c_G23[a][1] * d[1][v] + c_G23[a][2] * d[2][v];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            *reinterpret_cast<f32x4*>(DY + ((long)(a * 4 + b) * T + tile) * Ca + c0) =
                t[a][0] * c_G23[b][0] + t[a][1] * c_G23[b][1] + t[a][2] * c_G23[b][2];
}

// one thread per (a, phase, 4 consecutive b): 16-byte loads of the slabs, fixed slice order
__global__ __launch_bounds__(256) void k_wino2_wgrad_out(const float* __restrict__ S, int slices, float* __restrict__ dP,
                                                         int Ca, int Cb) {
    const int cq = Cb >> 2;
    const long ab = (long)Ca * Cb;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)Ca * 4 * cq) return;
    const int b = (int)(idx % cq) << 2;
    const int ph = (int)((idx / cq) & 3);
    const int a = (int)(idx / (4 * cq));
    const long K = 4L * Cb, slab = (long)Ca * K;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = z;
            for (int s = 0; s < slices; ++s)
                v += *reinterpret_cast<const f32x4*>(S + ((long)s * 16 + i * 4 + j) * slab + (long)a * K + ph * Cb + b);
            m[i][j] = v;
        }
    f32x4 t[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) t[u][j] = c_A2T[u][0] * m[0][j] + c_A2T[u][1] * m[1][j] + c_A2T[u][2] * m[2][j] + c_A2T[u][3] * m[3][j];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const f32x4 w = t[u][0] * c_A2T[v][0] + t[u][1] * c_A2T[v][1] + t[u][2] * c_A2T[v][2] + t[u][3] * c_A2T[v][3];
            const int kh = 2 * u + (ph >> 1), kw = 2 * v + (ph & 1);
            *reinterpret_cast<f32x4*>(dP + (long)(kh * 4 + kw) * ab + (long)a * Cb + b) = w;
        }
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// output tile edge of the stride-1 forward / data-gradient path: 2 = F(2x2,4x4), 3 = F(3x3,4x4); `forced` = 2 / 3 pins it
// (PG_TUNE_WINO1_F2 / _F3, PATCHGAN_WINO1_TILE), 0 = heuristic.
// F(3x3,4x4) keeps nine output accumulator sets per lane, so its workgroup tile is 64 tiles x 64 channels at 2 waves per SIMD:
// it is used when that grid still fills the chip (>= 240 workgroups), F(2x2,4x4) otherwise.
static bool wino1_split_off() {      // PATCHGAN_WINO1_SPLIT=0: no K split in the stride-1 GEMM (experiment switch)
    static const bool off = [] {
        const char* e = pg_exp_env("PATCHGAN_WINO1_SPLIT");
        return e && e[0] == '0';
    }();
    return off;
}
bool pg_wino_row_on() {      // F(3x3,4x4) as k_wino_gemm_row + k_wino_t_out (default) or the fully fused k_wino_gemm (PATCHGAN_WINO1_ROW=0)
    static const bool on = [] {
        const char* e = pg_exp_env("PATCHGAN_WINO1_ROW");
        return !(e && e[0] == '0');
    }();
    return on;
}
int pg_wino_mo(int N, int Hout, int Wout, int Cin, int Cout, int forced) {
    if (Cin % 64 != 0) return 2;                 // the F(3x3,4x4) instance walks K in chunks of 64
    if (forced == 2 || forced == 3) return forced;
    const long T3 = (long)N * ((Hout + 2) / 3) * ((Wout + 2) / 3);
    const long wg = ((T3 + 63) / 64) * ((Cout + 63) / 64);
    if (wg >= 240) return 3;
    // fewer workgroups than CUs: with the input channels split over 2 / 4 workgroups per tile (k_wino_gemm's nsl) the F(3x3,4x4)
    // instance still fills the chip (data gradient of the 512 -> 256 layer at batch 16: 124 tiles x 4 slices)
    const int nch = Cin / 64;
    const int smax = wino1_split_off() ? 1 : (nch % 4 == 0 ? 4 : nch % 2 == 0 ? 2 : 1);
    return wg * smax >= 240 ? 3 : 2;
}
// K slices of the F(3x3,4x4) GEMM: the fewest of 1 / 2 / 4 that give >= 400 workgroups (two per CU on most of the chip)
int pg_wino_gemm_slices(int N, int Hout, int Wout, int Cin, int Cout, int forced) {
    if (pg_wino_mo(N, Hout, Wout, Cin, Cout, forced) != 3 || wino1_split_off()) return 1;
    const long T3 = (long)N * ((Hout + 2) / 3) * ((Wout + 2) / 3);
    const long wg = ((T3 + 63) / 64) * ((Cout + 63) / 64);
    const int nch = Cin / 64;
    static const long want = [] {
        const char* e = pg_exp_env("PATCHGAN_WINO1_SPLIT_WG");
        return e ? atol(e) : 400L;
    }();
    if (wg >= want) return 1;
    if (nch % 2 == 0 && (wg * 2 >= want || nch % 4 != 0)) return 2;
    return nch % 4 == 0 ? 4 : 1;
}
static long wino1_tiles(int N, int Hout, int Wout, int Cin, int Cout, int forced) {
    const int mo = pg_wino_mo(N, Hout, Wout, Cin, Cout, forced);
    return (long)N * ((Hout + mo - 1) / mo) * ((Wout + mo - 1) / mo);
}
static long wino1_nxi(int N, int Hout, int Wout, int Cin, int Cout, int forced) {
    const int np = pg_wino_mo(N, Hout, Wout, Cin, Cout, forced) + 3;
    return (long)np * np;
}

bool pg_wino_geom_ok(int N, int Hout, int Wout, int Cin, int Cout, int forced) {
    if (Cin % 32 != 0 || Cin < 64 || Cout < 64) return false;
    const long T = wino1_tiles(N, Hout, Wout, Cin, Cout, forced), X = wino1_nxi(N, Hout, Wout, Cin, Cout, forced);
    if ((long)N * ((Hout + 1) / 2) * ((Wout + 1) / 2) < 2048) return false;       // needs enough tiles to fill the chip
    if ((double)X * T * Cin * 4 >= 1.5e9 || (double)X * Cout * Cin * 4 >= 1.0e9) return false;   // 32-bit buffer offsets
    return true;
}

bool pg_wino_eligible(int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int ld_in, const void* in, int forced) {
    (void)Hin;
    (void)Win;
    return pg_wino_geom_ok(N, Hout, Wout, Cin, Cout, forced) && ld_in % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
}

// 128-tile rows unless that leaves the 256 CUs short of two workgroups each
bool pg_wino_small_tile(int N, int Hout, int Wout, int Cin, int Cout, int mo_forced) {
    static const int forced = [] {
        const char* e = pg_exp_env("PATCHGAN_WINO_TILE");
        return e ? atoi(e) : 0;
    }();
    if (pg_wino_mo(N, Hout, Wout, Cin, Cout, mo_forced) == 3) return true;          // nine output accumulator sets: 64-tile rows only
    const long T = (long)N * ((Hout + 1) / 2) * ((Wout + 1) / 2);
    const long wg128 = ((T + 127) / 128) * ((Cout + 63) / 64);
    return forced ? forced == 1 : wg128 < 400;
}

size_t pg_wino_u_bytes(int N, int Hout, int Wout, int Cin, int Cout, int forced) {
    return align256((size_t)wino1_nxi(N, Hout, Wout, Cin, Cout, forced) * Cout * Cin * 4);
}
size_t pg_wino_ws_bytes(int N, int Hout, int Wout, int Cin, int Cout, int forced) {
    const long T = wino1_tiles(N, Hout, Wout, Cin, Cout, forced), X = wino1_nxi(N, Hout, Wout, Cin, Cout, forced);
    const int nsl = pg_wino_gemm_slices(N, Hout, Wout, Cin, Cout, forced);
    const size_t slabs = nsl > 1 ? align256((size_t)nsl * N * Hout * Wout * Cout * 4) : 0;
    const size_t trow = (X == 36) ? align256((size_t)18 * T * Cout * 4) : 0;          // T of the row-split form
    return align256((size_t)X * Cout * Cin * 4) + align256((size_t)X * T * Cin * 4) + std::max(slabs, trow);   // U | V | slabs or T
}
// the row-split form runs (and then NO slab reduce follows): F(3x3,4x4), register-staged, 16-byte-aligned output / bias / multiplier
bool pg_wino_gemm_rows(int N, int Hout, int Wout, int Cin, int Cout, int forced, int dma_mode, const float* out, int ld_out,
                       const float* bias, pg_epi_mul mul) {
    if (!pg_wino_row_on() || dma_mode || pg_wino_mo(N, Hout, Wout, Cin, Cout, forced) != 3 || (Cout & 3) || (ld_out & 3)) return false;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return al(out) && (!bias || al(bias)) && (!mul.t || (al(mul.t) && (mul.ld & 3) == 0));
}
float* pg_wino_gemm_slabs(void* ws, int N, int Hout, int Wout, int Cin, int Cout, int forced) {
    const long T = wino1_tiles(N, Hout, Wout, Cin, Cout, forced), X = wino1_nxi(N, Hout, Wout, Cin, Cout, forced);
    return (float*)((char*)ws + align256((size_t)X * Cout * Cin * 4) + align256((size_t)X * T * Cin * 4));
}

int pg_wino_prepare(const float* in, int ld_in, const float* P, int flip, int N, int Hin, int Win, int Cin, int Hout,
                    int Wout, int Cout, int pad, void* ws, hipStream_t st, int forced, float* Uext, int u_valid, float* Vext) {
    const int mo = pg_wino_mo(N, Hout, Wout, Cin, Cout, forced), TH = (Hout + mo - 1) / mo, TW = (Wout + mo - 1) / mo;
    const long T = (long)N * TH * TW, X = wino1_nxi(N, Hout, Wout, Cin, Cout, forced);
    float* U = Uext ? Uext : (float*)ws;       // Uext: caller-owned cache of the transformed weights (u_valid: already filled)
    float* V = Vext ? Vext : (float*)((char*)ws + align256((size_t)X * Cout * Cin * 4));      // Vext: caller-owned (kept for the weight gradient)
    const dim3 gu((unsigned)(((long)Cout * Cin + 255) / 256)), gv((unsigned)((T * (Cin / 4) + 255) / 256));
    if (!(Uext && u_valid)) {
        if (mo == 3)
            hipLaunchKernelGGL(k_wino_u<3>, gu, dim3(256), 0, st, P, U, Cout, Cin, flip);
        else
            hipLaunchKernelGGL(k_wino_u<2>, gu, dim3(256), 0, st, P, U, Cout, Cin, flip);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    }
    if (mo == 3)
        hipLaunchKernelGGL(k_wino_v<3>, gv, dim3(256), 0, st, in, ld_in, V, N, Hin, Win, Cin, TH, TW, pad);
    else
        hipLaunchKernelGGL(k_wino_v<2>, gv, dim3(256), 0, st, in, ld_in, V, N, Hin, Win, Cin, TH, TW, pad);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

// PATCHGAN_WINO_DMA (A/B switch): 0 (default) = register-staged stride-1 kernels only, 1 = LDS-DMA ring kernel for the
// F(3x3,4x4) instance, 2 = also for the 64-tile F(2x2,4x4) instance.  Measured equal within device noise (DESIGN.md section 3:
// on this 64x64-tile loop every ds_read_b128 costs ~34 and every DMA piece / load + ds_write pair ~64 cycles of MFMA-pipe time
// whichever way the tile is staged), so the register-staged kernel stays the default.
int pg_wino_dma_mode() {
    static const int mode = [] {
        const char* e = pg_exp_env("PATCHGAN_WINO_DMA");
        return e ? atoi(e) : 0;
    }();
    return mode;
}

int pg_wino_gemm(const float* bias, float* out, int ld_out, int N, int Cin, int Hout, int Wout, int Cout, int act,
                 void* ws, hipStream_t st, int forced, int dma_mode, const float* Uext, pg_epi_mul mul, const float* Vext, int s3) {
    const int mo = pg_wino_mo(N, Hout, Wout, Cin, Cout, forced), TH = (Hout + mo - 1) / mo, TW = (Wout + mo - 1) / mo;
    const long T = (long)N * TH * TW, X = wino1_nxi(N, Hout, Wout, Cin, Cout, forced);
    const float* U = Uext ? Uext : (const float*)ws;
    const float* V = Vext ? Vext : (const float*)((const char*)ws + align256((size_t)X * Cout * Cin * 4));
    const bool small_tile = pg_wino_small_tile(N, Hout, Wout, Cin, Cout, forced);
    const int v_bytes = (int)(X * T * Cin * 4), u_bytes = (int)(X * Cout * Cin * 4);
    const int tn = (Cout + 63) / 64;
    if (pg_wino_gemm_rows(N, Hout, Wout, Cin, Cout, forced, dma_mode, out, ld_out, bias, mul)) {
        float* Tt = pg_wino_gemm_slabs(ws, N, Hout, Wout, Cin, Cout, forced);
        // (MR = 2 -- 128 tiles per workgroup, 64 x 32 per wave, 2 waves per SIMD -- measured 0.14 ms per step slower)
        if (s3 && Cin % 32 == 0)
            hipLaunchKernelGGL((k_wino_gemm_row_s3<2>), dim3((unsigned)(((T + 127) / 128) * tn * 6)), dim3(256), 0, st, V, U, Tt, (int)T, Cin,
                               Cout, v_bytes, u_bytes, tn);
        else
            hipLaunchKernelGGL((k_wino_gemm_row<4, 1>), dim3((unsigned)(((T + 63) / 64) * tn * 6)), dim3(256), 0, st, V, U, Tt, (int)T, Cin, Cout,
                               v_bytes, u_bytes, tn);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        const dim3 go((unsigned)((T * (Cout / 4) + 255) / 256));
        if (mul.t)
            hipLaunchKernelGGL(k_wino_t_out<true>, go, dim3(256), 0, st, Tt, bias, out, ld_out, (int)T, Cout, TH, TW, Hout, Wout, act, mul);
        else
            hipLaunchKernelGGL(k_wino_t_out<false>, go, dim3(256), 0, st, Tt, bias, out, ld_out, (int)T, Cout, TH, TW, Hout, Wout, act, mul);
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    const int nsl = pg_wino_gemm_slices(N, Hout, Wout, Cin, Cout, forced);
    if (nsl > 1) {       // partial outputs into dense [pixel][Cout] slabs behind U | V; the caller reduces them (bias / act / mul there)
        float* slabs = pg_wino_gemm_slabs(ws, N, Hout, Wout, Cin, Cout, forced);
        dim3 grid((unsigned)(((T + 63) / 64) * tn * nsl));
        hipLaunchKernelGGL((k_wino_gemm<1, 1, 2, 2, 2, 3>), grid, dim3(256), 0, st, V, U, (const float*)nullptr, slabs, Cout, (int)T, Cin, Cout,
                           TH, TW, Hout, Wout, PG_ACT_NONE, v_bytes, u_bytes, tn, pg_epi_mul{nullptr, 0, 0}, nsl, (long)N * Hout * Wout * Cout);
        return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
    }
    if (mul.t) {                       // the register-staged kernels' MUL instantiations
        if (mo == 3) {
            dim3 grid((unsigned)(((T + 63) / 64) * tn));
            hipLaunchKernelGGL((k_wino_gemm<1, 1, 2, 2, 2, 3, true>), grid, dim3(256), 0, st, V, U, bias, out, ld_out, (int)T, Cin, Cout, TH,
                               TW, Hout, Wout, act, v_bytes, u_bytes, tn, mul, 1, 0L);
        } else if (small_tile) {
            dim3 grid((unsigned)(((T + 63) / 64) * tn));
            hipLaunchKernelGGL((k_wino_gemm<1, 1, 2, 2, 4, 2, true>), grid, dim3(256), 0, st, V, U, bias, out, ld_out, (int)T, Cin, Cout, TH,
                               TW, Hout, Wout, act, v_bytes, u_bytes, tn, mul, 1, 0L);
        } else {
            dim3 grid((unsigned)(((T + 127) / 128) * tn));
            hipLaunchKernelGGL((k_wino_gemm<2, 1, 2, 2, 2, 2, true>), grid, dim3(256), 0, st, V, U, bias, out, ld_out, (int)T, Cin, Cout, TH,
                               TW, Hout, Wout, act, v_bytes, u_bytes, tn, mul, 1, 0L);
        }
    } else if (mo == 3 && dma_mode) {
        dim3 grid((unsigned)(((T + 63) / 64) * tn));
        hipLaunchKernelGGL((k_wino_gemm_dma<3, 4, 2>), grid, dim3(256), 0, st, V, U, bias, out, ld_out, (int)T, Cin, Cout, TH, TW,
                           Hout, Wout, act, v_bytes, u_bytes, tn);
    } else if (mo == 3) {
        dim3 grid((unsigned)(((T + 63) / 64) * tn));
        hipLaunchKernelGGL((k_wino_gemm<1, 1, 2, 2, 2, 3>), grid, dim3(256), 0, st, V, U, bias, out, ld_out, (int)T, Cin, Cout, TH,
                           TW, Hout, Wout, act, v_bytes, u_bytes, tn, mul, 1, 0L);
    } else if (small_tile && dma_mode == 2) {
        dim3 grid((unsigned)(((T + 63) / 64) * tn));
        hipLaunchKernelGGL((k_wino_gemm_dma<2, 3, 3>), grid, dim3(256), 0, st, V, U, bias, out, ld_out, (int)T, Cin, Cout, TH, TW,
                           Hout, Wout, act, v_bytes, u_bytes, tn);
    } else if (small_tile) {
        dim3 grid((unsigned)(((T + 63) / 64) * tn));
        hipLaunchKernelGGL((k_wino_gemm<1, 1, 2, 2, 4, 2>), grid, dim3(256), 0, st, V, U, bias, out, ld_out, (int)T, Cin, Cout, TH,
                           TW, Hout, Wout, act, v_bytes, u_bytes, tn, mul, 1, 0L);
    } else {
        dim3 grid((unsigned)(((T + 127) / 128) * tn));
        hipLaunchKernelGGL((k_wino_gemm<2, 1, 2, 2, 2, 2>), grid, dim3(256), 0, st, V, U, bias, out, ld_out, (int)T, Cin, Cout, TH,
                           TW, Hout, Wout, act, v_bytes, u_bytes, tn, mul, 1, 0L);
    }
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

// ---- weight gradient ----
// dy-tile edge: 3 (F(4x4,3x3), 36 points) where that still leaves >= 1024 tiles (32 K chunks), else 2 (F(4x4,2x2), 25 points);
// PATCHGAN_WINOW_R=2|3 pins it (experiment switch)
// Workgroups the K split of a weight-gradient GEMM aims at.  fp32 kernels: 768 = three per CU.  Split-bf16 128 x 128 tile: 128 = HALF a round of
// the chip.  Every extra slice shortens the K loops (a prologue and an epilogue per workgroup) and adds a slab to write and to sum; in the
// training step these GEMMs run on the second stream beside the data-gradient chain, which keeps the chip full anyway, so the step prefers few
// slices, while the GEMM alone on the chip (one-stream steps, the bench's per-kernel sample) prefers many.  cfg2, same box -- step on two
// streams | the nine GEMMs alone on the chip | the cfg2 loss curve's largest distance from the reference's, by fill:
//   768: 7.24 ms | -       | <= 1e-4      384: 7.12 | 1.11 | 1.04e-4      192: 7.01 | 1.29 | 1.02e-4      96: 7.03      no split: 8.22
//   512: 7.18    | 1.08 ms | 1.02e-4      256: 7.06 | 1.21 | 1.06e-4      128: 6.94 | 1.61 | 0.94e-4      64: 7.46
// One value for every way of launching the step: the slices fix the order of the sums, and the one-stream and two-stream steps stay
// bit-identical.  (A bounded-grid launch of the 512-fill items -- the same sums on 128 / 256 workgroups -- measured WORSE: 7.6 / 7.08 ms: what
// helps is the longer loop and the smaller slab traffic, not the smaller footprint.)  The third column is why 128 and not 192: the loss curve
// of this chaotic problem moves by ~1e-5 with ANY change of a summation order (EXPERIMENTS.md, round 6), and the north star's gate is 1e-4.
// PATCHGAN_S3W_FILL overrides it (experiment).
static long wgrad_s3_fill(int s3, int tile) {
    static const int forced = [] {
        const char* e = pg_exp_env("PATCHGAN_S3W_FILL");
        return e ? atoi(e) : 0;
    }();
    if (s3 && tile == 128) return forced ? forced : 128;
    return 768;
}
int pg_wino_wgrad_r(int N, int Hs, int Ws) {
    static const int forced = [] {
        const char* e = pg_exp_env("PATCHGAN_WINOW_R");
        return e ? atoi(e) : 0;
    }();
    if (forced == 2 || forced == 3) return forced;
    return (long)N * ((Hs + 2) / 3) * ((Ws + 2) / 3) >= 1024 ? 3 : 2;
}
static long wgrad_tiles(int N, int Hs, int Ws) {
    const int r = pg_wino_wgrad_r(N, Hs, Ws);
    return (long)N * ((Hs + r - 1) / r) * ((Ws + r - 1) / r);
}
static int wgrad_nxi(int N, int Hs, int Ws) {
    const int np = pg_wino_wgrad_r(N, Hs, Ws) + 3;
    return np * np;
}
double pg_wino_wgrad_flops(int N, int Hs, int Ws, int Ca, int Cb) {
    return 2.0 * wgrad_nxi(N, Hs, Ws) * wgrad_tiles(N, Hs, Ws) * Ca * Cb;
}
bool pg_wino_wgrad_geom_ok(int N, int Hs, int Ws, int Ca, int Cb) {
    if (Ca % 4 != 0 || Cb % 4 != 0 || Ca < 64 || Cb < 64) return false;
    if ((long)N * ((Hs + 1) / 2) * ((Ws + 1) / 2) < 2048) return false;
    const double xt = (double)wgrad_nxi(N, Hs, Ws) * wgrad_tiles(N, Hs, Ws);
    if (xt * Ca * 4 >= 1.5e9 || xt * Cb * 4 >= 1.5e9) return false;
    return true;
}

// stride-1 weight gradient: 64x64 output tiles unless 128x128 ones alone give >= 768 workgroups (no K split then)
// (split-bf16 form: the 128x128 tile wherever both channel counts fill it -- its staging work per MFMA is half the 64x64 tile's, which
//  measured no faster than the fp32 kernel -- and correspondingly more K slices)
bool pg_wino_wgrad_tile64(int Ca, int Cb, int s3) {
    static const int forced = [] {
        const char* e = pg_exp_env("PATCHGAN_WINOW_TILE");
        return e ? atoi(e) : 0;
    }();
    if (forced) return forced == 64;
    if (s3 && Ca % 128 == 0 && Cb % 128 == 0) return false;
    return 25L * ((Ca + 127) / 128) * ((Cb + 127) / 128) < 768;
}
int pg_wino_wgrad_slices(int N, int Hs, int Ws, int Ca, int Cb, int s3) {
    const long T = wgrad_tiles(N, Hs, Ws);
    const int tt = pg_wino_wgrad_tile64(Ca, Cb, s3) ? 64 : 128;
    const long wgs = (long)wgrad_nxi(N, Hs, Ws) * ((Ca + tt - 1) / tt) * ((Cb + tt - 1) / tt);
    const long fill = wgrad_s3_fill(s3, tt);
    long s = (fill + wgs - 1) / wgs;                 // three workgroups per CU (fp32), two (split-bf16 128x128 tile)
    const long nchunks = (T + KC - 1) / KC;
    if (s > nchunks / 16) s = nchunks / 16;          // at least 16 chunks per slice
    return (int)(s < 1 ? 1 : s);
}

size_t pg_wino_wgrad_ws_bytes(int N, int Hs, int Ws, int Ca, int Cb) {
    const long T = wgrad_tiles(N, Hs, Ws), X = wgrad_nxi(N, Hs, Ws);
    const int sl = std::max(pg_wino_wgrad_slices(N, Hs, Ws, Ca, Cb, 0), pg_wino_wgrad_slices(N, Hs, Ws, Ca, Cb, 1));      // serves both forms
    return align256((size_t)X * T * Cb * 4) + align256((size_t)X * T * Ca * 4) + align256((size_t)sl * X * Ca * Cb * 4);
}

// the transformed input the forward F(3x3,4x4) leaves (same 6 x 6 windows at 3t - 1, same B^T) is the V of F(4x4,3x3)
size_t pg_wino_wgrad_v_bytes(int N, int Hs, int Ws, int Ca, int Cb, int fwd_forced) {
    if (pg_wino_wgrad_r(N, Hs, Ws) != 3 || pg_wino_mo(N, Hs, Ws, Cb, Ca, fwd_forced) != 3) return 0;
    return align256((size_t)36 * wgrad_tiles(N, Hs, Ws) * Cb * 4);
}

int pg_wino_wgrad(const float* small, int ld_small, const float* big, int ld_big, float* dP, int N, int Hb, int Wb, int Hs,
                  int Ws, int Ca, int Cb, void* ws, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, const float* Vpre, int s3) {
    const int R = pg_wino_wgrad_r(N, Hs, Ws), X = (R + 3) * (R + 3);
    const int TH = (Hs + R - 1) / R, TW = (Ws + R - 1) / R;
    const long T = (long)N * TH * TW;
    if (Vpre && R != 3) return PG_EINVAL;
    const float* V = Vpre ? Vpre : (const float*)ws;
    float* DY = (float*)((char*)ws + align256((size_t)X * T * Cb * 4));
    float* S = (float*)((char*)DY + align256((size_t)X * T * Ca * 4));
    const dim3 gv((unsigned)((T * (Cb / 4) + 255) / 256)), gd((unsigned)((T * (Ca / 4) + 255) / 256));
    if (R == 3) {
        if (!Vpre) hipLaunchKernelGGL(k_wino_v<3>, gv, dim3(256), 0, st, big, ld_big, (float*)ws, N, Hb, Wb, Cb, TH, TW, 1);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        hipLaunchKernelGGL(k_wino_dy<3>, gd, dim3(256), 0, st, small, ld_small, DY, N, Hs, Ws, Ca, TH, TW);
    } else {
        hipLaunchKernelGGL(k_wino_v<2>, gv, dim3(256), 0, st, big, ld_big, (float*)ws, N, Hb, Wb, Cb, TH, TW, 1);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        hipLaunchKernelGGL(k_wino_dy<2>, gd, dim3(256), 0, st, small, ld_small, DY, N, Hs, Ws, Ca, TH, TW);
    }
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    const int slices = pg_wino_wgrad_slices(N, Hs, Ws, Ca, Cb, s3);
    const int nchunks = (int)((T + KC - 1) / KC);
    const int cps = (nchunks + slices - 1) / slices;
    if (ev0) (void)hipEventRecord(ev0, st);
    if (pg_wino_wgrad_tile64(Ca, Cb, s3)) {
        const int tilesA = (Ca + 63) / 64, tilesB = (Cb + 63) / 64;
        if (s3)
            hipLaunchKernelGGL((k_wino_wgrad_gemm_s3<1, 1, 2, 2, 2, 3>), dim3(tilesA * tilesB * X * slices), dim3(256), 0, st, DY, V, S, (int)T,
                               Ca, Cb, cps, tilesA, tilesB, X, (int)((long)X * T * Ca * 4), (int)((long)X * T * Cb * 4));
        else
            hipLaunchKernelGGL((k_wino_wgrad_gemm<1, 1, 2, 2>), dim3(tilesA * tilesB * X * slices), dim3(256), 0, st, DY, V, S, (int)T, Ca,
                               Cb, cps, tilesA, tilesB, X, (int)((long)X * T * Ca * 4), (int)((long)X * T * Cb * 4));
    } else {
        const int tilesA = (Ca + 127) / 128, tilesB = (Cb + 127) / 128;
        if (s3)
            hipLaunchKernelGGL((k_wino_wgrad_gemm_s3<2, 2, 2, 2, 1, 2>), dim3(tilesA * tilesB * X * slices), dim3(256), 0, st, DY, V, S, (int)T,
                               Ca, Cb, cps, tilesA, tilesB, X, (int)((long)X * T * Ca * 4), (int)((long)X * T * Cb * 4));
        else
            hipLaunchKernelGGL((k_wino_wgrad_gemm<2, 2, 2, 2>), dim3(tilesA * tilesB * X * slices), dim3(256), 0, st, DY, V, S, (int)T, Ca,
                               Cb, cps, tilesA, tilesB, X, (int)((long)X * T * Ca * 4), (int)((long)X * T * Cb * 4));
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    const dim3 go((unsigned)(((long)Ca * Cb + 255) / 256));
    if (R == 3)
        hipLaunchKernelGGL(k_wino_wgrad_out<3>, go, dim3(256), 0, st, S, slices, dP, Ca, Cb);
    else
        hipLaunchKernelGGL(k_wino_wgrad_out<2>, go, dim3(256), 0, st, S, slices, dP, Ca, Cb);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

// batches per workgroup of k_wino_bgemm_mz (1 -> the single-batch kernel): the largest divisor of X that still leaves >= 768 workgroups.
// OFF by default since round 4 (PATCHGAN_BGEMM_MZ=1 enables it): measured in the cfg2 step on one box, the multi-batch kernel as hipcc
// built it until then (196 VGPRs + 64 AGPRs: ONE wave per SIMD) 9.07 ms, the same kernel held to two waves per SIMD 8.95, the single-batch
// kernel (126 registers, four waves per SIMD, four workgroups per CU covering each other's prologues) 8.90.
// the split-bf16 batched GEMM: 128-row tiles at two waves per SIMD (194 registers), 64-row tiles at three (118)
static void bgemm_s3_launch(int big_rows, dim3 grid, hipStream_t st, const float* A, const float* B, float* C, int Mrows, int Ncols, int K,
                            int a_bytes, int b_bytes, int tm, int tn) {
    static const int var = [] {      // PATCHGAN_S3_VAR (experiment): order of the six products, see k_wino_bgemm_s3
        const char* e = pg_exp_env("PATCHGAN_S3_VAR");
        return e ? atoi(e) : 1;
    }();
    static const int pipe = [] {     // PATCHGAN_S3_PIPE=1 (experiment): the register-pipelined loop (bit-identical; 4-6 % faster back to back in
        const char* e = pg_exp_env("PATCHGAN_S3_PIPE");      // tools/s3_probe.hip, no faster inside the step: 7.37 vs 7.41 ms, EXPERIMENTS.md)
        return e ? atoi(e) : 0;
    }();
    static const int big = [] {      // PATCHGAN_S3_BIG=<rows> (experiment): the 256 x 128 tile on eight waves for GEMMs with at least <rows> rows (bit-identical)
        const char* e = pg_exp_env("PATCHGAN_S3_BIG");
        return e ? atoi(e) : 0;
    }();
    if (var == 1 && big && big_rows && Mrows >= big) {
        const int X = (int)grid.x / (tm * tn), tm2 = (Mrows + 255) / 256;
        hipLaunchKernelGGL((k_wino_bgemm_s3<2, 2, 4, 2, 2, 1, false>), dim3(tm2 * tn * X), dim3(512), 0, st, A, B, C, Mrows, Ncols, K, a_bytes, b_bytes,
                           tm2, tn);
        return;
    }
    if (var == 1 && pipe) {
        if (big_rows)
            hipLaunchKernelGGL((k_wino_bgemm_s3<2, 2, 2, 2, 2, 1, true>), grid, dim3(256), 0, st, A, B, C, Mrows, Ncols, K, a_bytes, b_bytes, tm, tn);
        else
            hipLaunchKernelGGL((k_wino_bgemm_s3<1, 2, 2, 2, 3, 1, true>), grid, dim3(256), 0, st, A, B, C, Mrows, Ncols, K, a_bytes, b_bytes, tm, tn);
        return;
    }
#define S3_GO(MR, WPE, VAR) hipLaunchKernelGGL((k_wino_bgemm_s3<MR, 2, 2, 2, WPE, VAR>), grid, dim3(256), 0, st, A, B, C, Mrows, Ncols, K, a_bytes, b_bytes, tm, tn)
    if (var == 0) { if (big_rows) S3_GO(2, 2, 0); else S3_GO(1, 3, 0); }
    else if (var == 2) { if (big_rows) S3_GO(2, 2, 2); else S3_GO(1, 3, 2); }
    else if (var == 3) { if (big_rows) S3_GO(2, 2, 3); else S3_GO(1, 3, 3); }
    else if (big_rows) S3_GO(2, 2, 1);
    else S3_GO(1, 3, 1);
#undef S3_GO
}
static int bgemm_zb(long tiles_mn, int X) {
    static const bool on = [] {
        const char* e = pg_exp_env("PATCHGAN_BGEMM_MZ");
        return e && e[0] == '1';
    }();
    if (!on) return 1;
    int zb = 1;
    for (int d = 2; d <= X; ++d)
        if (X % d == 0 && tiles_mn * (X / d) >= 768) zb = d;
    return zb;
}

// ---- stride-2 layers (polyphase / parity classes) ----
int pg_wino2_mo() {     // output tile edge: 3 (default) or 4 (PATCHGAN_WINO2_TILE=4)
    static const int mo = [] {
        const char* e = pg_exp_env("PATCHGAN_WINO2_TILE");
        return (e && atoi(e) == 4) ? 4 : 3;
    }();
    return mo;
}
static long wino2_tiles(int N, int H, int W) {
    const int mo = pg_wino2_mo();
    return (long)N * ((H + mo - 1) / mo) * ((W + mo - 1) / mo);
}
static long wino2_nxi() { return (long)(pg_wino2_mo() + 1) * (pg_wino2_mo() + 1); }

long pg_wino2_tiles_b2s(int N, int Hs, int Ws) { return wino2_tiles(N, Hs, Ws); }
long pg_wino2_tiles_s2b(int N, int Hb, int Wb) { return wino2_tiles(N, (Hb + 1) / 2 + 1, (Wb + 1) / 2 + 1); }

bool pg_wino2_geom_ok(int N, int Hs, int Ws, int Ca, int Cb) {
    if (Cb % 8 != 0 || Ca % 4 != 0 || Cb < 32 || Ca < 64) return false;
    const long T = wino2_tiles(N, Hs, Ws), X = wino2_nxi();
    if (T < 64) return false;
    if ((double)X * T * 4 * Cb * 4 >= 1.5e9 || (double)X * Ca * 4 * Cb * 4 >= 1.5e9 || (double)X * T * Ca * 4 >= 1.5e9) return false;
    return true;
}

size_t pg_wino2_ws_bytes(int N, int Hs, int Ws, int Ca, int Cb) {
    const long T = wino2_tiles(N, Hs, Ws), X = wino2_nxi();
    return align256((size_t)X * Ca * 4 * Cb * 4) + align256((size_t)X * T * 4 * Cb * 4) + align256((size_t)X * T * Ca * 4);
}

template <int MO>
static int wino2_b2s_run(const float* big, int ld_big, const float* P, const float* bias, float* small, int ld_small, int N,
                         int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int act, void* ws, hipStream_t st, hipEvent_t ev0,
                         hipEvent_t ev1, const float* Vpre, double* part, float* Vkeep, float* Uext, int u_valid, int s3) {
    constexpr int X = (MO + 1) * (MO + 1);
    const int TH = (Hs + MO - 1) / MO, TW = (Ws + MO - 1) / MO, K = 4 * Cb;
    const long T = (long)N * TH * TW;
    // ws: U | V | M, or U | M when the caller supplies the transformed input (Vpre: shared with the weight gradient).
    // Vkeep: write V there (caller-owned, kept for the layer's weight gradient); Uext: caller-owned cache of U
    float* Uws = (float*)ws;
    float* U = Uext ? Uext : Uws;
    float* Vws = (float*)((char*)Uws + align256((size_t)X * Ca * K * 4));
    float* M = Vpre ? Vws : (float*)((char*)Vws + align256((size_t)X * T * K * 4));
    float* Vown = Vkeep ? Vkeep : Vws;
    const float* V = Vpre ? Vpre : Vown;
    if (!(Uext && u_valid)) {
        hipLaunchKernelGGL(k_wino2_u<MO>, dim3((unsigned)(((long)Ca * Cb + 255) / 256)), dim3(256), 0, st, P, U, Ca, Cb);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    }
    if (!Vpre) {
        hipLaunchKernelGGL(k_wino2_v<MO>, dim3((unsigned)((T * Cb + 255) / 256)), dim3(256), 0, st, big, ld_big, Vown, N, Hb, Wb, Cb,
                           TH, TW);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    }
    if (ev0) (void)hipEventRecord(ev0, st);
    const int a_bytes = (int)((long)X * T * K * 4), b_bytes = (int)((long)X * Ca * K * 4);
    {
        const int big_rows = T >= 1024;
        const int tm = (int)(big_rows ? (T + 127) / 128 : (T + 63) / 64), tn = (Ca + 127) / 128;
        const int zb = s3 ? 1 : bgemm_zb((long)tm * tn, X);
        const dim3 grid((unsigned)(tm * tn * (X / zb)));
        if (s3)
            bgemm_s3_launch(big_rows, grid, st, V, U, M, (int)T, Ca, K, a_bytes, b_bytes, tm, tn);
        else if (big_rows && zb > 1)
            hipLaunchKernelGGL((k_wino_bgemm_mz<2, 2, 2, 2>), grid, dim3(256), 0, st, V, U, M, (int)T, Ca, K, zb, a_bytes, b_bytes, tm, tn);
        else if (big_rows)
            hipLaunchKernelGGL((k_wino_bgemm<2, 2, 2, 2>), grid, dim3(256), 0, st, V, U, M, (int)T, Ca, K, a_bytes, b_bytes, tm, tn);
        else if (zb > 1)
            hipLaunchKernelGGL((k_wino_bgemm_mz<1, 2, 2, 2>), grid, dim3(256), 0, st, V, U, M, (int)T, Ca, K, zb, a_bytes, b_bytes, tm, tn);
        else
            hipLaunchKernelGGL((k_wino_bgemm<1, 2, 2, 2>), grid, dim3(256), 0, st, V, U, M, (int)T, Ca, K, a_bytes, b_bytes, tm, tn);
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    if (part)
        hipLaunchKernelGGL(k_wino2_out_stats<MO>, dim3((unsigned)pg_wino2_b2s_stats_chunks(N, Hs, Ws, Ca), N), dim3(256), 0, st, M, bias,
                           small, ld_small, N, Hs, Ws, Ca, TH, TW, act, part);
    else
        hipLaunchKernelGGL(k_wino2_out<MO>, dim3((unsigned)((T * (Ca / 4) + 255) / 256)), dim3(256), 0, st, M, bias, small, ld_small,
                           N, Hs, Ws, Ca, TH, TW, act);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

int pg_wino2_b2s(const float* big, int ld_big, const float* P, const float* bias, float* small, int ld_small, int N, int Hb,
                 int Wb, int Hs, int Ws, int Ca, int Cb, int act, void* ws, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                 const float* Vpre, double* part, float* Vkeep, float* Uext, int u_valid, int s3) {
    if (pg_wino2_mo() == 4)
        return wino2_b2s_run<4>(big, ld_big, P, bias, small, ld_small, N, Hb, Wb, Hs, Ws, Ca, Cb, act, ws, st, ev0, ev1, nullptr, part,
                                nullptr, Uext, u_valid, s3);
    return wino2_b2s_run<3>(big, ld_big, P, bias, small, ld_small, N, Hb, Wb, Hs, Ws, Ca, Cb, act, ws, st, ev0, ev1, Vpre, part, Vkeep,
                            Uext, u_valid, s3);
}
int pg_wino_prep_batch(int n, const pg_wino_prep* items, hipStream_t st) {
    if (n <= 0) return PG_OK;
    if (n > PG_WINO_PREP_MAX || !items) return PG_EINVAL;
    WinoPrepBatch b;
    b.n = n;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const pg_wino_prep& s = items[i];
        WinoPrepItem& t = b.it[i];
        if (!s.P || !s.U || s.Ca < 1 || s.Cb < 1) return PG_EINVAL;
        t.P = s.P;
        t.U = s.U;
        t.Ca = s.Ca;
        t.Cb = s.Cb;
        t.flip = s.flip;
        t.gx = 1;
        t.block0 = (int)blocks;
        if (s.kind == 0) {
            if (s.mo != 2 && s.mo != 3) return PG_EINVAL;
            t.kind = s.mo;
            blocks += ((long)s.Ca * s.Cb + 255) / 256;
        } else if (s.kind == 1) {
            t.kind = 10 + pg_wino2_mo();
            blocks += ((long)s.Ca * s.Cb + 255) / 256;
        } else if (s.kind == 2) {
            t.kind = 20 + pg_wino2_mo();
            t.gx = (s.Cb + 15) / 16;
            blocks += (long)t.gx * ((s.Ca + 15) / 16);
        } else {
            return PG_EINVAL;
        }
        if (blocks > 0x7fffffffL) return PG_EINVAL;
    }
    hipLaunchKernelGGL(k_wino_prep_batch, dim3((unsigned)blocks), dim3(256), 0, st, b);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

size_t pg_wino2_u_bytes(int Ca, int Cb) { return align256((size_t)wino2_nxi() * Ca * 4 * Cb * 4); }

// chunks of per-sample partial sums the output transforms emit (0: channel count not a power-of-two multiple of 4 up to 1024)
static int stats_chunks(long units_per_sample, int C) {
    const int cq = C >> 2;
    if (C % 4 != 0 || cq < 1 || cq > 256 || (256 % cq) != 0) return 0;
    return (int)((units_per_sample * cq + 255) / 256);
}
int pg_wino2_b2s_stats_chunks(int N, int Hs, int Ws, int Ca) {
    (void)N;
    const int mo = pg_wino2_mo();
    return stats_chunks((long)((Hs + mo - 1) / mo) * ((Ws + mo - 1) / mo), Ca);
}
int pg_wino2_s2b_stats_chunks(int N, int Hb, int Wb, int Cb) {
    (void)N;
    const int mo = pg_wino2_mo();
    const long TH = ((Hb + 1) / 2 + 1 + mo - 1) / mo, TW = ((Wb + 1) / 2 + 1 + mo - 1) / mo;
    return stats_chunks(TH * TW * 4, Cb);
}

// the polyphase input transform V[xi][tile][ph*Cb + b] of `big` on its own (F(3x3,2x2) tiles over the small side): what
// pg_wino2_b2s and pg_wino2_wgrad both start from -- computed once when a layer's data and weight gradients are taken together
size_t pg_wino2_v_bytes(int N, int Hs, int Ws, int Cb) {
    return align256((size_t)16 * N * ((Hs + 2) / 3) * ((Ws + 2) / 3) * 4 * Cb * 4);
}
int pg_wino2_v(const float* big, int ld_big, float* V, int N, int Hb, int Wb, int Hs, int Ws, int Cb, hipStream_t st) {
    const int TH = (Hs + 2) / 3, TW = (Ws + 2) / 3;
    const long T = (long)N * TH * TW;
    hipLaunchKernelGGL(k_wino2_v<3>, dim3((unsigned)((T * Cb + 255) / 256)), dim3(256), 0, st, big, ld_big, V, N, Hb, Wb, Cb, TH, TW);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

bool pg_wino2c_geom_ok(int N, int Hb, int Wb, int Ca, int Cb) {
    if (Ca % 32 != 0 || Cb % 4 != 0 || Ca < 64 || Cb < 32) return false;
    const long T = pg_wino2_tiles_s2b(N, Hb, Wb), X = wino2_nxi();
    if (T < 64) return false;
    if ((double)X * T * Ca * 4 >= 1.5e9 || (double)X * Ca * 4 * Cb * 4 >= 1.5e9 || (double)X * T * 4 * Cb * 4 >= 1.5e9) return false;
    return true;
}

size_t pg_wino2c_ws_bytes(int N, int Hb, int Wb, int Ca, int Cb) {
    const long T = pg_wino2_tiles_s2b(N, Hb, Wb), X = wino2_nxi();
    return align256((size_t)X * Ca * 4 * Cb * 4) + align256((size_t)X * T * Ca * 4) + align256((size_t)X * T * 4 * Cb * 4);
}

template <int MO>
static int wino2_s2b_run(const float* small, int ld_small, const float* P, const float* bias, float* big, int ld_big, int N,
                         int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int act, void* ws, hipStream_t st, hipEvent_t ev0,
                         hipEvent_t ev1, double* part, float* Uext, int u_valid, pg_epi_mul mul, int s3) {
    constexpr int X = (MO + 1) * (MO + 1);
    const int TH = ((Hb + 1) / 2 + 1 + MO - 1) / MO, TW = ((Wb + 1) / 2 + 1 + MO - 1) / MO, NC = 4 * Cb;
    const long T = (long)N * TH * TW;
    float* Uws = (float*)ws;
    float* U = Uext ? Uext : Uws;
    float* V = (float*)((char*)Uws + align256((size_t)X * Ca * NC * 4));
    float* M = (float*)((char*)V + align256((size_t)X * T * Ca * 4));
    if (!(Uext && u_valid)) {
        hipLaunchKernelGGL(k_wino2c_u<MO>, dim3((Cb + 15) / 16, (Ca + 15) / 16), dim3(256), 0, st, P, U, Ca, Cb);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    }
    hipLaunchKernelGGL(k_wino2c_v<MO>, dim3((unsigned)((T * (Ca / 4) + 255) / 256)), dim3(256), 0, st, small, ld_small, V, N, Hs, Ws,
                       Ca, TH, TW);
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    if (ev0) (void)hipEventRecord(ev0, st);
    const int a_bytes = (int)((long)X * T * Ca * 4), b_bytes = (int)((long)X * NC * Ca * 4);
    {
        const int big_rows = T >= 1024;
        const int tm = (int)(big_rows ? (T + 127) / 128 : (T + 63) / 64), tn = (NC + 127) / 128;
        const int zb = s3 ? 1 : bgemm_zb((long)tm * tn, X);
        const dim3 grid((unsigned)(tm * tn * (X / zb)));
        if (s3)
            bgemm_s3_launch(big_rows, grid, st, V, U, M, (int)T, NC, Ca, a_bytes, b_bytes, tm, tn);
        else if (big_rows && zb > 1)
            hipLaunchKernelGGL((k_wino_bgemm_mz<2, 2, 2, 2>), grid, dim3(256), 0, st, V, U, M, (int)T, NC, Ca, zb, a_bytes, b_bytes, tm, tn);
        else if (big_rows)
            hipLaunchKernelGGL((k_wino_bgemm<2, 2, 2, 2>), grid, dim3(256), 0, st, V, U, M, (int)T, NC, Ca, a_bytes, b_bytes, tm, tn);
        else if (zb > 1)
            hipLaunchKernelGGL((k_wino_bgemm_mz<1, 2, 2, 2>), grid, dim3(256), 0, st, V, U, M, (int)T, NC, Ca, zb, a_bytes, b_bytes, tm, tn);
        else
            hipLaunchKernelGGL((k_wino_bgemm<1, 2, 2, 2>), grid, dim3(256), 0, st, V, U, M, (int)T, NC, Ca, a_bytes, b_bytes, tm, tn);
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    if (part)
        hipLaunchKernelGGL(k_wino2c_out_stats<MO>, dim3((unsigned)pg_wino2_s2b_stats_chunks(N, Hb, Wb, Cb), N), dim3(256), 0, st, M, bias,
                           big, ld_big, N, Hb, Wb, Cb, TH, TW, act, part);
    else
        hipLaunchKernelGGL(k_wino2c_out<MO>, dim3((unsigned)((T * Cb + 255) / 256)), dim3(256), 0, st, M, bias, big, ld_big, N, Hb,
                           Wb, Cb, TH, TW, act, mul);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

int pg_wino2_s2b(const float* small, int ld_small, const float* P, const float* bias, float* big, int ld_big, int N, int Hb,
                 int Wb, int Hs, int Ws, int Ca, int Cb, int act, void* ws, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                 double* part, float* Uext, int u_valid, pg_epi_mul mul, int s3) {
    if (part && mul.t) return PG_EINVAL;
    if (pg_wino2_mo() == 4)
        return wino2_s2b_run<4>(small, ld_small, P, bias, big, ld_big, N, Hb, Wb, Hs, Ws, Ca, Cb, act, ws, st, ev0, ev1, part, Uext, u_valid, mul, s3);
    return wino2_s2b_run<3>(small, ld_small, P, bias, big, ld_big, N, Hb, Wb, Hs, Ws, Ca, Cb, act, ws, st, ev0, ev1, part, Uext, u_valid, mul, s3);
}

// ---- weight gradient of the stride-2 layers (polyphase F(2x2, 3x3)) ----
static long wino2w_tiles(int N, int Hs, int Ws) { return (long)N * ((Hs + 2) / 3) * ((Ws + 2) / 3); }

bool pg_wino2_wgrad_geom_ok(int N, int Hs, int Ws, int Ca, int Cb) {
    if (Ca % 4 != 0 || Cb % 4 != 0 || Ca < 64 || Cb < 32) return false;
    const long T = wino2w_tiles(N, Hs, Ws);
    if (T < 512) return false;
    if (16.0 * T * Ca * 4 >= 1.5e9 || 16.0 * T * 4 * Cb * 4 >= 1.5e9) return false;
    return true;
}

// 128x128 output tiles when they alone fill the chip, else 64x64 tiles (4x the workgroups) so that fewer, longer K slices do
bool pg_wino2_wgrad_tile64(int Ca, int Cb, int s3) {
    static const int forced = [] {
        const char* e = pg_exp_env("PATCHGAN_WINO2W_TILE");
        return e ? atoi(e) : 0;
    }();
    if (forced) return forced == 64;
    if (s3 && Ca % 128 == 0 && (4 * Cb) % 128 == 0) return false;      // (as pg_wino_wgrad_tile64)
    return 16L * ((Ca + 127) / 128) * ((4 * Cb + 127) / 128) < 768;
}

int pg_wino2_wgrad_slices(int N, int Hs, int Ws, int Ca, int Cb, int s3) {
    const long T = wino2w_tiles(N, Hs, Ws);
    const int t = pg_wino2_wgrad_tile64(Ca, Cb, s3) ? 64 : 128;
    const long wgs = 16L * ((Ca + t - 1) / t) * ((4 * Cb + t - 1) / t);
    const long fill = wgrad_s3_fill(s3, t);
    long s = (fill + wgs - 1) / wgs;
    const long nchunks = (T + KC - 1) / KC;
    if (s > nchunks / 8) s = nchunks / 8;
    return (int)(s < 1 ? 1 : s);
}

size_t pg_wino2_wgrad_ws_bytes(int N, int Hs, int Ws, int Ca, int Cb) {
    const long T = wino2w_tiles(N, Hs, Ws);
    const int sl = std::max(pg_wino2_wgrad_slices(N, Hs, Ws, Ca, Cb, 0), pg_wino2_wgrad_slices(N, Hs, Ws, Ca, Cb, 1));    // serves both forms
    return align256((size_t)16 * T * 4 * Cb * 4) + align256((size_t)16 * T * Ca * 4) + align256((size_t)sl * 16 * Ca * 4 * Cb * 4);
}

int pg_wino2_wgrad(const float* small, int ld_small, const float* big, int ld_big, float* dP, int N, int Hb, int Wb, int Hs,
                   int Ws, int Ca, int Cb, void* ws, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, const float* Vpre, int s3) {
    const int TH = (Hs + 2) / 3, TW = (Ws + 2) / 3, K = 4 * Cb;
    const long T = (long)N * TH * TW;
    // ws: V | DY | S, or DY | S when the caller supplies the transformed big-side tensor (Vpre)
    float* Vown = (float*)ws;
    float* DY = Vpre ? Vown : (float*)((char*)ws + align256((size_t)16 * T * K * 4));
    float* S = (float*)((char*)DY + align256((size_t)16 * T * Ca * 4));
    const float* V = Vpre ? Vpre : Vown;
    if (!Vpre) {
        hipLaunchKernelGGL(k_wino2_v<3>, dim3((unsigned)((T * Cb + 255) / 256)), dim3(256), 0, st, big, ld_big, Vown, N, Hb, Wb, Cb, TH,
                           TW);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    }
    hipLaunchKernelGGL(k_wino2_dy, dim3((unsigned)((T * (Ca / 4) + 255) / 256)), dim3(256), 0, st, small, ld_small, DY, N, Hs, Ws,
                       Ca, TH, TW);
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    const int slices = pg_wino2_wgrad_slices(N, Hs, Ws, Ca, Cb, s3);
    const int nchunks = (int)((T + KC - 1) / KC);
    const int cps = (nchunks + slices - 1) / slices;
    if (ev0) (void)hipEventRecord(ev0, st);
    if (pg_wino2_wgrad_tile64(Ca, Cb, s3)) {
        const int tilesA = (Ca + 63) / 64, tilesB = (K + 63) / 64;
        if (s3)
            hipLaunchKernelGGL((k_wino_wgrad_gemm_s3<1, 1, 2, 2, 2, 3>), dim3(tilesA * tilesB * 16 * slices), dim3(256), 0, st, DY, V, S, (int)T,
                               Ca, K, cps, tilesA, tilesB, 16, (int)(16L * T * Ca * 4), (int)(16L * T * K * 4));
        else
            hipLaunchKernelGGL((k_wino_wgrad_gemm<1, 1, 2, 2>), dim3(tilesA * tilesB * 16 * slices), dim3(256), 0, st, DY, V, S, (int)T, Ca,
                               K, cps, tilesA, tilesB, 16, (int)(16L * T * Ca * 4), (int)(16L * T * K * 4));
    } else {
        const int tilesA = (Ca + 127) / 128, tilesB = (K + 127) / 128;
        if (s3)
            hipLaunchKernelGGL((k_wino_wgrad_gemm_s3<2, 2, 2, 2, 1, 2>), dim3(tilesA * tilesB * 16 * slices), dim3(256), 0, st, DY, V, S, (int)T,
                               Ca, K, cps, tilesA, tilesB, 16, (int)(16L * T * Ca * 4), (int)(16L * T * K * 4));
        else
            hipLaunchKernelGGL((k_wino_wgrad_gemm<2, 2, 2, 2>), dim3(tilesA * tilesB * 16 * slices), dim3(256), 0, st, DY, V, S, (int)T, Ca,
                               K, cps, tilesA, tilesB, 16, (int)(16L * T * Ca * 4), (int)(16L * T * K * 4));
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    hipLaunchKernelGGL(k_wino2_wgrad_out, dim3((unsigned)(((long)Ca * Cb + 255) / 256)), dim3(256), 0, st, S, slices, dP, Ca, Cb);
    return hipGetLastError() == hipSuccess ? PG_OK : PG_ELAUNCH;
}

// batches per workgroup the two stride-2 paths would use (> 1: k_wino_bgemm_mz); for pg_conv_describe
int pg_wino2_b2s_zb(int N, int Hs, int Ws, int Ca) {
    const long T = wino2_tiles(N, Hs, Ws);
    const long tm = (T >= 1024) ? (T + 127) / 128 : (T + 63) / 64;
    return bgemm_zb(tm * ((Ca + 127) / 128), (int)wino2_nxi());
}
int pg_wino2_s2b_zb(int N, int Hb, int Wb, int Cb) {
    const long T = pg_wino2_tiles_s2b(N, Hb, Wb);
    const long tm = (T >= 1024) ? (T + 127) / 128 : (T + 63) / 64;
    return bgemm_zb(tm * ((4 * Cb + 127) / 128), (int)wino2_nxi());
}

#ifdef PG_TRACE_R
extern "C" int pg_debug_trace_set_w(void* buf) {
    unsigned long long* p = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(pg_trace_buf_w), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
