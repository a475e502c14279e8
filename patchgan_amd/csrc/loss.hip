// loss.hip -- GAN + segmentation losses of the patchGAN step, value and gradient seed, on the device
// (no host round trip inside the step).  Tiny, HBM-bound reductions; fp64 accumulators, fixed-order trees.
//
//   pg_loss_reduce   per-(n,c) sums over HW of {y*p, y, p, bce_elem, |p-y|}          (stage 1)
//   pg_loss_prepare  sum_n (1 - T_n), sum y                                          (stage 1b; all-reduced under DP)
//   pg_loss_finalize loss value + per-(n,c) gradient coefficients                    (stage 2)
//   pg_loss_grad     elementwise gradient wrt p                                      (stage 3)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "patchgan_hip.h"
#include "pg_common.h"

namespace {

// grid (N*C, nsplit): block (nc, z) reduces pixels z, z+nsplit, ... in 256-pixel strides; with nsplit > 1 the partial
// sums land in Spart[z][nc][5] and k_loss_combine adds them in z order
__global__ __launch_bounds__(256) void k_loss_reduce(const float* __restrict__ p, int ld_p, const float* __restrict__ y,
                                                     int ld_y, float tconst, int HW, int C, double* __restrict__ S) {
    __shared__ double red[5][256];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / C, c = blockIdx.x % C;
    const float* pb = p + (long)n * HW * ld_p + c;
    const float* yb = y ? y + (long)n * HW * ld_y + c : nullptr;
    double s[5] = {0, 0, 0, 0, 0};
    for (int i = blockIdx.y * 256 + tid; i < HW; i += 256 * gridDim.y) {
        const float pv = pb[(long)i * ld_p];
        const float yv = yb ? yb[(long)i * ld_y] : tconst;
        // F.binary_cross_entropy clamps both logs at -100
        const float lp = fmaxf(logf(pv), -100.f), lq = fmaxf(log1pf(-pv), -100.f);
        s[0] += (double)(yv * pv);
        s[1] += (double)yv;
        s[2] += (double)pv;
        s[3] += (double)(-(yv * lp + (1.f - yv) * lq));
        s[4] += (double)fabsf(pv - yv);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) red[k][tid] = s[k];
    for (int off = 128; off > 0; off >>= 1) {
        __syncthreads();
        if (tid < off) {
#pragma unroll
            for (int k = 0; k < 5; ++k) red[k][tid] += red[k][tid + off];
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) S[((long)blockIdx.y * gridDim.x + blockIdx.x) * 5 + k] = red[k][0];
    }
}

// C == 4: a block owns (n, z) and all four channels of its pixels -- every byte of a fetched line is used by the block that fetched
// it, instead of four blocks each touching 4 of every 16 (28) bytes.  Same output layout as k_loss_reduce with gridDim.x = N * 4.
__global__ __launch_bounds__(256) void k_loss_reduce_c4(const float* __restrict__ p, int ld_p, const float* __restrict__ y, int ld_y,
                                                        float tconst, int HW, int N, double* __restrict__ S) {
    __shared__ double red[20][256];
    const int tid = threadIdx.x, n = blockIdx.x;
    const float* pb = p + (long)n * HW * ld_p;
    const float* yb = y ? y + (long)n * HW * ld_y : nullptr;
    double s[4][5];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) s[c][k] = 0.0;
    for (int i = blockIdx.y * 256 + tid; i < HW; i += 256 * gridDim.y) {
        float pv4[4], yv4[4];       // (the generator output is a 4-channel slice of the 7-channel discriminator input: scalar loads)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            pv4[c] = pb[(long)i * ld_p + c];
            yv4[c] = yb ? yb[(long)i * ld_y + c] : tconst;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float pv = pv4[c], yv = yv4[c];
            const float lp = fmaxf(logf(pv), -100.f), lq = fmaxf(log1pf(-pv), -100.f);
            s[c][0] += (double)(yv * pv);
            s[c][1] += (double)yv;
            s[c][2] += (double)pv;
            s[c][3] += (double)(-(yv * lp + (1.f - yv) * lq));
            s[c][4] += (double)fabsf(pv - yv);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) red[c * 5 + k][tid] = s[c][k];
    for (int off = 128; off > 0; off >>= 1) {
        __syncthreads();
        if (tid < off) {
#pragma unroll
            for (int k = 0; k < 20; ++k) red[k][tid] += red[k][tid + off];
        }
    }
    if (tid < 20) S[((long)blockIdx.y * N * 4 + (long)n * 4) * 5 + tid] = red[tid][0];
}

__global__ void k_loss_combine(const double* __restrict__ Spart, int nsplit, int n5, double* __restrict__ S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n5) return;
    double v = 0.0;
    for (int z = 0; z < nsplit; ++z) v += Spart[(long)z * n5 + i];
    S[i] = v;
}

__device__ inline void tversky_terms(const double* S, int n, int C, double beta, double& tp, double& denom) {
    double syp = 0, sy = 0, sp = 0;
    for (int c = 0; c < C; ++c) {
        const double* s = S + ((long)n * C + c) * 5;
        syp += s[0];
        sy += s[1];
        sp += s[2];
    }
    tp = syp;
    const double fn = sy - syp, fp = sp - syp;
    denom = tp + beta * fn + (1.0 - beta) * fp + 1.0;
}

__global__ void k_loss_prepare(const double* __restrict__ S, int N, int C, float beta, double* __restrict__ local2) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double acc = 0, sy = 0;
    for (int n = 0; n < N; ++n) {
        double tp, denom;
        tversky_terms(S, n, C, (double)beta, tp, denom);
        acc += 1.0 - (tp + 1.0) / denom;
        for (int c = 0; c < C; ++c) sy += S[((long)n * C + c) * 5 + 1];
    }
    local2[0] = acc;
    local2[1] = sy;
}

// the two batch-global terms of pg_loss_prepare, in its summation order
__device__ inline void loss_sums(const double* S, int N, int C, double beta, double& acc, double& sy) {
    acc = 0;
    sy = 0;
    for (int n = 0; n < N; ++n) {
        double tp, denom;
        tversky_terms(S, n, C, beta, tp, denom);
        acc += 1.0 - (tp + 1.0) / denom;
        for (int c = 0; c < C; ++c) sy += S[((long)n * C + c) * 5 + 1];
    }
}

__global__ void k_loss_finalize(const double* __restrict__ S, const double* __restrict__ gsum2_in, int mode, int N, int C,
                                int HW, int Bglobal, float alpha, float beta, float gamma, float* __restrict__ coef,
                                float* __restrict__ loss_out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double cnt = (double)Bglobal * C * HW;
    double gsum2[2] = {0, 0};
    if (gsum2_in) {
        gsum2[0] = gsum2_in[0];
        gsum2[1] = gsum2_in[1];
    } else if (mode == PG_LOSS_TVERSKY || mode == PG_LOSS_WBCE) {
        loss_sums(S, N, C, (double)beta, gsum2[0], gsum2[1]);
    }
    if (mode == PG_LOSS_TVERSKY) {
        const double m = gsum2[0] / (double)Bglobal;
        *loss_out = (float)((double)alpha * pow(m, (double)gamma));
        // d/dp_i [alpha * m^gamma] = alpha*gamma*m^(gamma-1)/B * d(1-T_n)/dp_i ;  dT/dp_i = (y_i*D - (tp+1)*(1-beta))/D^2
        const double Kf = (double)alpha * (double)gamma * pow(m, (double)gamma - 1.0) / (double)Bglobal;
        for (int n = 0; n < N; ++n) {
            double tp, denom;
            tversky_terms(S, n, C, (double)beta, tp, denom);
            const float c1 = (float)(-Kf / denom);
            const float c0 = (float)(Kf * (tp + 1.0) * (1.0 - (double)beta) / (denom * denom));
            for (int c = 0; c < C; ++c) {
                coef[((long)n * C + c) * 2 + 0] = c1;
                coef[((long)n * C + c) * 2 + 1] = c0;
            }
        }
    } else if (mode == PG_LOSS_WBCE) {
        double acc = 0;
        for (int n = 0; n < N; ++n)
            for (int c = 0; c < C; ++c) {
                const double* s = S + ((long)n * C + c) * 5;
                // weight computed in fp32 like the reference: 1 - sum_hw(y)/sum(y)   (trainer.py:77)
                const float w = (C > 1) ? 1.f - (float)s[1] / (float)gsum2[1] : 1.f;
                acc += (double)w * s[3];
                coef[((long)n * C + c) * 2 + 0] = (float)((double)alpha * (double)w / cnt);
                coef[((long)n * C + c) * 2 + 1] = 0.f;
            }
        *loss_out = (float)((double)alpha * acc / cnt);
    } else {
        const int k = (mode == PG_LOSS_MAE) ? 4 : 3;
        double acc = 0;
        for (int n = 0; n < N; ++n)
            for (int c = 0; c < C; ++c) {
                acc += S[((long)n * C + c) * 5 + k];
                coef[((long)n * C + c) * 2 + 0] = (float)((double)alpha / cnt);
                coef[((long)n * C + c) * 2 + 1] = 0.f;
            }
        *loss_out = (float)((double)alpha * acc / cnt);
    }
}

// The gradient pass of a loss term, shared by k_loss_grad (coef in global memory) and k_loss_fused (coef in LDS): one thread per PIXEL
// and trip, 32-bit pixel decode (the per-ELEMENT 64-bit i % C, i / C, pix / HW of the first version cost more than the memory
// traffic: 35 us for 1 M elements), the four channels of a C == 4 pixel as one 16-byte load / store where the views allow it.
// Element arithmetic unchanged (bit-identical results).
__device__ __forceinline__ float loss_grad_elem(float pv, float yv, float k0, float k1, int gmode) {
    if (gmode == 0) return k0 * yv + k1;
    if (gmode == 1) return k0 * (pv - yv) / fmaxf((1.f - pv) * pv, 1e-12f);      // torch binary_cross_entropy_backward
    const float d = pv - yv;
    return k0 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
}
template <typename CoefPtr>
__device__ __forceinline__ void loss_grad_pass(const float* __restrict__ p, int ld_p, const float* __restrict__ y, int ld_y, float tconst,
                                               CoefPtr coef, float* __restrict__ g, int ld_g, int N, int HW, int C, int gmode) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const long npix = (long)N * HW;
    auto al = [](const void* q, int ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0; };
    if (npix < 0x7fffffffL) {
        const unsigned np = (unsigned)npix, stride = gridDim.x * blockDim.x, hw = (unsigned)HW;
        const bool vp = al(p, ld_p), vy = y && al(y, ld_y), vg = al(g, ld_g);
        for (unsigned pix = blockIdx.x * blockDim.x + threadIdx.x; pix < np; pix += stride) {
            const unsigned n = pix / hw;
            const float* pp = p + (size_t)pix * ld_p;
            const float* yp = y ? y + (size_t)pix * ld_y : nullptr;
            float* gp = g + (size_t)pix * ld_g;
            if (C == 4) {       // (a slice of a wider pixel -- the mask channels of the discriminator-input buffer -- is accessed by element)
                const f4 pv = vp ? *reinterpret_cast<const f4*>(pp) : f4{pp[0], pp[1], pp[2], pp[3]};
                const f4 yv = !yp ? f4{tconst, tconst, tconst, tconst} : vy ? *reinterpret_cast<const f4*>(yp) : f4{yp[0], yp[1], yp[2], yp[3]};
                f4 r;
#pragma unroll
                for (int c = 0; c < 4; ++c) r[c] = loss_grad_elem(pv[c], yv[c], coef[(n * 4 + c) * 2 + 0], coef[(n * 4 + c) * 2 + 1], gmode);
                if (vg) {
                    *reinterpret_cast<f4*>(gp) = r;
                } else {
                    gp[0] = r[0];
                    gp[1] = r[1];
                    gp[2] = r[2];
                    gp[3] = r[3];
                }
            } else {
                for (int c = 0; c < C; ++c)
                    gp[c] = loss_grad_elem(pp[c], yp ? yp[c] : tconst, coef[(n * C + c) * 2 + 0], coef[(n * C + c) * 2 + 1], gmode);
            }
        }
        return;
    }
    const long total = npix * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const long n = pix / HW;
        g[pix * ld_g + c] = loss_grad_elem(p[pix * ld_p + c], y ? y[pix * ld_y + c] : tconst, coef[(n * C + c) * 2 + 0],
                                           coef[(n * C + c) * 2 + 1], gmode);
    }
}

__global__ void k_loss_grad(const float* __restrict__ p, int ld_p, const float* __restrict__ y, int ld_y, float tconst,
                            const float* __restrict__ coef, float* __restrict__ g, int ld_g, int N, int HW, int C,
                            int mode) {
    loss_grad_pass(p, ld_p, y, ld_y, tconst, coef, g, ld_g, N, HW, C, mode);
}

// Stages 1 (tail) + 1b + 2 + 3 in ONE launch (no data parallelism between them): every workgroup first rebuilds the small per-(n, c)
// table -- the five sums (adding pg_loss_reduce's nsplit partial slabs in z order, as k_loss_combine does), the two batch-global terms
// (gsum2, or computed here in pg_loss_prepare's order when gsum2 == NULL) and the gradient coefficients, all with the arithmetic of
// k_loss_combine / k_loss_prepare / k_loss_finalize -- in LDS, workgroup 0 also writes the loss value and the summed S, then the grid
// strides over the gradient elements like k_loss_grad.  N * C <= FUSED_MAX_NC; the grid is small (<= 512 workgroups) so that the
// redundant table builds stay a few MB of L2 reads.
constexpr int FUSED_MAX_NC = 256;
__global__ __launch_bounds__(256) void k_loss_fused(const double* __restrict__ Spart, int nsplit, double* __restrict__ S_out,
                                                    const double* __restrict__ gsum2_in, int lmode, int N, int C, int HW, int Bglobal,
                                                    float alpha, float beta, float gamma, const float* __restrict__ p, int ld_p,
                                                    const float* __restrict__ y, int ld_y, float tconst, float* __restrict__ g, int ld_g,
                                                    float* __restrict__ loss_out) {
    __shared__ double S[FUSED_MAX_NC * 5];
    __shared__ float coef[FUSED_MAX_NC * 2];
    __shared__ double gs[2];
    const int tid = threadIdx.x, NC = N * C, n5 = NC * 5;
    for (int i = tid; i < n5; i += 256) {
        // the slabs in z order (k_loss_combine's order), 16 loads in flight at a time: one dependent load per slab made this loop the
        // longest part of the kernel (128 slabs = 128 L2 latencies per workgroup)
        double v = 0.0;
        int z = 0;
        for (; z + 16 <= nsplit; z += 16) {
            double t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = Spart[(long)(z + k) * n5 + i];
#pragma unroll
            for (int k = 0; k < 16; ++k) v += t[k];
        }
        for (; z < nsplit; ++z) v += Spart[(long)z * n5 + i];
        S[i] = v;
        if (blockIdx.x == 0 && S_out) S_out[i] = v;
    }
    __syncthreads();
    if (tid == 0) {
        if (gsum2_in) {
            gs[0] = gsum2_in[0];
            gs[1] = gsum2_in[1];
        } else if (lmode == PG_LOSS_TVERSKY || lmode == PG_LOSS_WBCE) {
            loss_sums(S, N, C, (double)beta, gs[0], gs[1]);
        } else {
            gs[0] = gs[1] = 0.0;
        }
    }
    __syncthreads();
    const double cnt = (double)Bglobal * C * HW;
    if (lmode == PG_LOSS_TVERSKY) {
        const double m = gs[0] / (double)Bglobal;
        const double Kf = (double)alpha * (double)gamma * pow(m, (double)gamma - 1.0) / (double)Bglobal;
        for (int n = tid; n < N; n += 256) {
            double tp, denom;
            tversky_terms(S, n, C, (double)beta, tp, denom);
            const float c1 = (float)(-Kf / denom);
            const float c0 = (float)(Kf * (tp + 1.0) * (1.0 - (double)beta) / (denom * denom));
            for (int c = 0; c < C; ++c) {
                coef[(n * C + c) * 2 + 0] = c1;
                coef[(n * C + c) * 2 + 1] = c0;
            }
        }
        if (blockIdx.x == 0 && tid == 0) *loss_out = (float)((double)alpha * pow(m, (double)gamma));
    } else {
        for (int i = tid; i < NC; i += 256) {
            float w = 1.f;
            if (lmode == PG_LOSS_WBCE && C > 1) w = 1.f - (float)S[i * 5 + 1] / (float)gs[1];
            coef[i * 2 + 0] = (lmode == PG_LOSS_WBCE) ? (float)((double)alpha * (double)w / cnt) : (float)((double)alpha / cnt);
            coef[i * 2 + 1] = 0.f;
        }
        if (blockIdx.x == 0 && tid == 0) {          // the value: k_loss_finalize's sums, in its (n, c) order
            const int k = (lmode == PG_LOSS_MAE) ? 4 : 3;
            double acc = 0;
            for (int i = 0; i < NC; ++i) {
                float w = 1.f;
                if (lmode == PG_LOSS_WBCE && C > 1) w = 1.f - (float)S[i * 5 + 1] / (float)gs[1];
                acc += (lmode == PG_LOSS_WBCE) ? (double)w * S[i * 5 + 3] : S[i * 5 + k];
            }
            *loss_out = (float)((double)alpha * acc / cnt);
        }
    }
    __syncthreads();
    if (!g) return;
    const int gmode = lmode == PG_LOSS_TVERSKY ? 0 : lmode == PG_LOSS_MAE ? 2 : 1;
    loss_grad_pass(p, ld_p, y, ld_y, tconst, (const float*)coef, g, ld_g, N, HW, C, gmode);
}

// pixel splits per (n, c) -- per n for the four-channel kernel -- so that a loss over a large map runs on >= ~256 workgroups
inline bool loss_c4(int HW, int C) { return C == 4 && HW >= 4096; }
int loss_nsplit(int N, int HW, int C) {
    const bool c4 = loss_c4(HW, C);
    int ns = c4 ? 1024 / N : 256 / (N * C);      // (the four-channel kernel is VALU-bound -- two logs, 20 fp64 sums per pixel: 4 blocks per CU)
    const int maxs = HW / 1024;
    if (ns > maxs) ns = maxs;
    if (ns > (c4 ? 128 : 64)) ns = c4 ? 128 : 64;
    return ns < 1 ? 1 : ns;
}

}  // namespace

extern "C" {

int pg_loss_fused_max_nc(void) { return FUSED_MAX_NC; }

long pg_loss_reduce_doubles(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    const int ns = loss_nsplit(N, HW, C);
    return (long)N * C * 5 * (ns > 1 ? 1 + ns : 1);
}

int pg_loss_reduce(const float* p, int ld_p, const float* y, int ld_y, float tconst, int N, int HW, int C, double* S,
                   void* stream) {
    if (!p || !S || N <= 0 || HW <= 0 || C <= 0 || ld_p < C || (y && ld_y < C)) return PG_EINVAL;
    // S must hold (1 + nsplit) * N*C*5 doubles when nsplit > 1 (pg_loss_reduce_doubles): partials behind the result
    const int nsplit = loss_nsplit(N, HW, C);
    if (loss_c4(HW, C) && nsplit > 1) {
        double* part = S + (long)N * C * 5;
        hipLaunchKernelGGL(k_loss_reduce_c4, dim3(N, nsplit), dim3(256), 0, (hipStream_t)stream, p, ld_p, y, ld_y, tconst, HW, N, part);
        if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
        hipLaunchKernelGGL(k_loss_combine, dim3((N * C * 5 + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, nsplit, N * C * 5, S);
        return pg_launch_status();
    }
    if (nsplit == 1) {
        hipLaunchKernelGGL(k_loss_reduce, dim3(N * C, 1), dim3(256), 0, (hipStream_t)stream, p, ld_p, y, ld_y, tconst, HW, C, S);
        return pg_launch_status();
    }
    double* part = S + (long)N * C * 5;
    hipLaunchKernelGGL(k_loss_reduce, dim3(N * C, nsplit), dim3(256), 0, (hipStream_t)stream, p, ld_p, y, ld_y, tconst, HW, C,
                       part);
    if (hipGetLastError() != hipSuccess) return PG_ELAUNCH;
    hipLaunchKernelGGL(k_loss_combine, dim3((N * C * 5 + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, nsplit, N * C * 5,
                       S);
    return pg_launch_status();
}

/* pg_loss_reduce without its combine launch: S receives the nsplit partial slabs [z][N*C][5] (nsplit = the return value >= 1; with
 * nsplit == 1 that IS the result) -- for pg_loss_value_grad, which adds them itself. */
int pg_loss_reduce_parts(const float* p, int ld_p, const float* y, int ld_y, float tconst, int N, int HW, int C, double* S,
                         void* stream) {
    if (!p || !S || N <= 0 || HW <= 0 || C <= 0 || ld_p < C || (y && ld_y < C)) return PG_EINVAL;
    const int nsplit = loss_nsplit(N, HW, C);
    if (loss_c4(HW, C) && nsplit > 1)
        hipLaunchKernelGGL(k_loss_reduce_c4, dim3(N, nsplit), dim3(256), 0, (hipStream_t)stream, p, ld_p, y, ld_y, tconst, HW, N, S);
    else
        hipLaunchKernelGGL(k_loss_reduce, dim3(N * C, nsplit), dim3(256), 0, (hipStream_t)stream, p, ld_p, y, ld_y, tconst, HW, C, S);
    return hipGetLastError() == hipSuccess ? nsplit : PG_ELAUNCH;
}

int pg_loss_value_grad(const double* Spart, int nsplit, double* S_out, const double* gsum2, int mode, int N, int C, int HW,
                       int Bglobal, float alpha, float beta, float gamma, const float* p, int ld_p, const float* y, int ld_y,
                       float tconst, float* g, int ld_g, float* loss_out, void* stream) {
    if (!Spart || !loss_out || nsplit < 1 || N <= 0 || C <= 0 || HW <= 0 || Bglobal < N || N * C > FUSED_MAX_NC) return PG_EINVAL;
    if (mode < PG_LOSS_TVERSKY || mode > PG_LOSS_BCE) return PG_EINVAL;
    if (g && (!p || ld_p < C || ld_g < C || (y && ld_y < C))) return PG_EINVAL;
    long b = g ? ((long)N * HW * C + 2047) / 2048 : 1;      // >= 8 elements per thread: the per-workgroup table build stays cheap
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(k_loss_fused, dim3((int)b), dim3(256), 0, (hipStream_t)stream, Spart, nsplit, S_out, gsum2, mode, N, C, HW,
                       Bglobal, alpha, beta, gamma, p, ld_p, y, ld_y, tconst, g, ld_g, loss_out);
    return pg_launch_status();
}

int pg_loss_prepare(const double* S, int N, int C, float beta, double* local2, void* stream) {
    if (!S || !local2 || N <= 0 || C <= 0) return PG_EINVAL;
    hipLaunchKernelGGL(k_loss_prepare, dim3(1), dim3(64), 0, (hipStream_t)stream, S, N, C, beta, local2);
    return pg_launch_status();
}

int pg_loss_finalize(const double* S, const double* gsum2, int mode, int N, int C, int HW, int Bglobal, float alpha,
                     float beta, float gamma, float* coef, float* loss_out, void* stream) {
    if (!S || !coef || !loss_out || N <= 0 || C <= 0 || HW <= 0 || Bglobal < N) return PG_EINVAL;      // gsum2 == NULL: computed here
    if (mode < PG_LOSS_TVERSKY || mode > PG_LOSS_BCE) return PG_EINVAL;
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, S, gsum2, mode, N, C, HW, Bglobal,
                       alpha, beta, gamma, coef, loss_out);
    return pg_launch_status();
}

int pg_loss_grad(const float* p, int ld_p, const float* y, int ld_y, float tconst, const float* coef, float* g,
                 int ld_g, int N, int HW, int C, int mode, void* stream) {
    if (!p || !coef || !g || N <= 0 || HW <= 0 || C <= 0 || ld_p < C || ld_g < C || (y && ld_y < C)) return PG_EINVAL;
    if (mode < 0 || mode > 2) return PG_EINVAL;
    const long total = (long)N * HW * C;
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(k_loss_grad, dim3((int)b), dim3(256), 0, (hipStream_t)stream, p, ld_p, y, ld_y, tconst, coef, g,
                       ld_g, N, HW, C, mode);
    return pg_launch_status();
}

}  // extern "C"
