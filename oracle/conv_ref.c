/* conv_ref.c -- plain-C restatement of the arithmetic at the bottom of the patchGAN hot path.  TEST INFRASTRUCTURE
 * ONLY (see oracle/patchgan_oracle.py): an independent cross-check of the torch-op oracle on small shapes, built by
 * __graft_entry__.build() / `make -C oracle` into oracle/libconv_ref.so and loaded only by tests/.
 *
 * Every function follows the definition of the torch operator the reference calls (cited), written as direct loops
 * over NCHW fp32 tensors with double accumulation:
 *   conv4x4        nn.Conv2d(k=4, stride s, padding 1)          reference unet.py:19, disc.py:19,27,37,45
 *   convT4x4       nn.ConvTranspose2d(k=4, stride 2, padding 1)  reference unet.py:53
 *   instnorm       nn.InstanceNorm2d(eps=1e-5, affine=False)     reference unet.py:20,55, disc.py:32,42
 *   conv4x4_wgrad  d/dW of conv4x4 (aten::convolution_backward)  reference trainer.py:89,106
 */
#include <math.h>
#include <stddef.h>

#define IDX4(n, c, h, w, C, H, W) ((((size_t)(n) * (C) + (c)) * (H) + (h)) * (W) + (w))

/* y[N,Co,Ho,Wo] = conv(x[N,Ci,H,W], w[Co,Ci,4,4]) + b ; Ho = (H-2)/s + 1 */
void conv4x4(const float* x, const float* w, const float* b, float* y, int N, int Ci, int H, int W, int Co, int s) {
    const int Ho = (H - 2) / s + 1, Wo = (W - 2) / s + 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co)
            for (int p = 0; p < Ho; ++p)
                for (int q = 0; q < Wo; ++q) {
                    double acc = b ? b[co] : 0.0;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int kh = 0; kh < 4; ++kh) {
                            const int h = s * p - 1 + kh;
                            if (h < 0 || h >= H) continue;
                            for (int kw = 0; kw < 4; ++kw) {
                                const int ww = s * q - 1 + kw;
                                if (ww < 0 || ww >= W) continue;
                                acc += (double)x[IDX4(n, ci, h, ww, Ci, H, W)] * w[IDX4(co, ci, kh, kw, Ci, 4, 4)];
                            }
                        }
                    y[IDX4(n, co, p, q, Co, Ho, Wo)] = (float)acc;
                }
}

/* y[N,Co,2H,2W] = conv_transpose(x[N,Ci,H,W], w[Ci,Co,4,4]), stride 2, padding 1: scatter form of the definition */
void convT4x4(const float* x, const float* w, float* y, int N, int Ci, int H, int W, int Co) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (size_t i = 0; i < (size_t)N * Co * Ho * Wo; ++i) y[i] = 0.f;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co)
            for (int oh = 0; oh < Ho; ++oh)
                for (int ow = 0; ow < Wo; ++ow) {
                    double acc = 0.0;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int kh = 0; kh < 4; ++kh) {
                            const int t = oh + 1 - kh;
                            if (t < 0 || (t & 1) || t / 2 >= H) continue;
                            for (int kw = 0; kw < 4; ++kw) {
                                const int u = ow + 1 - kw;
                                if (u < 0 || (u & 1) || u / 2 >= W) continue;
                                acc += (double)x[IDX4(n, ci, t / 2, u / 2, Ci, H, W)] * w[IDX4(ci, co, kh, kw, Co, 4, 4)];
                            }
                        }
                    y[IDX4(n, co, oh, ow, Co, Ho, Wo)] = (float)acc;
                }
}

/* in place: x = (x - mean_nc) / sqrt(var_nc + eps), biased variance over H*W */
void instnorm(float* x, int N, int C, int HW, float eps) {
    for (int i = 0; i < N * C; ++i) {
        float* p = x + (size_t)i * HW;
        double m = 0.0, v = 0.0;
        for (int k = 0; k < HW; ++k) m += p[k];
        m /= HW;
        for (int k = 0; k < HW; ++k) v += (p[k] - m) * (p[k] - m);
        v /= HW;
        const double r = 1.0 / sqrt(v + (double)eps);
        for (int k = 0; k < HW; ++k) p[k] = (float)((p[k] - m) * r);
    }
}

/* dw[Co,Ci,4,4] = sum_{n,p,q} dy[n,co,p,q] * x[n,ci,s*p-1+kh,s*q-1+kw] */
void conv4x4_wgrad(const float* x, const float* dy, float* dw, int N, int Ci, int H, int W, int Co, int s) {
    const int Ho = (H - 2) / s + 1, Wo = (W - 2) / s + 1;
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int kh = 0; kh < 4; ++kh)
                for (int kw = 0; kw < 4; ++kw) {
                    double acc = 0.0;
                    for (int n = 0; n < N; ++n)
                        for (int p = 0; p < Ho; ++p) {
                            const int h = s * p - 1 + kh;
                            if (h < 0 || h >= H) continue;
                            for (int q = 0; q < Wo; ++q) {
                                const int ww = s * q - 1 + kw;
                                if (ww < 0 || ww >= W) continue;
                                acc += (double)dy[IDX4(n, co, p, q, Co, Ho, Wo)] * x[IDX4(n, ci, h, ww, Ci, H, W)];
                            }
                        }
                    dw[IDX4(co, ci, kh, kw, Ci, 4, 4)] = (float)acc;
                }
}
