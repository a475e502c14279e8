/*
 * patchgan_hip.h -- C ABI of libpatchgan_hip.so (gfx950 / MI355X only).
 *
 * The reference (ramanakumars/patchGAN v0.2.2) has no native layer of its own:
 * every entry point below replaces the ATen operator that a line of the
 * reference's Python invokes, cited as file:line into /root/reference.
 *
 * Conventions
 *   - every tensor is fp32, device memory, NHWC ("pixel-major"): element
 *     (n, h, w, c) of a tensor with pixel stride `ld` lives at
 *     base[((n*H + h)*W + w)*ld + c].  `ld >= C` lets a tensor be a channel
 *     slice of a wider buffer, which is how torch.cat (unet.py:127,
 *     trainer.py:65,96,98) is made free.  Base pointers, `ld` and channel
 *     counts must be multiples of 4 floats (16 B) unless stated otherwise.
 *   - 4x4 convolution weights are kept in ONE packed layout for all uses:
 *         P[tap = kh*4 + kw][a][b]          (a*b floats per tap, b fastest)
 *     where a torch weight tensor W[a][b][kh][kw] has (a, b) = (Cout, Cin) for
 *     nn.Conv2d and (Cin, Cout) for nn.ConvTranspose2d.  Channel dim `a` lives
 *     on the SMALL spatial side (conv output / convT input), `b` on the BIG side
 *     (conv input / convT output);  big = stride*small - 1 + k.
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*),
 *     never allocate, never synchronise; the caller owns all memory including
 *     the workspace.  Return 0 on success or a negative PG_E* code; nothing is
 *     launched when an error is returned.
 *   - thread-compatible: no global mutable state (pg_conv_time_next keeps two thread-local event handles).
 */
#ifndef PATCHGAN_HIP_H
#define PATCHGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK 0
#define PG_EINVAL (-1)   /* bad shape / null pointer / misaligned pointer or ld */
#define PG_EWORKSPACE (-2) /* workspace too small */
#define PG_ELAUNCH (-3)  /* hipGetLastError() != hipSuccess after a launch */

/* activation codes (unet.py:12-17,42-51; disc.py:20,29,39,46) */
#define PG_ACT_NONE 0
#define PG_ACT_LEAKY 1   /* LeakyReLU(0.2) */
#define PG_ACT_RELU 2
#define PG_ACT_TANH 3
#define PG_ACT_SIGMOID 4

/* algorithm selector for the three conv entry points */
#define PG_ALGO_AUTO 0   /* fastest fp32 kernel for the geometry: PG_ALGO_MFMA, except Winograd on MFMA (fp32; 1.5e-6 .. 4e-6
                            relative error instead of 1e-7) for wide stride-1 layers (F(2x2,4x4) / F(3x3,4x4); weight gradient
                            F(4x4,2x2)) and channel-heavy stride-2 layers (polyphase F(3x3,2x2)) */
#define PG_ALGO_DIRECT 1 /* one-thread-per-output reference-quality kernels (any channel count) */
#define PG_ALGO_MFMA 2   /* LDS-tiled implicit GEMM on v_mfma_f32_32x32x2_f32 (channels % 4 == 0) */
#define PG_ALGO_BF16 3   /* same kernels with operand tiles rounded to bf16 in LDS and v_mfma_f32_32x32x16_bf16 (fp32
                            tensors, fp32 accumulate); layers the fast path does not cover fall back to fp32 */

#define PG_ALGO_MASK 0xF
/* Per-call tuning bits, OR-ed into the `algo` argument of the three conv entry points (and into pg_conv_describe's op as
 * op + 16 * (PG_ALGO_* | PG_TUNE_*)).  They override the size heuristics of PG_ALGO_AUTO (and the process-wide PATCHGAN_*
 * environment defaults that exist for A/B timing); results stay inside the stated per-kernel tolerances either way. */
#define PG_TUNE_WINO2_ALL 0x010   /* polyphase Winograd forward / data gradient on every stride-2 layer the geometry allows */
#define PG_TUNE_WINO2_OFF 0x020   /* ... on none */
#define PG_TUNE_WINO2W_ALL 0x040  /* polyphase Winograd weight gradient on every stride-2 layer the geometry allows */
#define PG_TUNE_WINO2W_OFF 0x080  /* ... on none */
#define PG_TUNE_WINO_OFF 0x100    /* no Winograd path at all: the kernels of PG_ALGO_MFMA */
#define PG_TUNE_WINOW_OFF 0x200   /* stride-1 weight gradient on the implicit GEMM */
#define PG_TUNE_WINO1_F2 0x400    /* stride-1 forward / data gradient: F(2x2,4x4) tiles */
#define PG_TUNE_WINO1_F3 0x800    /* ... F(3x3,4x4) tiles (needs Cin % 64 == 0) */
#define PG_TUNE_WINO_DMA 0x1000   /* stride-1 64-tile Winograd GEMMs staged by LDS-DMA (k_wino_gemm_dma) instead of registers */
#define PG_TUNE_BF16X_OFF 0x2000  /* PG_ALGO_BF16 on bf16 tensors: the register-staged k_*_bf16 kernels instead of the LDS-DMA
                                     kernels of conv_bf16.hip (k_conv_bf16x; input channels % 64 == 0) */
#define PG_TUNE_BF16X_RING 0x4000 /* k_conv_bf16x staging pinned: three-stage LDS ring of 32-wide K chunks (DMA two chunks ahead) */
#define PG_TUNE_BF16X_FLAT 0x8000 /* ... one LDS buffer of 64-wide K chunks */
#define PG_TUNE_S3_OFF 0x40000    /* polyphase Winograd GEMMs of PG_ALGO_AUTO on v_mfma_f32_32x32x2_f32 (k_wino_bgemm) instead of their default
                                     split-bf16 form k_wino_bgemm_s3: every fp32 operand value split into three bf16 pieces (8 + 8 + 8
                                     significand bits, exact) while it is staged, the six products with combined weight >= 2^-16 summed by
                                     v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- fp32-grade results (the dropped terms are <= 2^-25
                                     relative; per-kernel error against float64 as for the fp32 kernel) at 2.7x fewer matrix-pipe cycles.
                                     The bit also selects the fp32-MFMA forms of the other two split-bf16 GEMM families: the Winograd weight
                                     gradients (k_wino_wgrad_gemm instead of k_wino_wgrad_gemm_s3) and the row-fused stride-1 F(3x3,4x4) GEMM
                                     (k_wino_gemm_row instead of k_wino_gemm_row_s3) */

/* bf16 activation storage on the PG_ALGO_BF16 kernels: OR-ed into `algo` like the PG_TUNE_* bits.  The tensor named carries bf16
 * elements (NHWC, `ld` in bf16 elements, 8-byte-aligned base; 16-byte-aligned base and ld % 8 == 0 for the LDS-DMA kernels); weights,
 * biases, weight gradients and split-K slabs stay fp32.
 * pg_conv4x4_big2small: BIG = input, SMALL = output; pg_conv4x4_small2big: SMALL = input, BIG = output; pg_conv4x4_wgrad: both
 * or neither (dbias allowed with Ca % 4 == 0).  Honoured where a bf16 kernel covers the call -- input channels % 4 == 0 and >= 32 (the
 * register-staged kernels), % 64 == 0 (the LDS-DMA kernels of conv_bf16.hip), or a <= 8-channel `big` in 8-channel pixels (ld_big == 8,
 * pad channels zero: pg_pad8_bf16); pg_conv4x4_small2big onto <= 8 channels takes PG_IO_SMALL_BF16 alone and writes fp32.  Any other
 * combination returns PG_EINVAL. */
#define PG_IO_BIG_BF16 0x10000
#define PG_IO_SMALL_BF16 0x20000
#define PG_IO_MASK 0x30000

typedef struct pg_conv_geom {
    int N;        /* batch */
    int Hb, Wb;   /* big spatial extent  (conv input  / convT output) */
    int Hs, Ws;   /* small spatial extent (conv output / convT input)  */
    int Ca;       /* channels on the small side = weight dim a */
    int Cb;       /* channels on the big side   = weight dim b */
    int stride;   /* 1 or 2; kernel 4x4, padding 1 always (unet.py:80-81, disc.py:16-17) */
} pg_conv_geom;

int pg_version(void);

/* Byte extent (pixels * ld * element size of a tensor's view) below which the buffer-load kernels -- every fast fp32 kernel and
 * ALL kernels that take bf16 tensors (PG_IO_*) -- can address a tensor: their gathers use 32-bit byte offsets with an
 * out-of-range sentinel.  fp32 tensors beyond it run on the generic kernels (64-bit addressing); a bf16 tensor beyond it has no
 * kernel (PG_EINVAL), and the size queries below (pg_conv_u_bytes, pg_conv_stats_chunks, pg_conv_v_bytes, pg_conv_mul_ok), which
 * see the geometry only, do not know the views: a caller that plans hand-overs or bf16 storage checks its views against this
 * number first (patchgan_amd.engine does: ConvOp.fits / _bf16_tensors_ok; tiled inference streams its tiles in bounded batches). */
size_t pg_conv_max_tensor_bytes(void);

/* Bytes of workspace that lets op (0 = big2small, 1 = small2big, 2 = wgrad) use its preferred
 * split-K factor / Winograd path for geometry g under ANY algo / PG_TUNE_* combination.  A smaller (or NULL) workspace is
 * legal: the split shrinks, the Winograd paths fall back to the implicit GEMM. */
size_t pg_conv_workspace_bytes(const pg_conv_geom* g, int op);   /* op 3 = pg_conv4x4_bwd_big */

/* Reports which kernel the MFMA path of op would launch for g with a workspace of ws_bytes: tile_id
 * (0 = 128x128, 1 = 128x64, 2 = 128x32, 3 = 64x128, 4 = 64x64 output tile per workgroup; the kernel symbol is
 * k_big2small / k_small2big / k_wgrad <MR,NR,WM,WN> with <2,2,2,2>, <2,1,2,2>, <1,1,4,1>, <1,2,2,2>, <1,1,2,2>), the
 * split-K factor and the number of workgroups.  For op 2, tile_id + 10 / + 20 means the taps-folded-into-N kernel
 * k_wgrad_tapn<..., 1> / <..., 2> (few big-side / small-side channels); for ops 0/1, + 30 means the row-GEMM +
 * col2im / tap-gather path.  + 100 (+ 200: power-of-two pixel decode, wgrad) marks the fast buffer-load variant
 * (k_b2s_fast / k_s2b_fast / k_wgrad_fast) that runs when tensors are 16-byte aligned.  `op` may carry the algorithm as
 * op + 16 * PG_ALGO_*; for PG_ALGO_AUTO (op < 16) wide stride-1 layers report the Winograd kernels instead: + 40 / + 50 =
 * k_wino_gemm<2,1,2,2,2,2> / <1,1,2,2,4,2>, + 90 = its F(3x3,4x4) instance <1,1,2,2,2,3> (ops 0/1; tile / split of the implicit-GEMM plan otherwise unchanged), 60 =
 * k_wino_wgrad_gemm<2,2,2,2> with split = its K slices (op 2; 63: 64x64 tiles <1,1,2,2>; 61 / 62: its polyphase stride-2 form with 128x128 / 64x64 tiles; by default the split-bf16
 * forms k_wino_wgrad_gemm_s3<2,2,2,2,1,2> / <1,1,2,2,2,3>, and k_wino_gemm_row_s3<2> for + 90's row-fused form), 70 / 71 = k_wino_bgemm_s3<2,2,2,2,2> / <1,2,2,2,3>
 * (under PG_TUNE_S3_OFF k_wino_bgemm<2,2,2,2> / <1,2,2,2>, 72 / 73 = k_wino_bgemm_mz<...>) (polyphase Winograd of a stride-2 layer, ops 0/1).  80 + Cb: the image-facing forward kernels k_b2s_tapk / k_b2s_tapkp<Cb> (op 0, <= 5 big-side
 * channels).  Codes >= 1000: 1000 + 100 * ring + 10 * dir + tile = the LDS-DMA bf16 kernels k_conv_bf16x on bf16 tensors, 1020 + tile =
 * k_wgrad_bf16x, 1030 + tile = bf16 row GEMM + col2im, 1040 + tile = k_conv_bf16x on 8-channel pixels, 1050 = the one-channel head's data
 * gradient k_s2b_ca1 / k_s2b_ca1_s1, 1060 + Cb (1070 + Cb: from a bf16 tensor) = the one-pass transposed convolution onto <= 8 channels
 * k_s2b_tapnf<Cb> (Cb > 4: two launches, <4> + <Cb - 4>).  For profiling only. */
int pg_conv_describe(const pg_conv_geom* g, int op, size_t ws_bytes, int* tile_id, int* split, long* workgroups);

/* The kernel symbol (as rocprofv3 prints it, without the namespace / argument list) of the main GEMM kernel that op would
 * launch -- the same decision code as the entry points, so profiles can be attributed without re-deriving the dispatch --
 * its split-K factor, and the FLOPs that kernel executes on the MFMA pipe (the direct-convolution count 2*N*Hs*Ws*16*Ca*Cb
 * for the implicit-GEMM kernels, 2.25-4x fewer for the Winograd kernels, ragged tiles included).  `op` as in
 * pg_conv_describe.  For profiling only. */
int pg_conv_kernel(const pg_conv_geom* g, int op, size_t ws_bytes, char* name, size_t name_len, int* split, double* mfma_flops);

/* The two FLOP counts of that kernel: `executed` as pg_conv_kernel reports it (a Winograd kernel's ragged edge tiles are whole tiles on
 * the MFMA pipe: up to 27 % of the work on 16x16 maps), `useful` = the same algorithm's count on the exact extents (H / m instead of
 * ceil(H / m) tiles).  bench.py's roofline.frac / roofline.frac_useful; the direct-convolution count is 2*N*Hs*Ws*16*Ca*Cb for every
 * kernel.  For profiling only (replaces nothing in the reference: torch.profiler's per-op FLOP column is the closest thing). */
int pg_conv_kernel_flops(const pg_conv_geom* g, int op, size_t ws_bytes, double* executed, double* useful);

/* Arms per-launch timing: the NEXT pg_conv4x4_* call on this thread records the caller-owned hipEvent_t `ev_start`
 * immediately before and `ev_stop` immediately after its main GEMM kernel on the launch stream (the split-K reduce, the
 * bias column sums and the col2im / gather pass are outside the pair), then disarms itself.  Thread-local state; the
 * default (never armed) adds nothing to a launch.  Used by bench.py's roofline leg. */
int pg_conv_time_next(void* ev_start, void* ev_stop);

/* small[n,p,q,a] = act( sum_{kh,kw,b} big[n, s*p-1+kh, s*q-1+kw, b] * P[kh*4+kw][a][b] + bias[a] )
 * Replaces: nn.Conv2d forward (unet.py:19; disc.py:19,27,37,45) with (a,b) = (Cout,Cin), and the
 * data-gradient of nn.ConvTranspose2d (unet.py:53, aten::convolution_backward) with (a,b) = (Cin,Cout).
 * bias may be NULL; act is a PG_ACT_* code applied in the epilogue. */
int pg_conv4x4_big2small(const float* big, int ld_big, const float* P, const float* bias,
                         float* small, int ld_small, const pg_conv_geom* g, int act, int algo,
                         void* ws, size_t ws_bytes, void* stream);

/* big[n,h,w,b] = act( sum_{kh,kw,a : (h+1-kh)%s==0, (w+1-kw)%s==0}
 *                      small[n,(h+1-kh)/s,(w+1-kw)/s,a] * P[kh*4+kw][a][b] + bias[b] )
 * Replaces: nn.ConvTranspose2d forward (unet.py:53) with (a,b) = (Cin,Cout), and the data-gradient of
 * nn.Conv2d (aten::convolution_backward) with (a,b) = (Cout,Cin). */
int pg_conv4x4_small2big(const float* small, int ld_small, const float* P, const float* bias,
                         float* big, int ld_big, const pg_conv_geom* g, int act, int algo,
                         void* ws, size_t ws_bytes, void* stream);

/* dP[kh*4+kw][a][b] = sum_{n,p,q} small[n,p,q,a] * big[n, s*p-1+kh, s*q-1+kw, b]
 * Replaces: the weight-gradient half of aten::convolution_backward for both nn.Conv2d
 * (small = dL/dy, big = x) and nn.ConvTranspose2d (small = x, big = dL/dy).
 * Deterministic (fixed-order split-K slabs, no float atomics).  dbias (may be NULL) receives
 * sum_{n,p,q} small[n,p,q,a] (the Conv2d bias gradient, disc.py:19,45). */
int pg_conv4x4_wgrad(const float* small, int ld_small, const float* big, int ld_big,
                     float* dP, float* dbias, const pg_conv_geom* g, int algo,
                     void* ws, size_t ws_bytes, void* stream);

/* Optional hand-overs between calls on the SAME layer, for the Winograd paths of PG_ALGO_AUTO (all fields may be NULL / 0):
 *   part     out: InstanceNorm partial sums of the output (the conv epilogue emits sum / sum of squares -- unet.py:19-20,53-55,
 *            a Conv2d / ConvTranspose2d feeding an InstanceNorm2d): part[((n * chunks + chunk) * C + c) * 2 + {0, 1}], fp64,
 *            fixed summation order, chunks = pg_conv_stats_chunks(...)
 *   v_keep   big2small: write the polyphase-transformed input here (pg_conv_v_bytes) instead of into the workspace ...
 *   v_pre    wgrad: ... so that the weight gradient of the same layer (same `big` tensor) reads it instead of redoing the transform
 *   u_cache  big2small / small2big: the transformed weights of this (layer, direction) live here (pg_conv_u_bytes) ...
 *   u_valid  ... and, when non-zero, already hold the transform of the CURRENT weights: the weight transform is skipped
 *            (the discriminator's weights serve two forward and two data-gradient passes per step, trainer.py:66,98-99).
 * The size queries mirror the dispatch for 16-byte-aligned tensors and return 0 when the call would not take a path that has
 * such an operand; passing a hand-over to a call that does not take that path returns PG_EINVAL (nothing is launched). */
typedef struct pg_conv_extras {
    double* part;
    float* v_keep;
    const float* v_pre;
    float* u_cache;
    int u_valid;
    /* small2big as a data gradient: out = conv(...) * f'(t), the activation backward of the layer BELOW folded into the epilogue
     * (trainer.py:89,106: autograd's TanhBackward / LeakyReluBackward between two ConvolutionBackward nodes).  mul_t = that layer's
     * activation OUTPUT [pixels of big][Cb], pixel stride mul_ld, storage type of `big`; mul_act = its PG_ACT_* (the derivative is
     * expressed through the output).  Honoured where pg_conv_mul_ok() != 0; elsewhere PG_EINVAL. */
    const void* mul_t;
    int mul_ld;
    int mul_act;
} pg_conv_extras;
int pg_conv_mul_ok(const pg_conv_geom* g, int algo, size_t ws_bytes);   /* small2big: the kernel of this call can apply mul_t */
int pg_conv_stats_chunks(const pg_conv_geom* g, int op, int algo, size_t ws_bytes);   /* op 0 big2small, 1 small2big */
size_t pg_conv_u_bytes(const pg_conv_geom* g, int op, int algo, size_t ws_bytes);      /* op 0 big2small, 1 small2big */
size_t pg_conv_v_bytes(const pg_conv_geom* g, int algo, size_t ws_bytes);              /* big2small forward -> wgrad */
int pg_conv4x4_big2small_x(const float* big, int ld_big, const float* P, const float* bias, float* small, int ld_small,
                           const pg_conv_geom* g, int act, int algo, void* ws, size_t ws_bytes, void* stream,
                           const pg_conv_extras* x);
int pg_conv4x4_small2big_x(const float* small, int ld_small, const float* P, const float* bias, float* big, int ld_big,
                           const pg_conv_geom* g, int act, int algo, void* ws, size_t ws_bytes, void* stream,
                           const pg_conv_extras* x);
int pg_conv4x4_wgrad_x(const float* small, int ld_small, const float* big, int ld_big, float* dP, float* dbias,
                       const pg_conv_geom* g, int algo, void* ws, size_t ws_bytes, void* stream, const pg_conv_extras* x);
/* The weight preparations (Winograd weight transforms; packed bf16 weights) of SEVERAL layers in one launch per kernel family: item i
 * fills items[i].u with exactly what the call `op` (0 big2small, 1 small2big) on that geometry / algo / workspace size would write into
 * pg_conv_extras.u_cache with u_valid == 0 (pg_conv_u_bytes(...) bytes; PG_EINVAL, nothing launched, if that is 0 for any item).
 * The caller then passes u_valid = 1.  A network's weights change once per step (trainer.py:90,107 optimizer.step()): one call per
 * network and step instead of one small transform kernel per layer, direction and step. */
typedef struct pg_conv_prep_item {
    pg_conv_geom g;
    int op;
    int algo;
    size_t ws_bytes;
    const float* P;
    void* u;
} pg_conv_prep_item;
int pg_conv_prep_batch(int n, const pg_conv_prep_item* items, void* stream);
/* pg_instnorm_act_fwd with the statistics pass replaced by the producer's partial sums: merge (fixed order) + normalise. */
int pg_instnorm_act_fwd_parts(const float* y, int ld_y, float* out, int ld_out, float* stats, const double* part, int chunks,
                              int N, int HW, int C, int act, float eps, float drop_p, uint64_t seed, void* stream);

/* bf16 activation storage.  The *_t forms of the InstanceNorm / activation entry points take tensors that are
 * fp32 OR bf16 (NHWC, `ld` in elements of the tensor's own type; 4-element accesses need 16- resp. 8-byte alignment, else the
 * scalar kernels run): bit i of `dt` set = the i-th tensor argument, in signature order, is bf16 --
 *     pg_instnorm_act_fwd_t / _fwd_parts_t: y, out;   pg_instnorm_act_bwd_t: g1, g2, y, dy;   pg_act_fwd_t: y, out;
 *     pg_act_bwd_t: g1, g2, a, dy.
 * Statistics (fp32), partial sums (fp64) and every arithmetic step are those of the fp32 entry points: a bf16 tensor is widened
 * on load and rounded to nearest-even on store.  dt = 0 is exactly the fp32 entry point.  pg_act_fwd_t with PG_ACT_NONE converts
 * between the two storage types. */
int pg_instnorm_act_fwd_t(const void* y, int ld_y, void* out, int ld_out, float* stats, int N, int HW, int C, int act, float eps,
                          float drop_p, uint64_t seed, void* ws, size_t ws_bytes, void* stream, int dt);
int pg_instnorm_act_fwd_parts_t(const void* y, int ld_y, void* out, int ld_out, float* stats, const double* part, int chunks,
                                int N, int HW, int C, int act, float eps, float drop_p, uint64_t seed, void* stream, int dt);
int pg_instnorm_act_bwd_t(const void* g1, int ld_g1, const void* g2, int ld_g2, const void* y, int ld_y, const float* stats,
                          void* dy, int ld_dy, int N, int HW, int C, int act, float drop_p, uint64_t seed, void* ws,
                          size_t ws_bytes, void* stream, int dt);
int pg_act_fwd_t(const void* y, int ld_y, void* out, int ld_out, long npix, int C, int act, float drop_p, uint64_t seed,
                 void* stream, int dt);
int pg_act_bwd_t(const void* g1, int ld_g1, const void* g2, int ld_g2, const void* a, int ld_a, void* dy, int ld_dy, long npix,
                 int C, int act, float drop_p, uint64_t seed, void* stream, int dt);

/* Backward of a layer whose BIG side carries the incoming gradient -- nn.ConvTranspose2d (unet.py:53) with small = the layer's
 * input x and big = dL/dy -- in ONE call (aten::convolution_backward, trainer.py:89, is one op producing both):
 *     dP     = pg_conv4x4_wgrad(small = x, big = dy)                  (no bias: UpSampleBlock has none)
 *     dsmall = pg_conv4x4_big2small(big = dy, P)                      (dL/dx, no bias, no activation)
 * Where both halves take the polyphase Winograd path, the transformed gradient V(dy) is computed ONCE and read by both
 * GEMMs (the dy tile shared between the data- and the weight-gradient, north_star's "wgrad/dgrad fused"); otherwise the
 * call is exactly the two calls above.  Bit-identical to them either way.  ws: pg_conv_workspace_bytes(g, 3). */
int pg_conv4x4_bwd_big(const float* small, int ld_small, const float* big, int ld_big, const float* P, float* dP,
                       float* dsmall, int ld_dsmall, const pg_conv_geom* g, int algo, void* ws, size_t ws_bytes, void* stream);
/* ... with the data gradient's transformed weights in a caller-owned cache (x->u_cache / u_valid as for pg_conv4x4_big2small_x,
 * size pg_conv_u_bytes(g, 0, ...); every other field of x must be NULL / 0). */
int pg_conv4x4_bwd_big_x(const float* small, int ld_small, const float* big, int ld_big, const float* P, float* dP,
                         float* dsmall, int ld_dsmall, const pg_conv_geom* g, int algo, void* ws, size_t ws_bytes, void* stream,
                         const pg_conv_extras* x);

/* As pg_conv_time_next, for pg_conv4x4_bwd_big: the first pair goes around its weight-gradient GEMM, the second around its
 * data-gradient GEMM. */
int pg_conv_time_next2(void* ev_start, void* ev_stop, void* ev_start2, void* ev_stop2);

/* InstanceNorm2d(eps, biased var, no affine; unet.py:20,55, disc.py:32,42) + activation + Dropout(p)
 * (unet.py:28,65) over y[N, HW, C]:   out = dropout(act((y - mean_nc) * rstd_nc)).
 * stats[(n*C + c)*2 + {0,1}] receives (mean, rstd).  drop_p == 0 disables dropout; otherwise element e =
 * (n*HW + pix)*C + c is kept iff pg_dropout_keep(seed, e) and scaled by 1/(1-p). */
int pg_instnorm_act_fwd(const float* y, int ld_y, float* out, int ld_out, float* stats,
                        int N, int HW, int C, int act, float eps,
                        float drop_p, uint64_t seed, void* ws, size_t ws_bytes, void* stream);

/* Workspace (fp64 partial sums) that lets the two InstanceNorm entry points use their chunked, fully parallel path on
 * large planes; with ws == NULL or too small they fall back to one workgroup per (sample, channel group). */
size_t pg_instnorm_workspace_bytes(int N, int HW, int C);

/* Backward of the block above: given g = dL/dout (= g1 + g2, g2 may be NULL: the skip connection's
 * second consumer), the saved y and stats, writes dy = dL/dy. */
int pg_instnorm_act_bwd(const float* g1, int ld_g1, const float* g2, int ld_g2,
                        const float* y, int ld_y, const float* stats, float* dy, int ld_dy,
                        int N, int HW, int C, int act, float drop_p, uint64_t seed,
                        void* ws, size_t ws_bytes, void* stream);

/* Backward of a plain activation (+dropout) from its OUTPUT a:  dy = (g1+g2) * keep/(1-p) * act'(a).
 * act' from the output: leaky a>0?1:0.2, relu a>0, tanh 1-a^2, sigmoid a(1-a), none 1.  `a` may be NULL for
 * PG_ACT_NONE. */
int pg_act_bwd(const float* g1, int ld_g1, const float* g2, int ld_g2, const float* a, int ld_a,
               float* dy, int ld_dy, long npix, int C, int act, float drop_p, uint64_t seed, void* stream);

/* Elementwise activation (+dropout) forward, for blocks without a norm whose conv could not fuse it. */
int pg_act_fwd(const float* y, int ld_y, float* out, int ld_out, long npix, int C, int act,
               float drop_p, uint64_t seed, void* stream);

/* Softmax over the channel dim (nn.Softmax(dim=1), unet.py:48-49); C need not be a multiple of 4. */
int pg_softmax_fwd(const float* y, int ld_y, float* out, int ld_out, long npix, int C, void* stream);
int pg_softmax_bwd(const float* g1, int ld_g1, const float* g2, int ld_g2, const float* out, int ld_out,
                   float* dy, int ld_dy, long npix, int C, void* stream);

/* Writes the dropout keep-mask (1.0 / 0.0) the kernels above use, for tests. */
int pg_dropout_mask(float* mask, long nelem, float drop_p, uint64_t seed, void* stream);

/* ---- losses (losses.py:18-39, trainer.py:71-85,101-103) ------------------------------------------
 * Stage 1 (per rank): pg_loss_reduce accumulates, for each (sample n, channel c), the five sums
 *   S[n][c][0..4] = { sum y*p, sum y, sum p, sum bce_elem(p, y), sum |p - y| }       (double)
 * over HW, where y is the target tensor or, when y == NULL, the constant `tconst`;
 * bce_elem = -( y*max(log p,-100) + (1-y)*max(log(1-p),-100) ).  C need not be a multiple of 4. */
int pg_loss_reduce(const float* p, int ld_p, const float* y, int ld_y, float tconst,
                   int N, int HW, int C, double* S, void* stream);
/* Number of doubles S must hold for pg_loss_reduce(N, HW, C): N*C*5 results, followed by scratch for the per-split
 * partial sums large maps are reduced through. */
long pg_loss_reduce_doubles(int N, int HW, int C);
/* Largest N * C for which pg_loss_value_grad (value + gradient of a loss term in one launch, its per-(sample, channel) table in LDS)
 * applies; beyond it the caller uses pg_loss_finalize + pg_loss_grad (same results).  Replaces nothing in the reference. */
int pg_loss_fused_max_nc(void);

/* Stage 2: gradient of a scalar loss wrt p from per-(n,c) coefficients:
 *   mode 0 (affine in y; focal-Tversky):   g = coef[n][c][0] * y + coef[n][c][1]
 *   mode 1 (BCE):                           g = coef[n][c][0] * (p - y) / max(p*(1-p), 1e-12)
 *   mode 2 (MAE):                           g = coef[n][c][0] * sign(p - y)
 * coef is float [N][C][2] on the device. */
int pg_loss_grad(const float* p, int ld_p, const float* y, int ld_y, float tconst,
                 const float* coef, float* g, int ld_g, int N, int HW, int C, int mode, void* stream);

/* Stage 1b: local2[0] = sum_n (1 - T_n) with T_n = (tp+1)/(tp + beta*fn + (1-beta)*fp + 1) (losses.py:20-26,
 * tp/fn/fp summed over all channels of sample n); local2[1] = sum_{n,c} S[n][c][1] (= torch.sum(target),
 * trainer.py:77).  Under data parallelism the caller all-reduces local2 (sum) before stage 2. */
int pg_loss_prepare(const double* S, int N, int C, float beta, double* local2, void* stream);

/* Stage 2: from S and the (global) sums gsum2, write the loss value and the coefficients for pg_loss_grad.
 * Bglobal = global batch size (N on one GPU).  Gradients are seeded for a SUM all-reduce across ranks.
 *   PG_LOSS_TVERSKY : loss = alpha * (gsum2[0]/Bglobal)^gamma                          (global value)
 *   PG_LOSS_WBCE    : loss = alpha * sum w[n][c]*S[n][c][3] / (Bglobal*C*HW), w = C>1 ? 1 - S[n][c][1]/gsum2[1] : 1
 *   PG_LOSS_MAE     : loss = alpha * sum S[n][c][4] / (Bglobal*C*HW)
 *   PG_LOSS_BCE     : loss = alpha * sum S[n][c][3] / (Bglobal*C*HW)     (nn.BCELoss, mean; alpha = 1 or 0.5)
 * For the last three the value is this rank's partial (sum over ranks = global loss). */
#define PG_LOSS_TVERSKY 0
#define PG_LOSS_WBCE 1
#define PG_LOSS_MAE 2
#define PG_LOSS_BCE 3
int pg_loss_finalize(const double* S, const double* gsum2, int mode, int N, int C, int HW, int Bglobal,
                     float alpha, float beta, float gamma, float* coef, float* loss_out, void* stream);
/* gsum2 == NULL (one process: nothing to all-reduce): pg_loss_finalize computes the two terms of pg_loss_prepare itself.
 *
 * The same loss evaluation in TWO launches instead of five (losses.py:18-39 + trainer.py:71-85,101-103 are a handful of
 * elementwise / reduction ATen ops per term; here: one reduction pass, one value + gradient pass):
 *   pg_loss_reduce_parts   stage 1 without its combine launch: S receives the partial slabs [nsplit][N*C][5]; returns nsplit >= 1
 *                          (the buffer size is pg_loss_reduce_doubles(N, HW, C) as before) or a negative PG_E* code
 *   pg_loss_value_grad     adds the slabs in slab order, forms gsum2 (or takes the all-reduced one), writes the loss value to
 *                          loss_out and -- g != NULL -- the gradient wrt p into g, with exactly the arithmetic of
 *                          pg_loss_reduce's combine + pg_loss_prepare + pg_loss_finalize + pg_loss_grad (bit-identical results).
 *                          S_out (may be NULL) receives the summed S[N*C][5].  Needs N*C <= 256 (PG_EINVAL beyond: use the
 *                          staged entry points). */
int pg_loss_reduce_parts(const float* p, int ld_p, const float* y, int ld_y, float tconst, int N, int HW, int C, double* S,
                         void* stream);
int pg_loss_value_grad(const double* Spart, int nsplit, double* S_out, const double* gsum2, int mode, int N, int C, int HW,
                       int Bglobal, float alpha, float beta, float gamma, const float* p, int ld_p, const float* y, int ld_y,
                       float tconst, float* g, int ld_g, float* loss_out, void* stream);

/* ---- optimizer (torch.optim.Adam defaults, trainer.py:169-172) -----------------------------------
 * m += (g-m)*(1-b1); v = v*b2 + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt_bc2 + eps)
 * over n contiguous floats; bc1 = 1-b1^t and sqrt_bc2 = sqrt(1-b2^t) are computed by the caller. */
int pg_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                 float eps, float bc1, float sqrt_bc2, void* stream);
/* The same update with the two step-dependent scalars read from DEVICE memory: scalars[0] = lr / bc1, scalars[1] = sqrt(bc2) (floats the
 * caller wrote there, stream-ordered before this launch).  The launch then carries nothing that changes from step to step, so a
 * captured hipGraph of the whole G+D step (Trainer.graph) can be replayed with a new Adam step count and learning rate
 * (optimizer.step() of trainer.py:90,107 inside the replayed step). */
int pg_adam_step_dev(float* p, const float* g, float* m, float* v, long n, float beta1, float beta2, float eps, const float* scalars,
                     void* stream);

/* ---- layout (the reference is NCHW end to end; trainer.py:55-66) ---------------------------------- */
int pg_nchw_to_nhwc(const float* src, float* dst, int ld_dst, int N, int C, int H, int W, void* stream);
int pg_nhwc_to_nchw(const float* src, int ld_src, float* dst, int N, int C, int H, int W, void* stream);
/* The discriminator's two concatenated inputs in one pass (trainer.py:65-66,96-99: torch.cat((x, y), 1) and torch.cat((x, gen_img), 1)):
 * real[n][hw][0 .. ld) = x[n] | y[n] | 0-pad, fake[n][hw][0 .. ld) = x[n] | 0 (the generator's output is written into channels Cx ..
 * Cx + Cy - 1 of `fake` later); x, y contiguous NCHW fp32, real / fake NHWC with pixel stride ld (Cx + Cy <= ld <= 8). */
int pg_din_fill(const float* x, const float* y, float* real, float* fake, int ld, int N, int Cx, int Cy, int H, int W, void* stream);
/* dst[pix*ld_dst + c] = src[pix*ld_src + c] for c < C  (channel-slice copy; C need not be a multiple of 4) */
int pg_copy_channels(const float* src, int ld_src, float* dst, int ld_dst, long npix, int C, void* stream);
int pg_fill(float* dst, long n, float value, void* stream);
/* fp32 NHWC channel slice (C <= 8, pixel stride ld_src) -> bf16 tensor in 8-channel pixels (ld 8, channels C..7 zero, 16-byte-aligned
 * dst): the form in which the PG_ALGO_BF16 kernels take the image-facing tensors (x, x | mask, dL/d(generator output)) -- one 16-byte
 * LDS-DMA piece per pixel.  pg_conv4x4_big2small / _wgrad with PG_IO_BIG_BF16, Cb <= 8 and ld_big == 8 read that layout. */
int pg_pad8_bf16(const float* src, int ld_src, void* dst, long npix, int C, void* stream);

/* ---- input pipeline on the device (io.py:42-56: the dataset's `/ 255.` and one-hot mask) ----------------
 * dst[pix*ld_dst + c] = (float)src[pix*C + c] / div : decoded image bytes [npix][C] (HWC) into an NHWC channel slice */
int pg_u8_to_f32(const unsigned char* src, float* dst, int ld_dst, long npix, int C, float div, void* stream);
/* dst[pix*ld_dst + i] = ((unsigned char)(src[pix] + add) == labels[i]) ? 1 : 0 for i < nlabels <= PG_MAX_LABELS; `labels` is
 * a HOST array (copied into the launch); add = 1 reproduces the reference's uint8 `read_image(mask) + 1` incl. 255 -> 0 */
#define PG_MAX_LABELS 32
int pg_labels_to_onehot(const unsigned char* src, float* dst, int ld_dst, long npix, const int* labels, int nlabels, int add,
                        void* stream);

/* ---- tiled inference (infer.py:14-68: n_crop / build_mask around the generator forward) -----------
 * Tile k of an axis of `extent` pixels starts at k*eff - max(k*eff + size - extent, 0), eff = int(overlap*size),
 * k < pg_tiles_count(extent, size, eff) = ceil(extent / eff) (0 if extent < size: the reference cannot tile such an image).
 * Tiles are numbered row-major over (y tile, x tile) -- the reference's j*ncropsy + i on the square images it supports. */
int pg_tiles_count(int extent, int size, int eff);
/* n_crop (infer.py:14-35): image [C,H,W] (CHW, the dataset item layout) -> tiles [ny*nx, size, size, ld] NHWC, the batch
 * the generator kernels read */
int pg_tiles_gather(const float* image, int C, int H, int W, int size, int eff, float* tiles, int ld, void* stream);
/* build_mask (infer.py:38-68): predicted tiles [ny*nx, size, size, ld] NHWC -> overlap average in double, in tile order
 * (bit-identical to the reference's `mask += tile.double(); mask / count`), `>= threshold` -> {0,1} when threshold > 0.
 * mask: [C,H,W] doubles or NULL; argmax: [H,W] int64 class index (first maximum, numpy.argmax) or NULL. */
int pg_tiles_blend(const float* tiles, int ld, int C, int size, int eff, int H, int W, double threshold, double* mask,
                   long long* argmax, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PATCHGAN_HIP_H */
