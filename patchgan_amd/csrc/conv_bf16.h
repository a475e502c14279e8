// conv_bf16.h -- internal interface of the bf16-tensor convolution kernels (conv_bf16.hip): implicit GEMMs on
// v_mfma_f32_32x32x16_bf16 whose operand tiles are brought in by LDS-DMA straight from bf16 tensors in HBM.  Used by the
// C-ABI entry points in conv_gemm.hip for PG_ALGO_BF16 calls whose activations are stored as bf16 (PG_IO_*_BF16).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "pg_common.h"

// direction: 0 = big -> small (Conv2d forward / ConvTranspose2d data gradient), 1 = small -> big,
// 2 = big -> small from an 8-channel-per-pixel `big` (Cb <= 8 real channels, ld 8: the image-facing layers),
// 3 = row GEMM  out[m][n] = sum_a small[m][a] * W[n][a]  over the N*Hs*Ws pixels of `small`, n < Cb (callers pass Cb = 16 * real Cb:
//     the taps-folded-into-N first half of a few-channel ConvTranspose2d; W = pack dir 1 of the real layer)
struct pg_bf16x_plan {
    int tile;              // 0: 256x128 rows x channels per workgroup (four waves), 1: 128x128, 2: 256x64, 3: 256x128 on eight waves
    int bm, bn;
    int tiles_m, tiles_n, ncls;
    int nchunks;           // 64-wide K chunks: taps * Cin / 64
    int split, cps;        // split-K slices and chunks per slice (after pg_bf16x_clamp)
    int ring;              // 1: the three-stage ring kernel (32-wide chunks), 0: one buffer of 64-wide chunks
    int win;               // 1: the window-staged stride-2 kernel k_conv_bf16r (dir 0 / 1, tile 1 or 2, maps in whole R x 16 rectangles)
    long out_elems;        // elements of one fp32 slab
};

// geometry gate (channel multiples, 32-bit byte offsets); alignment of the actual pointers is checked by the caller
bool pg_bf16x_geom_ok(int dir, int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride);
// ring: 1 / 0 pins the staging variant, -1 = per-layer default
pg_bf16x_plan pg_bf16x_plan_of(int dir, int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride, int ring);
void pg_bf16x_clamp(pg_bf16x_plan* p, size_t slab_bytes_available);
const char* pg_bf16x_kernel_name(int dir, int tile, int ring);

// bf16 copy of the packed weights P[tap][a][b] (fp32): dir 0 keeps the layout, dir 1 / 3 transpose each tap to [tap][b][a]
// (the GEMM's K index must be the contiguous one of both operands), dir 2 writes [a][tap][8] zero padded
size_t pg_bf16x_w_bytes(int Ca, int Cb);
int pg_bf16x_pack(const float* P, void* W, int Ca, int Cb, int dir, hipStream_t st);
// the packs of several layers in one launch (pg_conv_prep_batch); at most PG_BF16X_PACK_MAX items per call
#define PG_BF16X_PACK_MAX 24
struct pg_bf16x_pack_item {
    const float* P;
    void* W;
    int Ca, Cb, dir;
};
int pg_bf16x_pack_batch(int n, const pg_bf16x_pack_item* items, hipStream_t st);

// out: the destination tensor (bf16 if out_bf, else fp32; bias and activation applied) when slab_stride == 0, else fp32 slabs
// [split][pixels][Cout] for the caller's reduce pass
int pg_bf16x_conv(int dir, const void* in, int ld_in, long in_bytes, const void* W, void* out, int ld_out, long slab_stride,
                  int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride, const pg_bf16x_plan* p, const float* bias,
                  int act, int out_bf, hipStream_t st, pg_epi_mul mul = pg_epi_mul{nullptr, 0, 0}, double* part = nullptr, int chunks = 0, int bt = 0);
// bt (dir 1): W is the dir-0 pack ([tap][a][b]): the kernel stages the weight tile [a][b] and reads it transposed
// part (dir 0 / 1, unsplit K, bf16 output): the epilogue also writes per-sample partial sums / sums of squares of the stored values,
// part[((n * chunks + chunk) * Cout + c) * 2 + {0, 1}] (fp64), chunks = pg_bf16x_stats_chunks(...) (0: not available)
int pg_bf16x_stats_chunks(int dir, const pg_bf16x_plan* p, int N, int Hb, int Wb, int Hs, int Ws);

// weight gradient on bf16 tensors (both operands), k_wgrad_bf16x: out = dP (slab_stride == 0, split 1) or fp32 slabs
// [split][16 * Ca * Cb].  The plan's tile / tiles_m / tiles_n are over (Ca, Cb), nchunks in 64-pixel chunks.
bool pg_bf16x_wgrad_geom_ok(int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride);
pg_bf16x_plan pg_bf16x_wgrad_plan(int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride);
const char* pg_bf16x_wgrad_kernel_name(int tile);
int pg_bf16x_wgrad(const void* small, int ld_small, long small_bytes, const void* big, int ld_big, long big_bytes, float* out,
                   long slab_stride, int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride, const pg_bf16x_plan* p,
                   hipStream_t st);
