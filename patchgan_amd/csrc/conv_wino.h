// conv_wino.h -- internal interface of the Winograd F(2x2, 4x4) path (conv_wino.hip), used by the C-ABI entry points in
// conv_gemm.hip for stride-1 layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "pg_common.h"


// geometry + alignment gate; (in, out) = (big, small) forward, (small, big) data gradient
// mo_forced: 0 = tile-edge heuristic, 2 / 3 = F(2x2,4x4) / F(3x3,4x4) pinned (PG_TUNE_WINO1_F2 / _F3)
bool pg_wino_eligible(int N, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int ld_in, const void* in, int mo_forced);
bool pg_wino_geom_ok(int N, int Hout, int Wout, int Cin, int Cout, int mo_forced);
int pg_wino_mo(int N, int Hout, int Wout, int Cin, int Cout, int mo_forced);     // 2: F(2x2,4x4), 3: F(3x3,4x4) where its 64x64 grid fills the chip
// true: k_wino_gemm<1,1,2,2> (64-tile rows), false: <2,1,2,2> (128-tile rows)
bool pg_wino_small_tile(int N, int Hout, int Wout, int Cin, int Cout, int mo_forced);
// U (25*Cout*Cin floats) followed by V (25*tiles*Cin floats), each 256-byte aligned
size_t pg_wino_ws_bytes(int N, int Hout, int Wout, int Cin, int Cout, int mo_forced);
// weight transform + input transform into ws
int pg_wino_prepare(const float* in, int ld_in, const float* P, int flip, int N, int Hin, int Win, int Cin, int Hout,
                    int Wout, int Cout, int pad, void* ws, hipStream_t st, int mo_forced, float* Uext, int u_valid, float* Vext = nullptr);
// Uext (optional, all paths with a weight transform): caller-owned cache of the transformed weights; u_valid != 0: it already
// holds the transform of the current weights and the transform kernel is skipped
size_t pg_wino_u_bytes(int N, int Hout, int Wout, int Cin, int Cout, int mo_forced);
size_t pg_wino2_u_bytes(int Ca, int Cb);     // both polyphase directions
// Weight transforms of several layers in one launch (pg_conv_prep_batch): kind 0 = stride-1 layer (Ca = Cout, Cb = Cin of the correlation,
// mo = 2 / 3, flip as pg_wino_prepare), 1 = polyphase big -> small (k_wino2_u), 2 = polyphase small -> big (k_wino2c_u); U as the
// Uext of the corresponding call.  At most PG_WINO_PREP_MAX items per call.
#define PG_WINO_PREP_MAX 24
struct pg_wino_prep {
    const float* P;
    float* U;
    int Ca, Cb, kind, mo, flip;
};
int pg_wino_prep_batch(int n, const pg_wino_prep* items, hipStream_t st);
// K slices of the F(3x3,4x4) GEMM (1: none).  > 1: pg_wino_gemm writes nsl partial outputs, dense [pixel][Cout], to pg_wino_gemm_slabs(ws, ...)
// (inside pg_wino_ws_bytes) WITHOUT bias / activation / multiplier, and the caller reduces them in slice order
int pg_wino_gemm_slices(int N, int Hout, int Wout, int Cin, int Cout, int mo_forced);
// true: pg_wino_gemm runs the row-split pair k_wino_gemm_row + k_wino_t_out (complete output, no slab reduce to follow)
bool pg_wino_gemm_rows(int N, int Hout, int Wout, int Cin, int Cout, int mo_forced, int dma_mode, const float* out, int ld_out,
                       const float* bias, pg_epi_mul mul);
bool pg_wino_row_on();
float* pg_wino_gemm_slabs(void* ws, int N, int Hout, int Wout, int Cin, int Cout, int mo_forced);
int pg_wino_dma_mode();   // process default (PATCHGAN_WINO_DMA): 0 register-staged k_wino_gemm only, 1: k_wino_gemm_dma<3,4,2> for F(3x3,4x4), 2: also <2,3,3> for 64-tile F(2x2,4x4)
// the batched GEMM with fused output transform, bias and activation
int pg_wino_gemm(const float* bias, float* out, int ld_out, int N, int Cin, int Hout, int Wout, int Cout, int act,
                 void* ws, hipStream_t st, int mo_forced, int dma_mode, const float* Uext, pg_epi_mul mul = pg_epi_mul{nullptr, 0, 0},
                 const float* Vext = nullptr, int s3 = 0);
// s3: the row-fused F(3x3,4x4) GEMM in split-bf16 form (k_wino_gemm_row_s3 instead of k_wino_gemm_row; Cin % 32 == 0)
// Vext (both): the transformed input lives in a caller-owned buffer (pg_wino_wgrad_v_bytes) instead of the workspace -- the forward
// call keeps it for the layer's weight gradient (Vpre of pg_wino_wgrad), F(3x3,4x4) forward + F(4x4,3x3) weight gradient only

// weight gradient of the same layers, F(4x4, 2x2) or F(4x4, 3x3) (X = 25 / 36 points): V (X*tiles*Cb) | DY (X*tiles*Ca) | S (slices*X*Ca*Cb) in ws
bool pg_wino_wgrad_geom_ok(int N, int Hs, int Ws, int Ca, int Cb);
int pg_wino_wgrad_r(int N, int Hs, int Ws);      // dy-tile edge: 2 = F(4x4,2x2), 3 = F(4x4,3x3)
double pg_wino_wgrad_flops(int N, int Hs, int Ws, int Ca, int Cb);      // FLOPs its GEMM executes
int pg_wino_wgrad_slices(int N, int Hs, int Ws, int Ca, int Cb, int s3);
bool pg_wino_wgrad_tile64(int Ca, int Cb, int s3);      // k_wino_wgrad_gemm<1,1,2,2> instead of <2,2,2,2>; s3: for the split-bf16 form
size_t pg_wino_wgrad_ws_bytes(int N, int Hs, int Ws, int Ca, int Cb);
// ev0 / ev1 (optional) are recorded around the GEMM kernel
int pg_wino_wgrad(const float* small, int ld_small, const float* big, int ld_big, float* dP, int N, int Hb, int Wb, int Hs,
                  int Ws, int Ca, int Cb, void* ws, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, const float* Vpre = nullptr, int s3 = 0);
// s3 (both weight gradients): != 0 runs the GEMM in split-bf16 form (k_wino_wgrad_gemm_s3; same tiles, slices and workspace)
size_t pg_wino_wgrad_v_bytes(int N, int Hs, int Ws, int Ca, int Cb, int fwd_mo_forced);     // 0: the two calls do not share a transform

// stride-2 layers, polyphase F(MO x MO, 2x2), MO = pg_wino2_mo() (3, or 4 with PATCHGAN_WINO2_TILE=4), X = (MO+1)^2:
// big -> small: U (X*Ca*4Cb) | V (X*tiles*4Cb) | M (X*tiles*Ca) in ws
int pg_wino2_mo();
long pg_wino2_tiles_b2s(int N, int Hs, int Ws);
long pg_wino2_tiles_s2b(int N, int Hb, int Wb);
int pg_wino2_b2s_zb(int N, int Hs, int Ws, int Ca);   // batches per workgroup (> 1: k_wino_bgemm_mz)
int pg_wino2_s2b_zb(int N, int Hb, int Wb, int Cb);
bool pg_wino2_geom_ok(int N, int Hs, int Ws, int Ca, int Cb);
size_t pg_wino2_ws_bytes(int N, int Hs, int Ws, int Ca, int Cb);
// Vpre (optional, F(3x3,2x2) only): the transformed input already computed by pg_wino2_v; ws then holds U | M only
int pg_wino2_b2s(const float* big, int ld_big, const float* P, const float* bias, float* small, int ld_small, int N, int Hb,
                 int Wb, int Hs, int Ws, int Ca, int Cb, int act, void* ws, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                 const float* Vpre, double* part, float* Vkeep, float* Uext, int u_valid, int s3);
// s3 (both directions): != 0 runs the batched GEMM in split-bf16 form (k_wino_bgemm_s3: fp32 operands split into three bf16 pieces while they
// are staged, six bf16 MFMAs per 16 k, fp32 accumulate), 0 on v_mfma_f32_32x32x2_f32 (k_wino_bgemm)
// part (optional, both directions): the output transform also writes per-sample partial sums / sums of squares of the output,
// part[((n * chunks + chunk) * C + c) * 2 + {0,1}] (fp64), chunks = pg_wino2_*_stats_chunks(...) (0: not available)
int pg_wino2_b2s_stats_chunks(int N, int Hs, int Ws, int Ca);
int pg_wino2_s2b_stats_chunks(int N, int Hb, int Wb, int Cb);
size_t pg_wino2_v_bytes(int N, int Hs, int Ws, int Cb);
int pg_wino2_v(const float* big, int ld_big, float* V, int N, int Hb, int Wb, int Hs, int Ws, int Cb, hipStream_t st);

// small -> big, four parity classes as column blocks: U (X*4Cb*Ca) | V (X*tiles*Ca) | M (X*tiles*4Cb) in ws
bool pg_wino2c_geom_ok(int N, int Hb, int Wb, int Ca, int Cb);
size_t pg_wino2c_ws_bytes(int N, int Hb, int Wb, int Ca, int Cb);
int pg_wino2_s2b(const float* small, int ld_small, const float* P, const float* bias, float* big, int ld_big, int N, int Hb,
                 int Wb, int Hs, int Ws, int Ca, int Cb, int act, void* ws, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                 double* part, float* Uext, int u_valid, pg_epi_mul mul, int s3);

// weight gradient of the stride-2 layers, polyphase F(2x2, 3x3): V (16*tiles*4Cb) | DY (16*tiles*Ca) | S (slices*16*Ca*4Cb)
bool pg_wino2_wgrad_geom_ok(int N, int Hs, int Ws, int Ca, int Cb);
int pg_wino2_wgrad_slices(int N, int Hs, int Ws, int Ca, int Cb, int s3);
bool pg_wino2_wgrad_tile64(int Ca, int Cb, int s3);     // k_wino_wgrad_gemm<1,1,2,2> (64x64 output tiles) instead of <2,2,2,2>
size_t pg_wino2_wgrad_ws_bytes(int N, int Hs, int Ws, int Ca, int Cb);
// Vpre (optional): as for pg_wino2_b2s; ws then holds DY | S only
int pg_wino2_wgrad(const float* small, int ld_small, const float* big, int ld_big, float* dP, int N, int Hb, int Wb, int Hs,
                   int Ws, int Ca, int Cb, void* ws, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, const float* Vpre, int s3 = 0);
