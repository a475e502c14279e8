// s3_probe.hip -- where does the split-bf16 batched row GEMM (k_wino_bgemm_s3 in conv_wino.hip) lose time, and what would pre-split
// operands buy?  Stand-alone probe (not part of the library): the library kernel's loop (128 x 128 tile, 16-wide K chunks, two LDS
// buffers, one barrier per chunk, two register sets of loads in flight) with parts switched off or replaced.
//   mode 0: the library loop (fp32 operands from global memory -> registers -> split into three bf16 pieces -> LDS -> 6 MFMAs per fp32 product)
//   mode 1: the split replaced by three register moves (no VALU arithmetic; wrong numbers): what the split's arithmetic costs
//   mode 2: no staging at all (loads consumed by one add each, no split, no LDS stores; barriers kept): what split + LDS stores cost
//   mode 3: LDS reads + MFMAs + barriers only (no global loads)
//   mode 4: MFMAs only
//   mode 5: B operand PRE-SPLIT in global memory ([row][chunk][piece][16 k] bf16, 96 bytes per row and chunk), brought in by LDS-DMA into
//           128-byte rows with swizzled 16-byte slots; A as in mode 0
//   mode 6: both operands pre-split + LDS-DMA (no VALU, no LDS stores in the loop)
//   modes 16 / 17: mode 0 with the split computed on PAIRS of values (one v_cvt_pk_bf16_f32 per pair and piece, widening by shift / mask of the
//           packed word: 4.5 instead of ~7 VALU instructions per value, the same RNE pieces bit for bit); 17 keeps the subtractions out of
//           v_pk_add_f32 (inline asm), which MI355X_MICROARCH.md prices at +13 cycles beside MFMAs
//   mode 22: mode 20 with 32-wide K chunks (a staged row piece = one whole 128-byte line)
//   mode 21: no barriers -- every wave stages its own operands into a private LDS region (mode 20's fp32 rows) and never synchronises
//   mode 20: split on read -- fp32 rows in LDS, fragments read as fp32 and split in registers by the wave that uses them
//   mode 19: a 256 x 128 tile on eight waves (B staged once for twice the MFMAs), one workgroup per CU
//   mode 18: the pair split, computed one half-iteration ahead of its LDS stores (pieces held in registers)
//   mode 7: mode 0's data path with the fragments double-buffered in REGISTERS: the MFMAs of chunk c run on fragments read during chunk
//           c - 1, while this iteration reads chunk c + 1's fragments and splits / stores chunk c + 2 (nothing an MFMA waits for was issued in
//           its own iteration); mode 8: the same with the instruction mix interleaved by sched_group_barrier
// build: hipcc -O3 --offload-arch=gfx950 tools/s3_probe.hip -o tools/s3_probe ; run: tools/s3_probe [Z] [M] [N] [K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)lds, 16, byte_off, 0, 0, 0);
}
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
template <bool ASM>
__device__ __forceinline__ float fsub(float a, float b) {
    if constexpr (ASM) {
        float r;
        asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    } else {
        return a - b;
    }
}
// the same three RNE pieces, computed on PAIRS: one v_cvt_pk per pair and piece, the bf16 -> f32 widening as a shift / a mask of the packed word
template <bool ASM>
__device__ __forceinline__ void split_pairs(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
    u32x2 H, M, Lw;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = v[2 * q], x1 = v[2 * q + 1];
        const unsigned hp = pk2(x0, x1);
        const float r0 = fsub<ASM>(x0, __builtin_bit_cast(float, hp << 16)), r1 = fsub<ASM>(x1, __builtin_bit_cast(float, hp & 0xffff0000u));
        const unsigned mp = pk2(r0, r1);
        const float s0 = fsub<ASM>(r0, __builtin_bit_cast(float, mp << 16)), s1 = fsub<ASM>(r1, __builtin_bit_cast(float, mp & 0xffff0000u));
        H[q] = hp;
        M[q] = mp;
        Lw[q] = pk2(s0, s1);
    }
    h = __builtin_bit_cast(bf16x4, H);
    m = __builtin_bit_cast(bf16x4, M);
    l = __builtin_bit_cast(bf16x4, Lw);
}
template <int MODE>
__device__ __forceinline__ void split(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
    if constexpr (MODE == 16) {
        split_pairs<false>(v, h, m, l);
    } else if constexpr (MODE == 17) {
        split_pairs<true>(v, h, m, l);
    } else if constexpr (MODE == 1) {
        const bf16x8 raw = __builtin_bit_cast(bf16x8, v);
        h = __builtin_shufflevector(raw, raw, 0, 1, 2, 3);
        m = __builtin_shufflevector(raw, raw, 4, 5, 6, 7);
        l = __builtin_shufflevector(raw, raw, 2, 3, 4, 5);
    } else {
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
}

constexpr int KC = 16, LDR = 3 * KC + 8;      // register-staged rows: 112 bytes
constexpr int DROW = 64;                      // DMA rows: 128 bytes = 64 bf16 (6 slots of data, 2 unused), slot ^= (row >> 1) & 7

// A: [Z][M][K] fp32 (modes 0-5), Ap: [Z][M][K/16][3][16] bf16 (mode 6); B likewise with N rows (Bp: modes 5, 6)
template <int MODE_, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_s3(const float* __restrict__ A, const float* __restrict__ B,
                                                                                          const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp,
                                                                                          float* __restrict__ C, int Mrows, int Ncols, int K, int a_bytes,
                                                                                          int b_bytes, int ap_bytes, int bp_bytes, int tiles_m, int tiles_n, int share) {
    // modes 9 / 10 = modes 0 / 8 with a scheduling fence behind every barrier (the compiler otherwise hoists the NEXT half-iteration's
    // split above the barrier and with it the wait for loads that were issued only half an iteration earlier)
    constexpr bool FENCE = MODE_ >= 9;
    constexpr int MODE = MODE_ == 9 ? 0 : (MODE_ == 10 ? 8 : ((MODE_ == 13 || MODE_ == 14) ? 1 : ((MODE_ == 16 || MODE_ == 17 || MODE_ == 18) ? 0 : MODE_)));
    constexpr int SPLITM = (MODE_ == 16 || MODE_ == 17) ? MODE_ : MODE;
    constexpr int LIN = MODE_ == 13 ? 1 : (MODE_ == 14 ? 2 : 0);
    constexpr int BM = 128, BN = 128, AI = 2, BI = 2, MR = 2, NR = 2, WN = 2;
    constexpr bool A_DMA = MODE == 6, B_DMA = MODE == 5 || MODE == 6;
    constexpr int A_ROW = A_DMA ? DROW : LDR, B_ROW = B_DMA ? DROW : LDR;
    constexpr int BUF = BM * A_ROW + BN * B_ROW;
    __shared__ __attribute__((aligned(128))) __bf16 smem[2 * BUF];
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rAp = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, ap_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBp = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, bp_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int m0 = (w % tiles_m) * BM, z = w / tiles_m;
    const int nch = K / KC;
    const int kq = tid & 3, r0 = tid >> 2;
    int a_off[AI], b_off[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) a_off[i] = share ? ((r0 + 64 * i) * K + kq * 4) * 4 : ((z * Mrows + min(m0 + r0 + 64 * i, Mrows - 1)) * K + kq * 4) * 4;
#pragma unroll
    for (int i = 0; i < BI; ++i) b_off[i] = ((z * Ncols + min(n0 + r0 + 64 * i, Ncols - 1)) * K + kq * 4) * 4;
    // DMA: a wave brings 32 rows of an operand per chunk = 4 instructions of 8 rows; lane -> row (lane >> 3), physical slot lane & 7
    int dma_a[4], dma_b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 32 + q * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        dma_a[q] = slot < 6 ? ((z * Mrows + min(m0 + row, Mrows - 1)) * nch) * 96 + slot * 16 : 0x7fffffff;
        dma_b[q] = slot < 6 ? ((z * Ncols + min(n0 + row, Ncols - 1)) * nch) * 96 + slot * 16 : 0x7fffffff;
    }
    f32x4 dummy = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE >= 2 && MODE <= 4) {
        // the modes whose MFMAs read an LDS image that is never written in the loop: fill it with pseudo-random bf16 values of order 1 first
        // (MFMAs on zeros clock higher and draw less: such rows would flatter the loop)
        unsigned x = 0x9e3779b9u * (tid + 1) + blockIdx.x;
        for (int i = tid; i < 2 * BUF; i += 256) {
            x = x * 1664525u + 1013904223u;
            smem[i] = (__bf16)(((int)(x >> 8) & 0xffff) / 32768.0f - 1.0f);
        }
        __syncthreads();
    }
    auto issue_loads = [&](f32x4 (&ra)[AI], f32x4 (&rb)[BI], int c, __bf16* buf) {
        const bool on = c < nch;
        if constexpr (MODE <= 2 || MODE == 5 || MODE >= 7) {
#pragma unroll
            for (int i = 0; i < AI; ++i) ra[i] = bload4(rA, on ? a_off[i] + c * KC * 4 : 0x7fffffff);
        }
        if constexpr (MODE <= 2 || MODE >= 7) {
#pragma unroll
            for (int i = 0; i < BI; ++i) rb[i] = bload4(rB, on ? b_off[i] + c * KC * 4 : 0x7fffffff);
        }
        // (DMA operands are brought straight into the buffer that chunk c will be read from)
        if constexpr (A_DMA) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dma16(rAp, (char*)buf + (wave * 32 + q * 8) * 128, (on && dma_a[q] != 0x7fffffff) ? dma_a[q] + c * 96 : 0x7fffffff);
        }
        if constexpr (B_DMA) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dma16(rBp, (char*)buf + BM * A_ROW * 2 + (wave * 32 + q * 8) * 128,
                      (on && dma_b[q] != 0x7fffffff) ? dma_b[q] + c * 96 : 0x7fffffff);
        }
    };
    auto stage = [&](const f32x4 (&ra)[AI], const f32x4 (&rb)[BI], __bf16* buf) {
        if constexpr (LIN == 1) {          // timing only: the same twelve 8-byte stores per thread at linear addresses (no bank conflicts)
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, ra[i]), b = __builtin_bit_cast(bf16x8, rb[i]);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    *reinterpret_cast<bf16x4*>(&buf[((i * 6 + p) * 256 + tid) * 4]) = __builtin_shufflevector(a, a, 0, 1, 2, 3);
                    *reinterpret_cast<bf16x4*>(&buf[((i * 6 + 3 + p) * 256 + tid) * 4]) = __builtin_shufflevector(b, b, 4, 5, 6, 7);
                }
            }
            return;
        }
        if constexpr (LIN == 2) {          // timing only: six 16-byte stores per thread at linear addresses
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, ra[i]), b = __builtin_bit_cast(bf16x8, rb[i]);
                *reinterpret_cast<bf16x8*>(&buf[((i * 3 + 0) * 256 + tid) * 8]) = a;
                *reinterpret_cast<bf16x8*>(&buf[((i * 3 + 1) * 256 + tid) * 8]) = b;
                *reinterpret_cast<bf16x8*>(&buf[((i * 3 + 2) * 256 + tid) * 8]) = __builtin_shufflevector(a, b, 0, 1, 2, 3, 12, 13, 14, 15);
            }
            return;
        }
        if constexpr (MODE <= 1 || MODE == 5 || MODE >= 7) {
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                bf16x4 h, m, l;
                split<SPLITM>(ra[i], h, m, l);
                __bf16* row = &buf[(r0 + 64 * i) * LDR + kq * 4];
                *reinterpret_cast<bf16x4*>(row) = h;
                *reinterpret_cast<bf16x4*>(row + KC) = m;
                *reinterpret_cast<bf16x4*>(row + 2 * KC) = l;
            }
        }
        if constexpr (MODE <= 1 || MODE >= 7) {
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                bf16x4 h, m, l;
                split<SPLITM>(rb[i], h, m, l);
                __bf16* row = &buf[(BM + r0 + 64 * i) * LDR + kq * 4];
                *reinterpret_cast<bf16x4*>(row) = h;
                *reinterpret_cast<bf16x4*>(row + KC) = m;
                *reinterpret_cast<bf16x4*>(row + 2 * KC) = l;
            }
        }
        if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < AI; ++i) dummy += ra[i];
#pragma unroll
            for (int i = 0; i < BI; ++i) dummy += rb[i];
        }
    };
    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 bf[NR][3], af[MR][3];
    auto read_frags = [&](const __bf16* buf) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int row = (wn * NR + j) * 32 + lrow;
                if constexpr (B_DMA)
                    bf[j][p] = *reinterpret_cast<const bf16x8*>(&buf[BM * A_ROW + row * DROW + (((2 * p + lh) ^ ((row >> 1) & 7)) << 3)]);
                else
                    bf[j][p] = *reinterpret_cast<const bf16x8*>(&buf[BM * A_ROW + row * LDR + p * KC + lh * 8]);
            }
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int row = (wm * MR + i) * 32 + lrow;
                if constexpr (A_DMA)
                    af[i][p] = *reinterpret_cast<const bf16x8*>(&buf[row * DROW + (((2 * p + lh) ^ ((row >> 1) & 7)) << 3)]);
                else
                    af[i][p] = *reinterpret_cast<const bf16x8*>(&buf[row * LDR + p * KC + lh * 8]);
            }
    };
    auto compute = [&](const __bf16* buf) {
        if constexpr (MODE != 4) read_frags(buf);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
#define S3_MM(pa, pb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0)
                S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
#undef S3_MM
            }
    };
    __bf16* const buf0 = smem;
    __bf16* const buf1 = smem + BUF;
    f32x4 ra0[AI], rb0[BI], ra1[AI], rb1[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) ra0[i] = ra1[i] = f32x4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < BI; ++i) rb0[i] = rb1[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    if constexpr (MODE == 4) read_frags(buf0);
    if constexpr (MODE_ == 18) {
        // the split decoupled from the stores: the pieces of chunk c + 1 were computed one half-iteration earlier and are stored first thing;
        // the split of chunk c + 2 (registers landed long ago) then runs under this half-iteration's MFMAs with no store waiting for it
        bf16x4 pa0[AI][3], pb0[BI][3], pa1[AI][3], pb1[BI][3];
        auto do_split = [&](const f32x4 (&ra)[AI], const f32x4 (&rb)[BI], bf16x4 (&pa)[AI][3], bf16x4 (&pb)[BI][3]) {
#pragma unroll
            for (int i = 0; i < AI; ++i) split_pairs<false>(ra[i], pa[i][0], pa[i][1], pa[i][2]);
#pragma unroll
            for (int i = 0; i < BI; ++i) split_pairs<false>(rb[i], pb[i][0], pb[i][1], pb[i][2]);
        };
        auto do_store = [&](const bf16x4 (&pa)[AI][3], const bf16x4 (&pb)[BI][3], __bf16* buf) {
#pragma unroll
            for (int i = 0; i < AI; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x4*>(&buf[(r0 + 64 * i) * LDR + p * KC + kq * 4]) = pa[i][p];
#pragma unroll
            for (int i = 0; i < BI; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x4*>(&buf[(BM + r0 + 64 * i) * LDR + p * KC + kq * 4]) = pb[i][p];
        };
        issue_loads(ra0, rb0, 0, buf0);
        issue_loads(ra1, rb1, 1, buf1);
        do_split(ra0, rb0, pa0, pb0);
        issue_loads(ra0, rb0, 2, buf0);
        do_store(pa0, pb0, buf0);
        do_split(ra1, rb1, pa1, pb1);          // chunk 1
        issue_loads(ra1, rb1, 3, buf1);
        __syncthreads();
        // top of step c: buf0 = chunk c; pa1 = pieces of chunk c + 1; ra0 = chunk c + 2, ra1 = chunk c + 3 (in flight)
        for (int c = 0; c < nch; c += 2) {
            do_store(pa1, pb1, buf1);
            compute(buf0);
            do_split(ra0, rb0, pa0, pb0);      // chunk c + 2
            issue_loads(ra0, rb0, c + 4, buf0);
            __syncthreads();
            do_store(pa0, pb0, buf0);
            compute(buf1);
            do_split(ra1, rb1, pa1, pb1);      // chunk c + 3
            issue_loads(ra1, rb1, c + 5, buf1);
            __syncthreads();
        }
    } else if constexpr (MODE == 15) {
        f32x4 ra2[AI], rb2[BI], ra3[AI], rb3[BI];
        issue_loads(ra0, rb0, 0, buf0);
        issue_loads(ra1, rb1, 1, buf1);
        issue_loads(ra2, rb2, 2, buf0);
        issue_loads(ra3, rb3, 3, buf1);
        stage(ra0, rb0, buf0);
        issue_loads(ra0, rb0, 4, buf0);
        __syncthreads();
        // top of step c: buf[c & 1] = chunk c; sets 1, 2, 3, 0 hold chunks c + 1 .. c + 4
        for (int c = 0; c < nch; c += 4) {
            stage(ra1, rb1, buf1);
            compute(buf0);
            issue_loads(ra1, rb1, c + 5, buf1);
            __syncthreads();
            stage(ra2, rb2, buf0);
            compute(buf1);
            issue_loads(ra2, rb2, c + 6, buf0);
            __syncthreads();
            stage(ra3, rb3, buf1);
            compute(buf0);
            issue_loads(ra3, rb3, c + 7, buf1);
            __syncthreads();
            stage(ra0, rb0, buf0);
            compute(buf1);
            issue_loads(ra0, rb0, c + 8, buf0);
            __syncthreads();
        }
    } else if constexpr (MODE >= 7) {
        bf16x8 fa0[MR][3], fb0[NR][3], fa1[MR][3], fb1[NR][3];
        auto rd = [&](bf16x8 (&fa)[MR][3], bf16x8 (&fb)[NR][3], const __bf16* buf) {
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[j][p] = *reinterpret_cast<const bf16x8*>(&buf[BM * LDR + ((wn * NR + j) * 32 + lrow) * LDR + p * KC + lh * 8]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[i][p] = *reinterpret_cast<const bf16x8*>(&buf[((wm * MR + i) * 32 + lrow) * LDR + p * KC + lh * 8]);
        };
        auto mm = [&](const bf16x8 (&fa)[MR][3], const bf16x8 (&fb)[NR][3]) {
#define S3_MM(pa, pb)                                                                       \
    _Pragma("unroll") for (int i = 0; i < MR; ++i) _Pragma("unroll") for (int j = 0; j < NR; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa], fb[j][pb], acc[i][j], 0, 0, 0)
            S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
#undef S3_MM
        };
        auto mix = [&]() {
            if constexpr (MODE == 8) {
#pragma unroll
                for (int t = 0; t < 24; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    if (t < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // the four global loads first
                    if (t < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // next chunk's fragments early
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // a slice of the split
                    if (t >= 12) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // the staged pieces late
                }
            }
        };
        issue_loads(ra0, rb0, 0, buf0);
        issue_loads(ra1, rb1, 1, buf1);
        stage(ra0, rb0, buf0);
        issue_loads(ra0, rb0, 2, buf0);
        __syncthreads();
        rd(fa0, fb0, buf0);
        stage(ra1, rb1, buf1);
        issue_loads(ra1, rb1, 3, buf1);
        __syncthreads();
        // top of iteration c: fa0 / fb0 = chunk c; buf1 = chunk c + 1; ra0 = chunk c + 2, ra1 = chunk c + 3 (in flight); buf0 is free
        for (int c = 0; c < nch; c += 2) {
            mm(fa0, fb0);
            rd(fa1, fb1, buf1);
            stage(ra0, rb0, buf0);
            issue_loads(ra0, rb0, c + 4, buf0);
            mix();
            __syncthreads();
            if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
            mm(fa1, fb1);
            rd(fa0, fb0, buf0);
            stage(ra1, rb1, buf1);
            issue_loads(ra1, rb1, c + 5, buf1);
            mix();
            __syncthreads();
            if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (MODE == 6) {
        // all-DMA pipeline: chunk c + 1 is brought into the other buffer while chunk c is multiplied
        issue_loads(ra0, rb0, 0, buf0);
        __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
        __syncthreads();
        for (int c = 0; c < nch; c += 2) {
            issue_loads(ra1, rb1, c + 1, buf1);
            compute(buf0);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            issue_loads(ra0, rb0, c + 2, buf0);
            compute(buf1);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
        }
    } else if constexpr (MODE == 5) {
        // A: registers, two chunks ahead, split while staging; B: LDS-DMA, issued as soon as its buffer is free (one chunk ahead)
        auto issue_a = [&](f32x4 (&ra)[AI], int c) {
#pragma unroll
            for (int i = 0; i < AI; ++i) ra[i] = bload4(rA, c < nch ? a_off[i] + c * KC * 4 : 0x7fffffff);
        };
        auto issue_b = [&](int c, __bf16* buf) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dma16(rBp, (char*)buf + BM * A_ROW * 2 + (wave * 32 + q * 8) * 128,
                      (c < nch && dma_b[q] != 0x7fffffff) ? dma_b[q] + c * 96 : 0x7fffffff);
        };
        issue_a(ra0, 0);
        issue_a(ra1, 1);
        issue_b(0, buf0);
        stage(ra0, rb0, buf0);
        issue_a(ra0, 2);
        __builtin_amdgcn_s_waitcnt(0x0f72);      // vmcnt(2): everything but the last two loads (A of chunk 2)
        __syncthreads();
        for (int c = 0; c < nch; c += 2) {
            issue_b(c + 1, buf1);
            stage(ra1, rb1, buf1);
            compute(buf0);
            issue_a(ra1, c + 3);
            __builtin_amdgcn_s_waitcnt(0x0f72);
            __syncthreads();
            issue_b(c + 2, buf0);
            stage(ra0, rb0, buf0);
            compute(buf1);
            issue_a(ra0, c + 4);
            __builtin_amdgcn_s_waitcnt(0x0f72);
            __syncthreads();
        }
    } else {
        issue_loads(ra0, rb0, 0, buf0);
        issue_loads(ra1, rb1, 1, buf1);
        stage(ra0, rb0, buf0);
        issue_loads(ra0, rb0, 2, buf0);
        __syncthreads();
        for (int c = 0; c < nch; c += 2) {
            stage(ra1, rb1, buf1);
            compute(buf0);
            issue_loads(ra1, rb1, c + 3, buf1);
            __syncthreads();
            if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
            stage(ra0, rb0, buf0);
            compute(buf1);
            issue_loads(ra0, rb0, c + 4, buf0);
            __syncthreads();
            if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
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
                if (m < Mrows && col < Ncols) o[(long)m * Ncols + col] = acc[i][j][r] + dummy[r & 3];
            }
        }
}


// mode 12: PRODUCER / CONSUMER waves.  512 threads: waves 0-3 only read fragments and issue MFMAs (each a 64 x 64 quarter of the 128 x 128 tile,
// fragments double-buffered in registers), waves 4-7 only load, split and store (DEPTH chunks of loads in flight in registers).  One s_barrier
// per chunk; two LDS buffers.  One workgroup (two waves per SIMD: one of each kind) per CU.
template <int DEPTH, int SG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_s3pc(const float* __restrict__ A, const float* __restrict__ B,
                                                                                         float* __restrict__ C, int Mrows, int Ncols, int K, int a_bytes,
                                                                                         int b_bytes, int tiles_m, int tiles_n, int share) {
    constexpr int BM = 128, BN = 128, MR = 2, NR = 2, BUF = (BM + BN) * LDR;
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int m0 = (w % tiles_m) * BM, z = w / tiles_m;
    const int nch = K / KC;
    __bf16* const buf0 = smem;
    __bf16* const buf1 = smem + BUF;
    if (wave >= 4) {
        // ---------------- producer ----------------
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
        const int t = tid - 256, kq = t & 3, r0 = t >> 2;
        int a_off[2], b_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a_off[i] = share ? ((r0 + 64 * i) * K + kq * 4) * 4 : ((z * Mrows + min(m0 + r0 + 64 * i, Mrows - 1)) * K + kq * 4) * 4;
            b_off[i] = ((z * Ncols + min(n0 + r0 + 64 * i, Ncols - 1)) * K + kq * 4) * 4;
        }
        f32x4 ra[DEPTH][2], rb[DEPTH][2];
        auto issue = [&](int slot, int c) {
            const bool on = c < nch;
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[slot][i] = bload4(rA, on ? a_off[i] + c * KC * 4 : 0x7fffffff);
#pragma unroll
            for (int i = 0; i < 2; ++i) rb[slot][i] = bload4(rB, on ? b_off[i] + c * KC * 4 : 0x7fffffff);
        };
        auto stage = [&](int slot, __bf16* buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x4 h, m, l;
                split<0>(ra[slot][i], h, m, l);
                __bf16* row = &buf[(r0 + 64 * i) * LDR + kq * 4];
                *reinterpret_cast<bf16x4*>(row) = h;
                *reinterpret_cast<bf16x4*>(row + KC) = m;
                *reinterpret_cast<bf16x4*>(row + 2 * KC) = l;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x4 h, m, l;
                split<0>(rb[slot][i], h, m, l);
                __bf16* row = &buf[(BM + r0 + 64 * i) * LDR + kq * 4];
                *reinterpret_cast<bf16x4*>(row) = h;
                *reinterpret_cast<bf16x4*>(row + KC) = m;
                *reinterpret_cast<bf16x4*>(row + 2 * KC) = l;
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(d, d);
        stage(0, buf0);
        issue(0, DEPTH);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the stores have landed
        __builtin_amdgcn_s_barrier();           // #1: chunk 0 is in buf0
        stage(1 % DEPTH, buf1);
        issue(1 % DEPTH, DEPTH + 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();           // #2: chunk 1 is in buf1, the consumers hold chunk 0's fragments
        // iteration c: chunk c + 2 -> buf[c & 1]; the loop is unrolled over the register slots (DEPTH) and the two buffers
        static_assert(DEPTH == 2 || DEPTH == 4, "depth");
        for (int c = 0; c < nch; c += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                stage((u + 2) % DEPTH, (u & 1) ? buf1 : buf0);
                issue((u + 2) % DEPTH, c + u + 2 + DEPTH);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
        }
        return;
    }
    // ---------------- consumer ----------------
    const int wm = wave >> 1, wn = wave & 1, lrow = lane & 31, lh = lane >> 5;
    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa0[MR][3], fb0[NR][3], fa1[MR][3], fb1[NR][3];
    auto rd = [&](bf16x8 (&fa)[MR][3], bf16x8 (&fb)[NR][3], const __bf16* buf) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[j][p] = *reinterpret_cast<const bf16x8*>(&buf[BM * LDR + ((wn * NR + j) * 32 + lrow) * LDR + p * KC + lh * 8]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[i][p] = *reinterpret_cast<const bf16x8*>(&buf[((wm * MR + i) * 32 + lrow) * LDR + p * KC + lh * 8]);
    };
    auto mm = [&](const bf16x8 (&fa)[MR][3], const bf16x8 (&fb)[NR][3]) {
#define S3_MM(pa, pb)                                                                       \
    _Pragma("unroll") for (int i = 0; i < MR; ++i) _Pragma("unroll") for (int j = 0; j < NR; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa], fb[j][pb], acc[i][j], 0, 0, 0)
        S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
#undef S3_MM
    };
    auto mix = [&]() {
        if constexpr (SG) {
#pragma unroll
            for (int t = 0; t < 24; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (t < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };
    __builtin_amdgcn_s_barrier();               // #1
    rd(fa0, fb0, buf0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();               // #2
    for (int c = 0; c < nch; c += 2) {
        mm(fa0, fb0);
        rd(fa1, fb1, buf1);
        mix();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        mm(fa1, fb1);
        rd(fa0, fb0, buf0);
        mix();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
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


// mode 19: the library loop on a 256 x 128 tile with EIGHT waves (4 x 2, each 64 x 64): the B rows are staged once for twice the MFMAs
// (staging work per MFMA x 0.75); 86 KB of LDS, one workgroup per CU (two waves per SIMD as before)
template <int WPE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_s3big(const float* __restrict__ A, const float* __restrict__ B,
                                                                                             float* __restrict__ C, int Mrows, int Ncols, int K, int a_bytes,
                                                                                             int b_bytes, int tiles_m, int tiles_n) {
    constexpr int BM = 256, BN = 128, MR = 2, NR = 2, WN = 2, BUF = (BM + BN) * LDR;
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * BUF];
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int m0 = (w % tiles_m) * BM, z = w / tiles_m;
    const int nch = K / KC;
    // staging: 384 rows x 4 quads = 1536 float4 per chunk = 3 per thread: rows r0, r0 + 128, r0 + 256 of the [A | B] stack
    const int kq = tid & 3, r0 = tid >> 2;
    int off[3];
    bool isb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = r0 + 128 * i;
        isb[i] = r >= BM;
        off[i] = isb[i] ? ((z * Ncols + min(n0 + r - BM, Ncols - 1)) * K + kq * 4) * 4 : ((z * Mrows + min(m0 + r, Mrows - 1)) * K + kq * 4) * 4;
    }
    auto issue_loads = [&](f32x4 (&r)[3], int c) {
        const bool on = c < nch;
        r[0] = bload4(rA, on ? off[0] + c * KC * 4 : 0x7fffffff);
        r[1] = bload4(rA, on ? off[1] + c * KC * 4 : 0x7fffffff);
        r[2] = bload4(rB, on ? off[2] + c * KC * 4 : 0x7fffffff);
    };
    auto stage = [&](const f32x4 (&r)[3], __bf16* buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bf16x4 h, m, l;
            split<0>(r[i], h, m, l);
            __bf16* row = &buf[(r0 + 128 * i) * LDR + kq * 4];
            *reinterpret_cast<bf16x4*>(row) = h;
            *reinterpret_cast<bf16x4*>(row + KC) = m;
            *reinterpret_cast<bf16x4*>(row + 2 * KC) = l;
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
        bf16x8 bf[NR][3], af[MR][3];
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(&buf[(BM + (wn * NR + j) * 32 + lrow) * LDR + p * KC + lh * 8]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(&buf[((wm * MR + i) * 32 + lrow) * LDR + p * KC + lh * 8]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
#define S3_MM(pa, pb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0)
                S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
#undef S3_MM
            }
    };
    __bf16* const buf0 = smem;
    __bf16* const buf1 = smem + BUF;
    f32x4 q0[3], q1[3];
    issue_loads(q0, 0);
    issue_loads(q1, 1);
    stage(q0, buf0);
    issue_loads(q0, 2);
    __syncthreads();
    for (int c = 0; c < nch; c += 2) {
        stage(q1, buf1);
        compute(buf0);
        issue_loads(q1, c + 3);
        __syncthreads();
        stage(q0, buf0);
        compute(buf1);
        issue_loads(q0, c + 4);
        __syncthreads();
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


// mode 20: SPLIT ON READ.  LDS holds the fp32 operand rows as they come from memory (80-byte rows: 16 k + pad; four 16-byte stores per thread and
// chunk instead of twelve 8-byte ones, 40 KB instead of 57); a wave reads its fragments as fp32 (two ds_read_b128 per 8 k) and splits them in
// registers right before the MFMAs -- every value is split by the two waves that use it (twice the VALU work), 16 instead of 22 bytes move per value
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_s3sor(const float* __restrict__ A, const float* __restrict__ B,
                                                                                              float* __restrict__ C, int Mrows, int Ncols, int K, int a_bytes,
                                                                                              int b_bytes, int tiles_m, int tiles_n) {
    constexpr int BM = 128, BN = 128, MR = 2, NR = 2, WN = 2, LDF = 20, BUF = (BM + BN) * LDF;      // LDF floats per row (16 + 4 pad)
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int m0 = (w % tiles_m) * BM, z = w / tiles_m;
    const int nch = K / KC;
    const int kq = tid & 3, r0 = tid >> 2;
    int a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = ((z * Mrows + min(m0 + r0 + 64 * i, Mrows - 1)) * K + kq * 4) * 4;
        b_off[i] = ((z * Ncols + min(n0 + r0 + 64 * i, Ncols - 1)) * K + kq * 4) * 4;
    }
    auto issue_loads = [&](f32x4 (&ra)[2], f32x4 (&rb)[2], int c) {
        const bool on = c < nch;
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = bload4(rA, on ? a_off[i] + c * KC * 4 : 0x7fffffff);
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = bload4(rB, on ? b_off[i] + c * KC * 4 : 0x7fffffff);
    };
    auto stage = [&](const f32x4 (&ra)[2], const f32x4 (&rb)[2], float* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&buf[(r0 + 64 * i) * LDF + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&buf[(BM + r0 + 64 * i) * LDF + kq * 4]) = rb[i];
    };
    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto frag = [&](const float* rowp, bf16x8 (&f)[3]) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(rowp), hi = *reinterpret_cast<const f32x4*>(rowp + 4);
        bf16x4 h0, m0_, l0, h1, m1, l1;
        split_pairs<false>(lo, h0, m0_, l0);
        split_pairs<false>(hi, h1, m1, l1);
        f[0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[1] = __builtin_shufflevector(m0_, m1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[2] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto compute = [&](const float* buf) {
        bf16x8 bf[NR][3], af[MR][3];
#pragma unroll
        for (int j = 0; j < NR; ++j) frag(&buf[(BM + (wn * NR + j) * 32 + lrow) * LDF + lh * 8], bf[j]);
#pragma unroll
        for (int i = 0; i < MR; ++i) frag(&buf[((wm * MR + i) * 32 + lrow) * LDF + lh * 8], af[i]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
#define S3_MM(pa, pb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0)
                S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
#undef S3_MM
            }
    };
    float* const buf0 = smem;
    float* const buf1 = smem + BUF;
    f32x4 ra0[2], rb0[2], ra1[2], rb1[2];
    issue_loads(ra0, rb0, 0);
    issue_loads(ra1, rb1, 1);
    stage(ra0, rb0, buf0);
    issue_loads(ra0, rb0, 2);
    __syncthreads();
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


// mode 22: mode 20 with 32-wide K chunks: a row piece is a whole 128-byte line (one load instruction = 8 rows x 128 B instead of 16 rows x 64 B),
// two MFMA k-steps per barrier; 144-byte LDS rows, 72 KB per workgroup.
// (mode 20's comment follows)  SPLIT ON READ.  LDS holds the fp32 operand rows as they come from memory (80-byte rows: 16 k + pad; four 16-byte stores per thread and
// chunk instead of twelve 8-byte ones, 40 KB instead of 57); a wave reads its fragments as fp32 (two ds_read_b128 per 8 k) and splits them in
// registers right before the MFMAs -- every value is split by the two waves that use it (twice the VALU work), 16 instead of 22 bytes move per value
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_s3sor32(const float* __restrict__ A, const float* __restrict__ B,
                                                                                              float* __restrict__ C, int Mrows, int Ncols, int K, int a_bytes,
                                                                                              int b_bytes, int tiles_m, int tiles_n) {
    constexpr int BM = 128, BN = 128, MR = 2, NR = 2, WN = 2, KC2 = 32, LDF = 36, BUF = (BM + BN) * LDF;      // LDF floats per row (32 + 4 pad)
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * BN;
    w /= tiles_n;
    const int m0 = (w % tiles_m) * BM, z = w / tiles_m;
    const int nch = K / KC2;
    const int kq = tid & 7, r0 = tid >> 3;
    int a_off[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = ((z * Mrows + min(m0 + r0 + 32 * i, Mrows - 1)) * K + kq * 4) * 4;
        b_off[i] = ((z * Ncols + min(n0 + r0 + 32 * i, Ncols - 1)) * K + kq * 4) * 4;
    }
    auto issue_loads = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], int c) {
        const bool on = c < nch;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = bload4(rA, on ? a_off[i] + c * KC2 * 4 : 0x7fffffff);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = bload4(rB, on ? b_off[i] + c * KC2 * 4 : 0x7fffffff);
    };
    auto stage = [&](const f32x4 (&ra)[4], const f32x4 (&rb)[4], float* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&buf[(r0 + 32 * i) * LDF + kq * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&buf[(BM + r0 + 32 * i) * LDF + kq * 4]) = rb[i];
    };
    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto frag = [&](const float* rowp, bf16x8 (&f)[3]) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(rowp), hi = *reinterpret_cast<const f32x4*>(rowp + 4);
        bf16x4 h0, m0_, l0, h1, m1, l1;
        split_pairs<false>(lo, h0, m0_, l0);
        split_pairs<false>(hi, h1, m1, l1);
        f[0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[1] = __builtin_shufflevector(m0_, m1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[2] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto compute = [&](const float* buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bf[NR][3], af[MR][3];
#pragma unroll
            for (int j = 0; j < NR; ++j) frag(&buf[(BM + (wn * NR + j) * 32 + lrow) * LDF + ks * 16 + lh * 8], bf[j]);
#pragma unroll
            for (int i = 0; i < MR; ++i) frag(&buf[((wm * MR + i) * 32 + lrow) * LDF + ks * 16 + lh * 8], af[i]);
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
    float* const buf0 = smem;
    float* const buf1 = smem + BUF;
    f32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    issue_loads(ra0, rb0, 0);
    issue_loads(ra1, rb1, 1);
    stage(ra0, rb0, buf0);
    issue_loads(ra0, rb0, 2);
    __syncthreads();
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


// mode 21: NO BARRIERS.  Every wave stages ITS OWN operands (the 64 A rows and 64 B rows of its 64 x 64 quarter of the tile) into a private
// double-buffered LDS region as fp32 rows and splits on read (mode 20): each operand tile is loaded by the two waves that use it (twice the
// global loads, from L2 the second time), and a wave only ever waits for its own loads and its own LDS traffic (in order per wave: no
// synchronisation at all).  20 KB of LDS per wave, 80 per workgroup: two workgroups = eight independent pipelines per CU.
template <int WPE, int DEPTH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_s3nb(const float* __restrict__ A, const float* __restrict__ B,
                                                                                             float* __restrict__ C, int Mrows, int Ncols, int K, int a_bytes,
                                                                                             int b_bytes, int tiles_m, int tiles_n) {
    constexpr int MR = 2, NR = 2, WN = 2, LDF = 20, WROWS = 128, WBUF = WROWS * LDF;      // per wave: 64 A rows then 64 B rows, LDF floats each
    __shared__ __attribute__((aligned(16))) float smem[4 * 2 * WBUF];
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (w % tiles_n) * 128;
    w /= tiles_n;
    const int m0 = (w % tiles_m) * 128, z = w / tiles_m;
    const int nch = K / KC;
    const int kq = lane & 3, r0 = lane >> 2;                 // 16 rows per pass, 8 passes: 4 of A, 4 of B
    int off[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        off[i] = ((z * Mrows + min(m0 + wm * 64 + r0 + 16 * i, Mrows - 1)) * K + kq * 4) * 4;
        off[4 + i] = ((z * Ncols + min(n0 + wn * 64 + r0 + 16 * i, Ncols - 1)) * K + kq * 4) * 4;
    }
    float* const mybuf = smem + wave * 2 * WBUF;
    auto issue_loads = [&](f32x4 (&r)[8], int c) {
        const bool on = c < nch;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = bload4(rA, on ? off[i] + c * KC * 4 : 0x7fffffff);
#pragma unroll
        for (int i = 4; i < 8; ++i) r[i] = bload4(rB, on ? off[i] + c * KC * 4 : 0x7fffffff);
    };
    auto stage = [&](const f32x4 (&r)[8], float* buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&buf[(r0 + 16 * i) * LDF + kq * 4]) = r[i];
    };
    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto frag = [&](const float* rowp, bf16x8 (&f)[3]) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(rowp), hi = *reinterpret_cast<const f32x4*>(rowp + 4);
        bf16x4 h0, m0_, l0, h1, m1, l1;
        split_pairs<false>(lo, h0, m0_, l0);
        split_pairs<false>(hi, h1, m1, l1);
        f[0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[1] = __builtin_shufflevector(m0_, m1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[2] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto compute = [&](const float* buf) {
        bf16x8 bf[NR][3], af[MR][3];
#pragma unroll
        for (int j = 0; j < NR; ++j) frag(&buf[(64 + j * 32 + lrow) * LDF + lh * 8], bf[j]);
#pragma unroll
        for (int i = 0; i < MR; ++i) frag(&buf[(i * 32 + lrow) * LDF + lh * 8], af[i]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) {
#define S3_MM(pa, pb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0)
                S3_MM(0, 0); S3_MM(0, 1); S3_MM(1, 0); S3_MM(1, 1); S3_MM(0, 2); S3_MM(2, 0);
#undef S3_MM
            }
    };
    float* const buf0 = mybuf;
    float* const buf1 = mybuf + WBUF;
    f32x4 q0[8], q1[8];
    issue_loads(q0, 0);
    issue_loads(q1, 1);
    stage(q0, buf0);
    issue_loads(q0, 2);
    // same software pipeline as the library loop, minus the barriers (LDS operations of one wave complete in order)
    for (int c = 0; c < nch; c += 2) {
        stage(q1, buf1);
        compute(buf0);
        issue_loads(q1, c + 3);
        stage(q0, buf0);
        compute(buf1);
        issue_loads(q0, c + 4);
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

static void presplit(const std::vector<float>& src, std::vector<__bf16>& dst, long rows, int K) {
    const int nch = K / 16;
    dst.resize((size_t)rows * nch * 48);
    for (long r = 0; r < rows; ++r)
        for (int c = 0; c < nch; ++c)
            for (int k = 0; k < 16; ++k) {
                const float v = src[r * K + c * 16 + k];
                const __bf16 a = (__bf16)v;
                const float r1 = v - (float)a;
                const __bf16 b = (__bf16)r1;
                const float r2 = r1 - (float)b;
                __bf16* o = &dst[((size_t)r * nch + c) * 48];
                o[k] = a;
                o[16 + k] = b;
                o[32 + k] = (__bf16)r2;
            }
}

static int g_share = 0;
template <int MODE, int WPE>
static void run(const char* name, const float* A, const float* B, const __bf16* Ap, const __bf16* Bp, float* C, int Z, int M, int N, int K,
                const std::vector<double>* ref, std::vector<float>* host_c) {
    const int tm = (M + 127) / 128, tn = (N + 127) / 128;
    const dim3 grid(Z * tm * tn);
    const int ab = (int)((long)Z * M * K * 4), bb = (int)((long)Z * N * K * 4), apb = (int)((long)Z * M * K * 6), bpb = (int)((long)Z * N * K * 6);
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL((k_s3<MODE, WPE>), grid, dim3(256), 0, 0, A, B, Ap, Bp, C, M, N, K, ab, bb, apb, bpb, tm, tn, g_share);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 100;
    hipEventRecord(e0, 0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_s3<MODE, WPE>), grid, dim3(256), 0, 0, A, B, Ap, Bp, C, M, N, K, ab, bb, apb, bpb, tm, tn, g_share);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double fl = 2.0 * Z * (double)(tm * 128) * (tn * 128) * K;
    double err = -1;
    if (ref && host_c) {
        hipMemcpy(host_c->data(), C, host_c->size() * 4, hipMemcpyDeviceToHost);
        double mx = 0, sc = 0;
        for (size_t i = 0; i < ref->size(); ++i) {
            mx = std::max(mx, std::abs((double)(*host_c)[i] - (*ref)[i]));
            sc = std::max(sc, std::abs((*ref)[i]));
        }
        err = mx / sc;
    }
    printf("mode %2d wpe %d %-44s %8.1f us  %7.1f TFLOP/s fp32-equivalent  %7.1f bf16 issued  err %.2e  %s\n", MODE, WPE, name, us, fl / us * 1e-6,
           6 * fl / us * 1e-6, err, hipGetErrorString(hipGetLastError()));
}


template <int DEPTH, int SG>
static void run_pc(const char* name, const float* A, const float* B, float* C, int Z, int M, int N, int K, const std::vector<double>* ref,
                   std::vector<float>* host_c) {
    const int tm = (M + 127) / 128, tn = (N + 127) / 128;
    const dim3 grid(Z * tm * tn);
    const int ab = (int)((long)Z * M * K * 4), bb = (int)((long)Z * N * K * 4);
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL((k_s3pc<DEPTH, SG>), grid, dim3(512), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn, g_share);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 100;
    hipEventRecord(e0, 0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_s3pc<DEPTH, SG>), grid, dim3(512), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn, g_share);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double fl = 2.0 * Z * (double)(tm * 128) * (tn * 128) * K;
    double err = -1;
    if (ref && host_c) {
        hipMemcpy(host_c->data(), C, host_c->size() * 4, hipMemcpyDeviceToHost);
        double mx = 0, sc = 0;
        for (size_t i = 0; i < ref->size(); ++i) {
            mx = std::max(mx, std::abs((double)(*host_c)[i] - (*ref)[i]));
            sc = std::max(sc, std::abs((*ref)[i]));
        }
        err = mx / sc;
    }
    printf("mode 12 depth %d sg %d %-38s %8.1f us  %7.1f TFLOP/s fp32-equivalent  %7.1f bf16 issued  err %.2e  %s\n", DEPTH, SG, name, us,
           fl / us * 1e-6, 6 * fl / us * 1e-6, err, hipGetErrorString(hipGetLastError()));
}

static void run_big(const float* A, const float* B, float* C, int Z, int M, int N, int K, const std::vector<double>* ref, std::vector<float>* host_c) {
    const int tm = (M + 255) / 256, tn = (N + 127) / 128;
    const dim3 grid(Z * tm * tn);
    const int ab = (int)((long)Z * M * K * 4), bb = (int)((long)Z * N * K * 4);
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL((k_s3big<2>), grid, dim3(512), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 100;
    hipEventRecord(e0, 0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_s3big<2>), grid, dim3(512), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double fl = 2.0 * Z * (double)(((M + 127) / 128) * 128) * (tn * 128) * K;
    hipMemcpy(host_c->data(), C, host_c->size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, sc = 0;
    for (size_t i = 0; i < ref->size(); ++i) {
        mx = std::max(mx, std::abs((double)(*host_c)[i] - (*ref)[i]));
        sc = std::max(sc, std::abs((*ref)[i]));
    }
    printf("mode 19 256 x 128 tile, eight waves, one workgroup per CU   %8.1f us  %7.1f TFLOP/s fp32-equivalent  err %.2e  (%d workgroups) %s\n", us,
           fl / us * 1e-6, mx / sc, (int)grid.x, hipGetErrorString(hipGetLastError()));
}

template <int WPE>
static void run_sor(const float* A, const float* B, float* C, int Z, int M, int N, int K, const std::vector<double>* ref, std::vector<float>* host_c) {
    const int tm = (M + 127) / 128, tn = (N + 127) / 128;
    const dim3 grid(Z * tm * tn);
    const int ab = (int)((long)Z * M * K * 4), bb = (int)((long)Z * N * K * 4);
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL((k_s3sor<WPE>), grid, dim3(256), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 100;
    hipEventRecord(e0, 0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_s3sor<WPE>), grid, dim3(256), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double fl = 2.0 * Z * (double)(tm * 128) * (tn * 128) * K;
    hipMemcpy(host_c->data(), C, host_c->size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, sc = 0;
    for (size_t i = 0; i < ref->size(); ++i) {
        mx = std::max(mx, std::abs((double)(*host_c)[i] - (*ref)[i]));
        sc = std::max(sc, std::abs((*ref)[i]));
    }
    printf("mode 20 wpe %d split on read (fp32 rows in LDS)                 %8.1f us  %7.1f TFLOP/s fp32-equivalent  err %.2e  %s\n", WPE, us,
           fl / us * 1e-6, mx / sc, hipGetErrorString(hipGetLastError()));
}

template <int WPE>
static void run_sor32(const float* A, const float* B, float* C, int Z, int M, int N, int K, const std::vector<double>* ref, std::vector<float>* host_c) {
    const int tm = (M + 127) / 128, tn = (N + 127) / 128;
    const dim3 grid(Z * tm * tn);
    const int ab = (int)((long)Z * M * K * 4), bb = (int)((long)Z * N * K * 4);
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL((k_s3sor32<WPE>), grid, dim3(256), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 100;
    hipEventRecord(e0, 0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_s3sor32<WPE>), grid, dim3(256), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double fl = 2.0 * Z * (double)(tm * 128) * (tn * 128) * K;
    hipMemcpy(host_c->data(), C, host_c->size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, sc = 0;
    for (size_t i = 0; i < ref->size(); ++i) {
        mx = std::max(mx, std::abs((double)(*host_c)[i] - (*ref)[i]));
        sc = std::max(sc, std::abs((*ref)[i]));
    }
    printf("mode 22 wpe %d split on read, 32-wide chunks (whole lines)      %8.1f us  %7.1f TFLOP/s fp32-equivalent  err %.2e  %s\n", WPE, us,
           fl / us * 1e-6, mx / sc, hipGetErrorString(hipGetLastError()));
}

template <int WPE>
static void run_nb(const float* A, const float* B, float* C, int Z, int M, int N, int K, const std::vector<double>* ref, std::vector<float>* host_c) {
    const int tm = (M + 127) / 128, tn = (N + 127) / 128;
    const dim3 grid(Z * tm * tn);
    const int ab = (int)((long)Z * M * K * 4), bb = (int)((long)Z * N * K * 4);
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL((k_s3nb<WPE, 2>), grid, dim3(256), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 100;
    hipEventRecord(e0, 0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_s3nb<WPE, 2>), grid, dim3(256), 0, 0, A, B, C, M, N, K, ab, bb, tm, tn);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double fl = 2.0 * Z * (double)(tm * 128) * (tn * 128) * K;
    hipMemcpy(host_c->data(), C, host_c->size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, sc = 0;
    for (size_t i = 0; i < ref->size(); ++i) {
        mx = std::max(mx, std::abs((double)(*host_c)[i] - (*ref)[i]));
        sc = std::max(sc, std::abs((*ref)[i]));
    }
    printf("mode 21 wpe %d no barriers: wave-private staging, split on read   %8.1f us  %7.1f TFLOP/s fp32-equivalent  err %.2e  %s\n", WPE, us,
           fl / us * 1e-6, mx / sc, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int Z = argc > 1 ? atoi(argv[1]) : 16, M = argc > 2 ? atoi(argv[2]) : 7744, N = argc > 3 ? atoi(argv[3]) : 128,
              K = argc > 4 ? atoi(argv[4]) : 256;
    printf("Z %d  M %d  N %d  K %d   (%d workgroups)\n", Z, M, N, K, Z * ((M + 127) / 128) * ((N + 127) / 128));
    std::vector<float> hA((size_t)Z * M * K), hB((size_t)Z * N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = rnd() * (1.f + 0.37f * rnd());
    for (auto& v : hB) v = rnd() * (1.f + 0.37f * rnd());
    std::vector<__bf16> hAp, hBp;
    presplit(hA, hAp, (long)Z * M, K);
    presplit(hB, hBp, (long)Z * N, K);
    float *A, *B, *C;
    __bf16 *Ap, *Bp;
    hipMalloc(&A, hA.size() * 4);
    hipMalloc(&B, hB.size() * 4);
    hipMalloc(&Ap, hAp.size() * 2);
    hipMalloc(&Bp, hBp.size() * 2);
    hipMalloc(&C, (size_t)Z * M * N * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(Ap, hAp.data(), hAp.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(Bp, hBp.data(), hBp.size() * 2, hipMemcpyHostToDevice);
    // float64 reference of batch 0's first 128 x N block only (cheap), compared on the whole C of batch 0 rows < 128
    std::vector<double> ref((size_t)128 * N);
    for (int m = 0; m < 128; ++m)
        for (int n = 0; n < N; ++n) {
            double a = 0;
            for (int k = 0; k < K; ++k) a += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
            ref[(size_t)m * N + n] = a;
        }
    std::vector<float> hc((size_t)128 * N);
    run<0, 2>("library loop", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<1, 2>("split = register moves", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    run<2, 2>("no split, no LDS stores", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    run<3, 2>("LDS reads + MFMA + barriers", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    run<4, 2>("MFMA only", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    run<5, 2>("B pre-split by LDS-DMA", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<6, 2>("A and B pre-split by LDS-DMA", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<6, 3>("A and B pre-split by LDS-DMA", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<7, 2>("fragments double-buffered in registers", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<8, 2>("  + sched_group_barrier interleave", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<9, 2>("library loop + fence behind the barriers", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<10, 2>("mode 8 + fence behind the barriers", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<1, 2>("split = register moves", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    run<13, 2>("  the same stores at linear addresses", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    run<14, 2>("  six 16-byte stores, linear addresses", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    run<16, 2>("library loop, split on pairs (packed converts)", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<17, 2>("  ... its subtractions kept unpacked (asm)", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run_big(A, B, C, Z, M, N, K, &ref, &hc);
    run_sor32<2>(A, B, C, Z, M, N, K, &ref, &hc);
    run_nb<2>(A, B, C, Z, M, N, K, &ref, &hc);
    run_sor<2>(A, B, C, Z, M, N, K, &ref, &hc);
    run_sor<3>(A, B, C, Z, M, N, K, &ref, &hc);
    run<18, 2>("split one half-iteration ahead of its stores", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<0, 2>("library loop (between)", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run<15, 2>("library loop, loads FOUR chunks ahead", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    run_pc<2, 0>("producer / consumer waves", A, B, C, Z, M, N, K, &ref, &hc);
    run_pc<4, 0>("producer / consumer waves", A, B, C, Z, M, N, K, &ref, &hc);
    run_pc<4, 1>("producer / consumer waves", A, B, C, Z, M, N, K, &ref, &hc);
    g_share = 1;
    run_pc<4, 1>("producer / consumer, the same A rows", A, B, C, Z, M, N, K, nullptr, nullptr);
    run<0, 2>("library loop, every workgroup the SAME A rows (L2)", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    run<8, 2>("mode 8, the same A rows", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    run<2, 2>("mode 2, the same A rows", A, B, Ap, Bp, C, Z, M, N, K, nullptr, nullptr);
    g_share = 0;
    run<0, 2>("library loop (again)", A, B, Ap, Bp, C, Z, M, N, K, &ref, &hc);
    return 0;
}
