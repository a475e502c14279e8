// conv_bf16.hip -- the 4x4 convolution pair on bf16 TENSORS (module.set_precision('bf16') with bf16 activation storage;
// reference layers: nn.Conv2d(4,2,1) / nn.ConvTranspose2d(4,2,1) of unet.py:13-18,36-41 and disc.py:26-45, whose autocast-free
// fp32 arithmetic this mode trades for bf16 multiplies with fp32 accumulation).
//
// One implicit-GEMM kernel, two directions:
//   dir 0 (big -> small):  out[m][a]    = sum_{tap, b} big[pix(m, tap)][b]   * W0[tap][a][b]        K = 16 * Cb
//   dir 1 (small -> big):  out[m'][b]   = sum_{t, a}   small[pix(m', t)][a]  * W1[tap(t)][b][a]     K = 4 * Ca per parity class
// Both operands are K-contiguous bf16 rows, so a 64-wide K chunk of a tile row is 128 contiguous bytes in HBM: the tiles go
// global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 B per lane, no staging registers, no ds_write pass), zero padding
// comes from the buffer descriptor's range check (an out-of-range lane writes zeros), and fragments are read with ds_read_b128.
// LDS rows are 128 B linear; the 16-byte slot index is XOR-swizzled with (row >> 1) & 7 on the SOURCE address (the DMA
// destination is wave-uniform base + lane * 16) and on the read address: conflict-free for ds_read_b128's 16-lane groups.
//
// What bounds it (MI355X_MICROARCH.md, cycle constants): a 1-KiB DMA piece costs ~60-100 cycles of the issuing SIMD's MFMA
// stream and a v_mfma_f32_32x32x16_bf16 32 cycles, so the flops per staged byte decide the ceiling: a 256 x 128 workgroup tile
// with 128 x 64 per wave (8 MFMA tiles = 128 accumulator registers) gives 12 pieces against 32 MFMAs per wave and chunk, half
// the staging of the 128 x 128 / 64 x 64-per-wave kernels in conv_gemm.hip, and one fragment read serves 2.7 MFMAs instead of 2.
// The product is formed transposed (D[channel][pixel], weights as the MFMA's A operand): a lane then owns 4 consecutive
// channels of one pixel per accumulator quad, the two halves of the wave exchange quads (v_permlane32_swap) and every store is
// 16 bytes of one pixel row -- 8 bf16 channels -- instead of sixteen 2-byte stores.
#include "pg_common.h"
#include "conv_bf16.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct XGeom {
    int N, Hb, Wb, Hs, Ws, Ca, Cb, s;
};

constexpr int BK = 64;                  // K chunk (bf16 elements): 128-byte LDS rows
constexpr int ROWB = BK * 2;            // bytes per LDS row

__device__ __forceinline__ float act_epi(float v, int act) {
    if (act == PG_ACT_NONE) return v;
    if (act == PG_ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    if (act == PG_ACT_RELU) return v > 0.f ? v : 0.f;
    return pg_act(v, act);
}

__device__ __forceinline__ unsigned pack2(float x, float y) {
    bf16x2 v;
    v[0] = (__bf16)x;
    v[1] = (__bf16)y;
    return __builtin_bit_cast(unsigned, v);
}

// one LDS-DMA piece: 64 lanes x 16 bytes, global (per-lane byte offset, out of range -> zeros) -> LDS (wave-uniform base + lane * 16).
// (A __device__ helper on purpose: written inline in the __global__ template the builtin's LDS-pointer argument fails the HOST
// pass's type check silently and the kernel's host stub is never emitted.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)lds, 16, byte_off, 0, 0, 0);
}

template <int MR, int NR, int WM, int WN, int DRC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_conv_bf16x(const __bf16* __restrict__ in, int ld_in, const __bf16* __restrict__ W,
                                                    void* __restrict__ out, int ld_out, long slab_stride, XGeom g, int cps,
                                                    const float* __restrict__ bias, int act, int in_bytes, int w_bytes, int out_bf,
                                                    int tiles_n) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int AP = BM / 32, BP = BN / 32;                      // DMA pieces (8 rows x 128 B) per wave and chunk
    __shared__ __attribute__((aligned(1024))) char smem[(BM + BN) * ROWB];
    char* const As = smem;
    char* const Bs = smem + BM * ROWB;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;

    // direction-dependent view of the problem: rows = output pixels (of this parity class), Cin / Cout, local taps
    const int ncls = (DRC == 1 && g.s == 2) ? 4 : 1;
    const int cls = blockIdx.z % ncls, slice = blockIdx.z / ncls;
    const int ah = (ncls == 4) ? (cls >> 1) : 0, aw = (ncls == 4) ? (cls & 1) : 0;
    const int T = (DRC == 1 && g.s == 2) ? 2 : 4, Tsh = (T == 2) ? 1 : 2;
    const int Hc = (DRC == 0) ? g.Hs : (g.s == 2 ? (g.Hb - ah + 1) / 2 : g.Hb);
    const int Wc = (DRC == 0) ? g.Ws : (g.s == 2 ? (g.Wb - aw + 1) / 2 : g.Wb);
    const int Mc = g.N * Hc * Wc;
    const int Cin = (DRC == 0) ? g.Cb : g.Ca, Cout = (DRC == 0) ? g.Ca : g.Cb;
    const int Hin = (DRC == 0) ? g.Hb : g.Hs, Win = (DRC == 0) ? g.Wb : g.Ws;
    const int kh0 = (DRC == 1 && g.s == 2) ? (1 - ah) : 0, kw0 = (DRC == 1 && g.s == 2) ? (1 - aw) : 0;

    const int wk = pg_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = wk / tiles_n, tn = wk - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= Mc) return;                                          // a smaller parity class (odd Hb / Wb): whole workgroup
    const int nchunks = T * T * Cin / BK;
    const int c_begin = slice * cps, c_end = min(nchunks, c_begin + cps);

    // ---- per-lane DMA sources: piece i of this wave covers tile rows (wave * AP + i) * 8 .. + 7, lane -> (row, 16-byte slot)
    int a_off[AP];
    unsigned a_mask[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int r = (wave * AP + i) * 8 + (lane >> 3);
        const int cc = (lane & 7) ^ ((r >> 1) & 7);               // logical 8-element K group stored in this slot
        const int m = m0 + r, mm = min(m, Mc - 1);
        const int n = mm / (Hc * Wc);
        const int rem = mm - n * (Hc * Wc);
        const int ii = rem / Wc, jj = rem - ii * Wc;
        unsigned wv = 0, mask = 0;
        if (DRC == 0) {                                            // tap (kh, kw) reads big pixel (s * ii - 1 + kh, s * jj - 1 + kw)
            const int h0 = g.s * ii - 1, w0 = g.s * jj - 1;
            a_off[i] = (((n * Hin + h0) * Win + w0) * ld_in + cc * 8) * 2;
#pragma unroll
            for (int t = 0; t < 4; ++t) wv |= ((unsigned)(w0 + t) < (unsigned)Win) ? (1u << t) : 0u;
#pragma unroll
            for (int t = 0; t < 4; ++t) mask |= ((unsigned)(h0 + t) < (unsigned)Hin) ? (wv << (4 * t)) : 0u;
        } else {                                                   // local tap (th, tw) reads small pixel (ib - th, jb - tw)
            const int ib = (g.s == 2) ? ii + ah : ii + 1, jb = (g.s == 2) ? jj + aw : jj + 1;
            a_off[i] = (((n * Hin + ib) * Win + jb) * ld_in + cc * 8) * 2;
#pragma unroll
            for (int t = 0; t < 4; ++t) wv |= (t < T && (unsigned)(jb - t) < (unsigned)Win) ? (1u << t) : 0u;
#pragma unroll
            for (int t = 0; t < 4; ++t) mask |= (t < T && (unsigned)(ib - t) < (unsigned)Hin) ? (wv << (T * t)) : 0u;
        }
        a_mask[i] = (m < Mc) ? mask : 0u;
    }
    int b_off[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        const int r = (wave * BP + j) * 8 + (lane >> 3);
        const int cc = (lane & 7) ^ ((r >> 1) & 7);
        const int n = n0 + r;
        b_off[j] = (n < Cout) ? (n * Cin + cc * 8) * 2 : (int)0x80000000u;
    }
    char* const a_dst = As + wave * AP * 8 * ROWB;
    char* const b_dst = Bs + wave * BP * 8 * ROWB;

    // ---- fragment read addresses: row = 32 * tile + lrow, so the swizzle term depends on lrow only
    const int sw = (lrow >> 1) & 7;
    int slot[BK / 16];
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) slot[ks] = lrow * ROWB + (((ks * 2 + lh) ^ sw) << 4);
    const char* const a_rd = As + wm * MR * 32 * ROWB;
    const char* const b_rd = Bs + wn * NR * 32 * ROWB;

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int tl = (c_begin * BK) / Cin;                                 // local tap and first input channel of the chunk (uniform)
    int cin0 = c_begin * BK - tl * Cin;
    const int CC = Cout * Cin;
    for (int c = c_begin; c < c_end; ++c) {
        int a_uni, w_uni;
        if (DRC == 0) {
            a_uni = (((tl >> 2) * Win + (tl & 3)) * ld_in + cin0) * 2;
            w_uni = (tl * CC + cin0) * 2;
        } else {
            const int th = tl >> Tsh, tw = tl & (T - 1);
            a_uni = (cin0 - (th * Win + tw) * ld_in) * 2;
            w_uni = (((kh0 + g.s * th) * 4 + (kw0 + g.s * tw)) * CC + cin0) * 2;
        }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool ok = (a_mask[i] >> tl) & 1u;
            const int off = ok ? a_off[i] + a_uni : (int)0x80000000u;
            dma16(rin, a_dst + i * 8 * ROWB, off);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j)
            dma16(rw, b_dst + j * 8 * ROWB, b_off[j] + w_uni);
        cin0 += BK;
        if (cin0 >= Cin) {
            cin0 = 0;
            ++tl;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_rd + i * 32 * ROWB + slot[ks]);
#pragma unroll
            for (int j = 0; j < NR; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(b_rd + j * 32 * ROWB + slot[ks]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);   // D[channel][pixel]
        }
        __syncthreads();                                           // every wave's reads are done before the next chunk's DMA lands
    }

    // ---- epilogue: lane = pixel lrow of tile i; register r = channel (r & 3) + 8 * (r >> 2) + 4 * lh of tile j
    const bool fin = (slab_stride == 0);
    const int ldo = fin ? ld_out : Cout;
    char* const obase = (char*)out + (fin ? 0L : (long)slice * slab_stride * 4);
    const bool obf = fin && out_bf;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int m = m0 + (wm * MR + i) * 32 + lrow;
        long orow;
        if (DRC == 0) {
            orow = (long)m * ldo;
        } else {
            const int mm = min(m, Mc - 1);
            const int n = mm / (Hc * Wc);
            const int rem = mm - n * (Hc * Wc);
            const int ii = rem / Wc, jj = rem - ii * Wc;
            const int h = (g.s == 2) ? 2 * ii + ah : ii, w = (g.s == 2) ? 2 * jj + aw : jj;
            orow = (long)((n * g.Hb + h) * g.Wb + w) * ldo;
        }
        const bool mok = m < Mc;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int nb = n0 + (wn * NR + j) * 32;
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = nb + 8 * q + 4 * lh;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (fin && bias != nullptr && ch < Cout) bv = *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[i][j][4 * q + e];
                    v[q][e] = fin ? act_epi(x + bv[e], act) : x;
                }
            }
            if (obf) {
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    // blocks q (this half's channels 8q + 4lh ..) and q + 1: after the swap the lower half of the wave holds
                    // channels 8q .. 8q + 7 of its pixel, the upper half 8(q+1) .. 8(q+1) + 7
                    const u32x2 lo = __builtin_amdgcn_permlane32_swap(pack2(v[q][0], v[q][1]), pack2(v[q + 1][0], v[q + 1][1]), false, false);
                    const u32x2 hi = __builtin_amdgcn_permlane32_swap(pack2(v[q][2], v[q][3]), pack2(v[q + 1][2], v[q + 1][3]), false, false);
                    const int ch = nb + 8 * (q + lh);
                    const u32x4 o4 = {lo[0], hi[0], lo[1], hi[1]};
                    if (mok && ch < Cout) *reinterpret_cast<u32x4*>(obase + (orow + ch) * 2) = o4;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = nb + 8 * q + 4 * lh;
                    if (mok && ch < Cout) *reinterpret_cast<f32x4*>(obase + (orow + ch) * 4) = v[q];
                }
            }
        }
    }
}

// P[tap][a][b] fp32 -> bf16, optionally transposing each tap to [b][a] (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void k_pack_w_bf16(const float* __restrict__ P, __bf16* __restrict__ W, int Ca, int Cb, int transpose) {
    if (!transpose) {
        const long total4 = 4L * Ca * Cb;                           // float4 groups
        for (long i = blockIdx.x * 256L + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(P + 4 * i);
            u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(W + 4 * i) = o;
        }
        return;
    }
    __shared__ float tile[32][33];
    const int tb = (Cb + 31) / 32, ta = (Ca + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    for (long t = blockIdx.x; t < 16L * ta * tb; t += gridDim.x) {
        const int tap = (int)(t / (ta * tb));
        const int rem = (int)(t - (long)tap * ta * tb);
        const int a0 = (rem / tb) * 32, b0 = (rem % tb) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = a0 + ty + 8 * k, b = b0 + tx;
            tile[ty + 8 * k][tx] = (a < Ca && b < Cb) ? P[((long)tap * Ca + a) * Cb + b] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = b0 + ty + 8 * k, a = a0 + tx;
            if (a < Ca && b < Cb) W[((long)tap * Cb + b) * Ca + a] = (__bf16)tile[tx][ty + 8 * k];
        }
        __syncthreads();
    }
}

}  // namespace

bool pg_bf16x_geom_ok(int dir, int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride) {
    const int Cin = dir == 0 ? Cb : Ca, Cout = dir == 0 ? Ca : Cb;
    if (Cin % BK != 0 || Cout % 8 != 0 || Cout < 32) return false;
    if (16L * Ca * Cb * 2 >= 0x40000000L) return false;
    const long pix = (long)N * (dir == 0 ? Hs * Ws : Hb * Wb);
    if (pix * Cout >= 0x7fffffffL) return false;
    (void)stride;
    return true;
}

pg_bf16x_plan pg_bf16x_plan_of(int dir, int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride) {
    pg_bf16x_plan p;
    const int Cin = dir == 0 ? Cb : Ca, Cout = dir == 0 ? Ca : Cb;
    p.ncls = (dir == 1 && stride == 2) ? 4 : 1;
    const long Mc = (dir == 0) ? (long)N * Hs * Ws : (p.ncls == 4 ? (long)N * ((Hb + 1) / 2) * ((Wb + 1) / 2) : (long)N * Hb * Wb);
    const int taps = (p.ncls == 4) ? 4 : 16;
    p.nchunks = taps * Cin / BK;
    p.out_elems = (long)N * (dir == 0 ? Hs * Ws : Hb * Wb) * Cout;
    auto blocks = [&](int bm, int bn) { return ((Mc + bm - 1) / bm) * ((Cout + bn - 1) / bn) * p.ncls; };
    static const int forced = getenv("PATCHGAN_BF16X_TILE") ? atoi(getenv("PATCHGAN_BF16X_TILE")) : -1;
    if (forced >= 0 && forced <= 2) p.tile = forced;
    else if (Cout <= 64) p.tile = 2;
    else if (blocks(256, 128) >= 200) p.tile = 0;
    else p.tile = 1;
    p.bm = (p.tile == 1) ? 128 : 256;
    p.bn = (p.tile == 2) ? 64 : 128;
    p.tiles_m = (int)((Mc + p.bm - 1) / p.bm);
    p.tiles_n = (Cout + p.bn - 1) / p.bn;
    const long nb = (long)p.tiles_m * p.tiles_n * p.ncls;
    // two workgroups per CU overlap one's DMA wait with the other's MFMAs: split K until >= 512, at least 4 chunks per slice
    static const int target = getenv("PATCHGAN_BF16X_TARGET") ? atoi(getenv("PATCHGAN_BF16X_TARGET")) : 512;
    long s = (nb >= target) ? 1 : (target + nb - 1) / nb;
    const long smax = std::max<long>(1, p.nchunks / 4);
    if (s > smax) s = smax;
    p.split = (int)s;
    p.cps = (p.nchunks + p.split - 1) / p.split;
    p.split = (p.nchunks + p.cps - 1) / p.cps;
    return p;
}

void pg_bf16x_clamp(pg_bf16x_plan* p, size_t avail) {
    const long smax = (long)(avail / (sizeof(float) * (size_t)p->out_elems));
    if (p->split > 1 && smax < p->split) p->split = smax < 2 ? 1 : (int)smax;
    p->cps = (p->nchunks + p->split - 1) / p->split;
    p->split = (p->nchunks + p->cps - 1) / p->cps;
}

const char* pg_bf16x_kernel_name(int dir, int tile) {
    static const char* const names[2][3] = {{"k_conv_bf16x<4,2,2,2,0>", "k_conv_bf16x<2,2,2,2,0>", "k_conv_bf16x<2,2,4,1,0>"},
                                            {"k_conv_bf16x<4,2,2,2,1>", "k_conv_bf16x<2,2,2,2,1>", "k_conv_bf16x<2,2,4,1,1>"}};
    return names[dir ? 1 : 0][tile < 0 || tile > 2 ? 0 : tile];
}

size_t pg_bf16x_w_bytes(int Ca, int Cb) { return ((size_t)16 * Ca * Cb * 2 + 255) & ~(size_t)255; }

int pg_bf16x_pack(const float* P, void* W, int Ca, int Cb, int dir, hipStream_t st) {
    if (dir == 0) {
        if ((Cb & 3) != 0) return PG_EINVAL;
        const long total4 = 4L * Ca * Cb;
        const int blocks = (int)std::min<long>((total4 + 255) / 256, 2048);
        hipLaunchKernelGGL(k_pack_w_bf16, dim3(blocks), dim3(256), 0, st, P, (__bf16*)W, Ca, Cb, 0);
    } else {
        const long tiles = 16L * ((Ca + 31) / 32) * ((Cb + 31) / 32);
        hipLaunchKernelGGL(k_pack_w_bf16, dim3((int)std::min<long>(tiles, 4096)), dim3(256), 0, st, P, (__bf16*)W, Ca, Cb, 1);
    }
    return pg_launch_status();
}

int pg_bf16x_conv(int dir, const void* in, int ld_in, long in_bytes, const void* W, void* out, int ld_out, long slab_stride,
                  int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride, const pg_bf16x_plan* p, const float* bias,
                  int act, int out_bf, hipStream_t st) {
    const XGeom g{N, Hb, Wb, Hs, Ws, Ca, Cb, stride};
    const dim3 grid((unsigned)(p->tiles_m * p->tiles_n), 1, (unsigned)(p->ncls * p->split));
    const int w_bytes = 16 * Ca * Cb * 2;
    const __bf16* I = (const __bf16*)in;
    const __bf16* Wp = (const __bf16*)W;
#define PG_BF16X_LAUNCH(MR, NR, WM, WN)                                                                                      \
    do {                                                                                                                     \
        if (dir == 0)                                                                                                        \
            hipLaunchKernelGGL((k_conv_bf16x<MR, NR, WM, WN, 0>), grid, dim3(256), 0, st, I, ld_in, Wp, out, ld_out, slab_stride, \
                               g, p->cps, bias, act, (int)in_bytes, w_bytes, out_bf, p->tiles_n);                            \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_conv_bf16x<MR, NR, WM, WN, 1>), grid, dim3(256), 0, st, I, ld_in, Wp, out, ld_out, slab_stride, \
                               g, p->cps, bias, act, (int)in_bytes, w_bytes, out_bf, p->tiles_n);                            \
    } while (0)
    switch (p->tile) {
        case 0: PG_BF16X_LAUNCH(4, 2, 2, 2); break;
        case 1: PG_BF16X_LAUNCH(2, 2, 2, 2); break;
        default: PG_BF16X_LAUNCH(2, 2, 4, 1); break;
    }
#undef PG_BF16X_LAUNCH
    return pg_launch_status();
}
