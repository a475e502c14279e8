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
#include <type_traits>

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

constexpr int BK = 64;                  // K granularity of the plan (bf16 elements): Cin % 64 == 0, split-K slices in 64-chunks

__device__ __forceinline__ float act_epi(float v, int act) {
    if (act == PG_ACT_NONE) return v;
    if (act == PG_ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    if (act == PG_ACT_RELU) return v > 0.f ? v : 0.f;
    return pg_act(v, act);
}

// The epilogues below run over 64 accumulators per lane, fully unrolled.  With the activation as a RUN-TIME value every element carried
// its own compare-and-branch ladder and an inlined tanhf / expf: ~7000 instructions (56 KB of code) per kernel, executed in 5-14 us per
// tile -- MORE than the MFMA loop of a two-chunk layer (measured with the phase stamps of PG_TRACE_R: main loop 4.7 us, epilogue 14 us on
// the 128 -> 64 channel data gradient).  The activation and the multiplier's activation are therefore dispatched ONCE per tile into a
// body compiled for that value (tag -1: the run-time ladder, kept for the combination nothing on the training path uses).
template <int A>
using pg_ic = std::integral_constant<int, A>;
template <int ACT>
__device__ __forceinline__ float act_sel(float v, int act) {
    if constexpr (ACT < 0) return act_epi(v, act);
    else return pg_act(v, ACT);
}
template <int MACT>
__device__ __forceinline__ float act_grad_c(float a, int mact) {
    if constexpr (MACT < 0) return pg_act_grad_sel(a, mact);
    else return pg_act_grad_sel(a, MACT);
}
template <bool MUL, class F>
__device__ __forceinline__ void epi_dispatch(int act, int mact, F&& f) {
    if constexpr (MUL) {
        if (act != PG_ACT_NONE) {
            f(pg_ic<-1>{}, pg_ic<-1>{});
            return;
        }
        switch (mact) {
            case PG_ACT_LEAKY: f(pg_ic<PG_ACT_NONE>{}, pg_ic<PG_ACT_LEAKY>{}); break;
            case PG_ACT_RELU: f(pg_ic<PG_ACT_NONE>{}, pg_ic<PG_ACT_RELU>{}); break;
            case PG_ACT_TANH: f(pg_ic<PG_ACT_NONE>{}, pg_ic<PG_ACT_TANH>{}); break;
            case PG_ACT_SIGMOID: f(pg_ic<PG_ACT_NONE>{}, pg_ic<PG_ACT_SIGMOID>{}); break;
            default: f(pg_ic<PG_ACT_NONE>{}, pg_ic<PG_ACT_NONE>{}); break;
        }
    } else {
        switch (act) {
            case PG_ACT_LEAKY: f(pg_ic<PG_ACT_LEAKY>{}, pg_ic<PG_ACT_NONE>{}); break;
            case PG_ACT_RELU: f(pg_ic<PG_ACT_RELU>{}, pg_ic<PG_ACT_NONE>{}); break;
            case PG_ACT_TANH: f(pg_ic<PG_ACT_TANH>{}, pg_ic<PG_ACT_NONE>{}); break;
            case PG_ACT_SIGMOID: f(pg_ic<PG_ACT_SIGMOID>{}, pg_ic<PG_ACT_NONE>{}); break;
            default: f(pg_ic<PG_ACT_NONE>{}, pg_ic<PG_ACT_NONE>{}); break;
        }
    }
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
// 8 consecutive K rows of one column as an MFMA bf16 fragment from a [K][column] LDS image: two transposing reads
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_frag2(const char* p, int second) {   // rows p and p + second (4 pixel rows further)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + second));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KB = 64: one LDS buffer of 128-byte rows, all of a chunk's DMA, wait, barrier, 4 MFMA k-steps, barrier; two workgroups per CU
//          cover each other's DMA phase.
// KB = 32: a ring of three stages of 64-byte rows: chunk c + 2 is in flight while chunk c is multiplied, one barrier per chunk,
//          counted vmcnt (never 0 in the loop).
// MUL: the epilogue also multiplies by f'(t) (pg_epi_mul: the activation backward of the layer below a data gradient) -- its own
// instantiation with unconditional, clamped loads of t (a conditional load inside the unrolled epilogue sends the accumulators to scratch)
// STATS: the epilogue also emits per-channel sums / sums of squares of the values it STORES (bf16-rounded) for the InstanceNorm that
// follows: part[((n * chunks + chunk) * Cout + c) * 2 + {0, 1}], fp64, one chunk per workgroup tile (the host guarantees
// that a tile lies inside one sample and that the K loop is not split).  Per 32-channel column block: in-lane sums over the wave's
// pixel tiles, a transpose through LDS (lane <-> value, fixed order), the waves that share the channels combined in wave order.
// BT (dir 1): the weights come in the SAME packed layout as dir 0 ([tap][a][b], b contiguous): the B tile is staged [k = a][n = b] and its
// fragments come out of transposing reads, so one bf16 copy of a layer's weights serves both directions.
// OCC: waves per SIMD = workgroups per CU.  The 64-accumulator tiles (128 x 128, 256 x 64) fit four (<= 128 registers, 32 - 40 KB of
// LDS each): three other workgroups' DMA phases then cover a workgroup's multiply phase instead of one (cfg4 layers, same box:
// enc1 forward 61 -> 51 us, dec5 data gradient 104 -> 80 us, d1 forward at 2N 156 -> 117 us; EXPERIMENTS.md round 4)
template <int MR, int NR, int WM, int WN, int DRC, int KB, bool MUL = false, bool STATS = false, bool BT = false,
          int OCC = (MR * NR <= 4 && KB == 64) ? 4 : 2>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_conv_bf16x(
    const __bf16* __restrict__ in, int ld_in, const __bf16* __restrict__ W, void* __restrict__ out, int ld_out, long slab_stride, XGeom g,
    int cps, const float* __restrict__ bias, int act, int in_bytes, int w_bytes, int out_bf, int tiles_n, pg_epi_mul mul, double* __restrict__ part,
    int chunks, int cls_in_x) {
    constexpr int NW = WM * WN;                                    // waves per workgroup: 4, or 8 (the 256 x 128 tile of 64 x 64 per wave:
    static_assert(NW == 4 || NW == 8, "four or eight waves");      //  0.375 DMA pieces per MFMA at four waves per SIMD from two workgroups)
    static_assert(KB == 32 || KB == 64, "chunk width");
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
    constexpr int ROWB = KB * 2;                                   // bytes per LDS row
    constexpr int RPP = 1024 / ROWB, SPR = ROWB / 16;              // rows per DMA piece, 16-byte slots per row
    constexpr int AP = BM / (NW * RPP), BP = BN / (NW * RPP);      // pieces per wave and chunk
    constexpr int NST = (KB == 32) ? 3 : 1, STAGE = (BM + BN) * ROWB;
    constexpr int STATB = STATS ? (NW * 64 * 33 + NW * 64) * 4 : 0; // STATS: [wave][lane][33] transpose area + [wave][64] results
    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE > STATB ? NST * STAGE : STATB];
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;

    // direction-dependent view of the problem: rows = output pixels (of this parity class), Cin / Cout, local taps
    // DRC 0: big -> small; 1: small -> big; 2: big -> small where `big` has 8 channels per pixel (ld_in == 8; an image-facing tensor
    // padded to one 16-byte DMA piece per pixel): the 16 taps x 8 channels of an output pixel ARE its K = 128 row, one tap per
    // LDS slot; 3: plain row GEMM out[m][n] = sum_k in[m][k] * W[n][k] over the N * Hs * Ws rows of `in` (K = Ca, n < Cb):
    // the first half of the taps-folded-into-N ConvTranspose2d of the few-channel heads (second half: k_col2im_small2big)
    constexpr bool B2S = (DRC == 0 || DRC == 2 || DRC == 3);
    const int ncls = (DRC == 1 && g.s == 2) ? 4 : 1;
    // cls_in_x: the four parity classes of a tile position are neighbours in the work index (same XCD, in flight together), so the
    // `small` window they all read comes from that XCD's L2 three times out of four instead of from the fabric once per class
    const int wk0 = pg_xcd_remap(blockIdx.x, gridDim.x);
    const int cls = cls_in_x ? (wk0 & (ncls - 1)) : blockIdx.z % ncls, slice = cls_in_x ? blockIdx.z : blockIdx.z / ncls;
    const int ah = (ncls == 4) ? (cls >> 1) : 0, aw = (ncls == 4) ? (cls & 1) : 0;
    const int T = (DRC == 3) ? 1 : (DRC == 1 && g.s == 2) ? 2 : 4, Tsh = (T == 2) ? 1 : 2;
    const int Hc = B2S ? g.Hs : (g.s == 2 ? (g.Hb - ah + 1) / 2 : g.Hb);
    const int Wc = B2S ? g.Ws : (g.s == 2 ? (g.Wb - aw + 1) / 2 : g.Wb);
    const int Mc = g.N * Hc * Wc;
    const int Cin = (DRC == 0) ? g.Cb : (DRC == 2) ? 128 : g.Ca, Cout = (DRC == 0 || DRC == 2) ? g.Ca : g.Cb;
    const int Hin = (DRC == 0 || DRC == 2) ? g.Hb : g.Hs, Win = (DRC == 0 || DRC == 2) ? g.Wb : g.Ws;
    const int kh0 = (DRC == 1 && g.s == 2) ? (1 - ah) : 0, kw0 = (DRC == 1 && g.s == 2) ? (1 - aw) : 0;
    constexpr int TPC = KB / 8;                                    // DRC 2: taps per chunk

    const int wk = (cls_in_x && ncls == 4) ? (wk0 >> 2) : wk0;
    const int tm = wk / tiles_n, tn = wk - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= Mc) return;                                          // a smaller parity class (odd Hb / Wb): whole workgroup
    constexpr int CPB = BK / KB;                                   // kernel chunks per plan chunk
    const int nchunks = (DRC == 2) ? 128 / KB : T * T * Cin / KB;
    const int c_begin = slice * cps * CPB, c_end = min(nchunks, c_begin + cps * CPB);

    // ---- per-lane DMA sources: piece i of this wave covers tile rows (wave * AP + i) * RPP .. + RPP - 1, lane -> (row, slot);
    //      slot s of row r holds the logical 8-element K group s ^ swz(r)
    auto swz = [](int r) { return KB == 64 ? (r >> 1) & 7 : (r >> 2) & 3; };
    int a_off[AP];
    unsigned a_mask[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int r = (wave * AP + i) * RPP + lane / SPR;
        const int cc = (lane % SPR) ^ swz(r);
        const int m = m0 + r, mm = min(m, Mc - 1);
        const int n = mm / (Hc * Wc);
        const int rem = mm - n * (Hc * Wc);
        const int ii = rem / Wc, jj = rem - ii * Wc;
        unsigned wv = 0, mask = 0;
        if (DRC == 3) {
            a_off[i] = (mm * ld_in + cc * 8) * 2;
            mask = 1u;
        } else if (DRC == 0 || DRC == 2) {                         // tap (kh, kw) reads big pixel (s * ii - 1 + kh, s * jj - 1 + kw)
            const int h0 = g.s * ii - 1, w0 = g.s * jj - 1;
            a_off[i] = (((n * Hin + h0) * Win + w0) * ld_in + (DRC == 2 ? 0 : cc * 8)) * 2;
#pragma unroll
            for (int t = 0; t < 4; ++t) wv |= ((unsigned)(w0 + t) < (unsigned)Win) ? (1u << t) : 0u;
#pragma unroll
            for (int t = 0; t < 4; ++t) mask |= ((unsigned)(h0 + t) < (unsigned)Hin) ? (wv << (4 * t)) : 0u;
            if (DRC == 2) {                                        // slot cc of chunk c holds tap c * TPC + cc: the lane's own (kh, kw) offset
                a_off[i] += (((cc >> 2) * Win + (cc & 3)) * ld_in) * 2;
                mask >>= cc;
            }
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
    static_assert(!BT || (DRC == 1 && KB == 64), "transposed B staging: small -> big, 64-wide chunks");
    constexpr int RBB = BN * 2, RPB = 1024 / RBB, LPRB = RBB / 16;        // BT: B rows = the chunk's 64 input channels, BN * 2 bytes each
    auto swzT = [](int r) { return RBB >= 256 ? (r & 3) << 2 : ((r >> 1) & 1) << 2; };
    int b_off[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        if constexpr (BT) {
            const int r = (wave * BP + j) * RPB + lane / LPRB;     // (BP == 64 / RPB / NW for every tile width)
            const int ch = (lane % LPRB) ^ swzT(r);
            const int b = n0 + ch * 8;
            b_off[j] = (b < Cout) ? (r * Cout + b) * 2 : (int)0x80000000u;
        } else {
            const int r = (wave * BP + j) * RPP + lane / SPR;
            const int cc = (lane % SPR) ^ swz(r);
            const int n = n0 + r;
            b_off[j] = (n < Cout) ? (n * Cin + cc * 8) * 2 : (int)0x80000000u;
        }
    }
    char* const a_dst = smem + wave * AP * 1024;
    char* const b_dst = smem + BM * ROWB + wave * BP * 1024;

    // ---- fragment read addresses: row = 32 * tile + lrow, so the swizzle term depends on lrow only
    int slot[KB / 16];
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks) slot[ks] = lrow * ROWB + (((ks * 2 + lh) ^ swz(lrow)) << 4);
    const char* const a_rd = smem + wm * MR * 32 * ROWB;
    const char* const b_rd = smem + BM * ROWB + wn * NR * 32 * ROWB;
    int b_rdt[BT ? NR : 1];                                        // BT: transposed-read addresses (as k_wgrad_bf16x)
    if constexpr (BT) {
        const int q4 = (lane & 15) >> 2, g4 = lane >> 4;
        const int chl = (g4 & 1) * 2 + ((lane & 3) >> 1), half8 = (lane & 1) * 8;
#pragma unroll
        for (int j = 0; j < NR; ++j) b_rdt[j] = (lh * 8 + q4) * RBB + ((((wn * NR + j) * 4 + chl) ^ swzT(q4)) << 4) + half8;
    }

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // chunk order: taps fastest, channel chunk slowest -- the 16 (4) taps of one channel chunk re-read the same pixels' 128 bytes, a
    // working set of a few tens of KB per workgroup that stays in the XCD's L2 (taps outermost streamed the whole input 4 - 16 x
    // from beyond L2: 590 MB instead of 69 MB per launch on the stride-1 layer)
    const int ntap = T * T;
    int tl = (DRC == 2) ? c_begin * TPC : c_begin % ntap;          // local tap and first input channel of the next chunk to load
    int cin0 = (DRC == 2) ? 0 : (c_begin / ntap) * KB;
    const int CC = Cout * Cin;
    auto issue = [&](int stage, bool on) {                         // DMA of that chunk into `stage`; off: zeros (keeps vmcnt counts)
        int a_uni, w_uni;
        if (DRC == 3) {
            a_uni = cin0 * 2;
            w_uni = cin0 * 2;
        } else if (DRC == 2) {                                     // here tl = first tap of the chunk (a multiple of TPC)
            a_uni = ((tl >> 2) * Win * ld_in) * 2;
            w_uni = (tl * 8) * 2;
        } else if (DRC == 0) {
            a_uni = (((tl >> 2) * Win + (tl & 3)) * ld_in + cin0) * 2;
            w_uni = (tl * CC + cin0) * 2;
        } else {
            const int th = tl >> Tsh, tw = tl & (T - 1);
            a_uni = (cin0 - (th * Win + tw) * ld_in) * 2;
            w_uni = BT ? (((kh0 + g.s * th) * 4 + (kw0 + g.s * tw)) * CC + cin0 * Cout) * 2
                       : (((kh0 + g.s * th) * 4 + (kw0 + g.s * tw)) * CC + cin0) * 2;
        }
        const unsigned kill = on ? 0u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool ok = (a_mask[i] >> tl) & 1u;
            const int off = ok ? (int)((unsigned)(a_off[i] + a_uni) | kill) : (int)0x80000000u;
            dma16(rin, a_dst + stage * STAGE + i * 1024, off);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) dma16(rw, b_dst + stage * STAGE + j * 1024, (int)((unsigned)(b_off[j] + w_uni) | kill));
        if (DRC == 2) {
            tl += TPC;
        } else {
            ++tl;
            if (tl == ntap) {
                tl = 0;
                cin0 += KB;
            }
        }
    };
    auto multiply = [&](int stage) {
#pragma unroll
        for (int ks = 0; ks < KB / 16; ++ks) {
            bf16x8 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_rd + stage * STAGE + i * 32 * ROWB + slot[ks]);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                if constexpr (BT)
                    bf[j] = tr_frag2(smem + BM * ROWB + stage * STAGE + ks * 16 * RBB + b_rdt[j], 4 * RBB);
                else
                    bf[j] = *reinterpret_cast<const bf16x8*>(b_rd + stage * STAGE + j * 32 * ROWB + slot[ks]);
            }
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);   // D[channel][pixel]
        }
    };

    if constexpr (KB == 64) {
        for (int c = c_begin; c < c_end; ++c) {
            issue(0, true);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            multiply(0);
            __syncthreads();                                       // every wave's reads are done before the next chunk's DMA lands
        }
    } else {
        const int nloc = c_end - c_begin;
        issue(0, 0 < nloc);
        issue(1, 1 < nloc);
        int stage = 0;
        for (int it = 0; it < nloc; ++it) {
            wait_vmcnt<AP + BP>();                                 // this wave's pieces of chunk `it` have landed (chunk it + 1 may be in flight)
            __builtin_amdgcn_s_barrier();                          // ... and everyone's; stage (it + 2) % 3 was last read in iteration it - 1
            const int s2 = (stage >= 1) ? stage - 1 : 2;           // == (it + 2) % 3
            issue(s2, it + 2 < nloc);
            multiply(stage);
            stage = (stage == 2) ? 0 : stage + 1;
        }
        wait_vmcnt<0>();                                           // the killed tail DMAs drain before the kernel ends
    }

    // ---- epilogue: lane = pixel lrow of tile i; register r = channel (r & 3) + 8 * (r >> 2) + 4 * lh of tile j
    const bool fin = (slab_stride == 0);
    const int ldo = fin ? ld_out : Cout;
    char* const obase = (char*)out + (fin ? 0L : (long)slice * slab_stride * 4);
    const bool obf = fin && out_bf;
    long opix[MR];
    bool mok[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int m = m0 + (wm * MR + i) * 32 + lrow;
        if (B2S) {
            opix[i] = m;
        } else {
            const int mm = min(m, Mc - 1);
            const int n = mm / (Hc * Wc);
            const int rem = mm - n * (Hc * Wc);
            const int ii = rem / Wc, jj = rem - ii * Wc;
            const int h = (g.s == 2) ? 2 * ii + ah : ii, w = (g.s == 2) ? 2 * jj + aw : jj;
            opix[i] = (long)((n * g.Hb + h) * g.Wb + w);
        }
        mok[i] = m < Mc;
    }
    if constexpr (STATS) __syncthreads();                          // the operand tiles in LDS are dead: the statistics reuse the space
    auto body = [&](auto act_tag, auto mact_tag) {
    constexpr int ACT = decltype(act_tag)::value, MACT = decltype(mact_tag)::value;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int nb = n0 + (wn * NR + j) * 32;
        float s1[STATS ? 16 : 1], s2[STATS ? 16 : 1];
        if constexpr (STATS) {
#pragma unroll
            for (int k = 0; k < 16; ++k) s1[k] = s2[k] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const long orow = opix[i] * ldo;
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = nb + 8 * q + 4 * lh;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (fin && bias != nullptr && ch < Cout) bv = *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[i][j][4 * q + e];
                    v[q][e] = fin ? act_sel<ACT>(x + bv[e], act) : x;
                }
                if constexpr (MUL) {           // data gradient times the activation derivative of the layer below (t: its output)
                    const long tidx = (mok[i] ? opix[i] : 0L) * mul.ld + min(ch, Cout - 4);
                    f32x4 tv;
                    if (obf) {
                        const u32x2 h = *reinterpret_cast<const u32x2*>((const char*)mul.t + tidx * 2);
                        tv = f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
                                   __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
                    } else {
                        tv = *reinterpret_cast<const f32x4*>((const char*)mul.t + tidx * 4);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q][e] *= act_grad_c<MACT>(tv[e], mul.act);
                }
                if constexpr (STATS) {         // what InstanceNorm will read back: the bf16-rounded value
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r = mok[i] ? (float)(__bf16)v[q][e] : 0.f;
                        s1[4 * q + e] += r;
                        s2[4 * q + e] += r * r;
                    }
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
                    if (mok[i] && ch < Cout) *reinterpret_cast<u32x4*>(obase + (orow + ch) * 2) = o4;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = nb + 8 * q + 4 * lh;
                    if (mok[i] && ch < Cout) *reinterpret_cast<f32x4*>(obase + (orow + ch) * 4) = v[q];
                }
            }
        }
        if constexpr (STATS) {
            // lane-major [lane][33] image of the 32 per-lane sums (s1 | s2), then lane L adds column L & 31 over the 32 lanes of its
            // half in lane order: sums over the wave's MR * 32 pixels for channel nb + 8q + 4 * half + e, k = 4q + e = (L & 15)
            float* const tr = reinterpret_cast<float*>(smem) + wave * 64 * 33;
            float* const res = reinterpret_cast<float*>(smem) + NW * 64 * 33 + wave * 64;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                tr[lane * 33 + k] = s1[k];
                tr[lane * 33 + 16 + k] = s2[k];
            }
            __syncthreads();
            float t = 0.f;
#pragma unroll
            for (int l = 0; l < 32; ++l) t += tr[(lh * 32 + l) * 33 + lrow];
            res[lane] = t;
            __syncthreads();
            if (wm == 0) {                                         // the WM waves that share these channels, in wave order
                float tot = t;
#pragma unroll
                for (int w2 = 1; w2 < WM; ++w2) tot += reinterpret_cast<const float*>(smem)[NW * 64 * 33 + (w2 * WN + wn) * 64 + lane];
                const int which = lrow >> 4, k = lrow & 15;
                const int ch = nb + 8 * (k >> 2) + 4 * lh + (k & 3);
                // the tile's sample and chunk: every row of the tile lies in one sample (host-checked)
                const int hw = Hc * Wc;
                const int nsmp = m0 / hw, chunk = cls * (hw / BM) + (m0 - nsmp * hw) / BM;
                if (ch < Cout) part[(((long)nsmp * chunks + chunk) * Cout + ch) * 2 + which] = (double)tot;
            }
            __syncthreads();
        }
    }
    };
    epi_dispatch<MUL>(fin ? act : PG_ACT_NONE, mul.act, body);
}

// ------------------------------------------------------------------------------------------------------------------------
// Window-staged form of the stride-2 layers (round 4): pixels through an LDS window, weights straight into registers.
// What bounds the kernel above inside the training step is not its MFMA rate and not its bytes (a first form of this kernel that only
// staged the A side once per four taps moved 35 % fewer bytes in the same time) but the per-tap cycle itself -- DMA, wait for it to
// land (~1-2 us from HBM / the Infinity Cache under load), barrier, 16 MFMAs, barrier -- whose waits only the OTHER workgroups of the CU
// cover.  Here:
//   * a workgroup owns an R x 16 rectangle of output (class) pixels.  The four taps of a parity class (dir 1), or of one
//     input-parity group (dir 0: taps (2 dh + ph, 2 dw + pw) read big pixels (2 (i + dh) - 1 + ph, 2 (j + dw) - 1 + pw): the same
//     2 x 2-tap pattern on the (ph, pw) sub-image), read ONE (R + 1) x 17 window of pixels shifted by a pixel: it is staged once per
//     64-channel chunk by LDS-DMA (19.6 KB instead of 4 x 16 KB) into one of TWO buffers -- the next chunk's window lands while this
//     one is multiplied -- and the taps read their fragments from it at tap-shifted row addresses;
//   * the weight fragments never touch LDS: a lane's MFMA operand is 8 consecutive K values of one output channel = 16 contiguous bytes
//     of the packed weights W[tap][n][k], loaded with one buffer_load_dwordx4 per fragment a whole tap ahead (two register sets);
//   * so a macro chunk (4 taps, 64 MFMAs per wave) costs ONE barrier and no exposed wait in steady state, against 8 barriers and 4
//     exposed waits above.
// DRC 0: big -> small; DRC 1: small -> big, class = blockIdx.z % 4; both take weights with K contiguous per output channel (dir 1: the
// per-tap transposed pack, pg_bf16x_pack dir 1).
#ifdef PG_TRACE_R
// Diagnostics build only (make trace -> libpatchgan_hip_trace.so, tools/trace_r.py): per wave, real-time-clock (100 MHz) stamps of the phases of
// k_conv_bf16r -- entry, first barrier passed, time spent waiting at the later chunk starts, time issuing a chunk's MFMAs, epilogue
// start, end -- into a buffer set with pg_debug_trace_set.  Not part of the product library.
__device__ unsigned long long* pg_trace_buf = nullptr;
#define PG_TR_NOW() __builtin_amdgcn_s_memrealtime()      // 100 MHz, the same on every XCD
#endif
template <int MR, int NR, int WM, int WN, int DRC, bool MUL = false, bool STATS = false, int OCC = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_conv_bf16r(
    const __bf16* __restrict__ in, int ld_in, const __bf16* __restrict__ W, void* __restrict__ out, int ld_out, long slab_stride, XGeom g,
    int cps, const float* __restrict__ bias, int act, int in_bytes, int w_bytes, int out_bf, int tiles_n, pg_epi_mul mul, double* __restrict__ part,
    int chunks, int cls_in_x) {
    static_assert(WM * WN == 4 && (DRC == 0 || DRC == 1), "four waves; stride-2 conv pair");
    constexpr int BM = WM * MR * 32, BN = WN * NR * 32, R = BM / 16;          // tile = R rows x 16 columns of (class) pixels
    constexpr int WC = 17, NWR = (R + 1) * WC;                                 // window: (R + 1) x 17 pixels, one 128-byte LDS row each
    constexpr int WP = (NWR + 31) / 32;                                        // window DMA pieces per wave (8 rows per piece)
    constexpr int WBYTES = WP * 4 * 1024;
    constexpr int STATB = STATS ? (4 * 64 * 33 + 4 * 64) * 4 : 0;
    // epilogue staging (bf16 outputs): a wave's 32 pixel x 32 channel tile goes through LDS so that FOUR lanes store one pixel's 64
    // contiguous bytes -- 16 line requests per store instruction instead of 64 (lane = pixel, 16 bytes each): measured, the stores were
    // 2.2 us of a 256 x 64 tile's 3.6-us epilogue.  80-byte rows: conflict-free 8-byte writes, 16-byte aligned reads.
    constexpr int STG_RS = 80, STG_W = 32 * STG_RS, STG_OFF = STATB;
    // MUL with a bf16 `t`: the tile's BM x BN slice of t is fetched by LDS-DMA DURING the last chunk (into the window buffer that chunk no
    // longer needs, or an area of its own where a window is smaller than the slice) instead of by 16 scattered 8-byte loads per lane in
    // the epilogue, whose latency nothing covered (measured: 7 of the 9.8 us of that epilogue, against 4.7 us of MFMA loop on the 128 -> 64
    // channel data gradient).  Image: [pixel][16-byte slot], slot XOR-swizzled by the pixel (2-way instead of 16-way conflicts on the reads).
    static_assert(!(MUL && STATS), "the multiplier and the statistics share LDS");
    constexpr int TBYTES = BM * BN * 2, TP = TBYTES / 1024 / 4, NSL = BN / 8, PPP = 64 / NSL;
    constexpr bool TDED = MUL && WBYTES < TBYTES;                              // the slice does not fit a window buffer: its own area
    constexpr int SMEM0 = 2 * WBYTES + (TDED ? TBYTES : 0);
    constexpr int SMEM = SMEM0 > STG_OFF + 4 * STG_W ? SMEM0 : STG_OFF + 4 * STG_W;
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lh = lane >> 5;

    const int ncls = (DRC == 1) ? 4 : 1;
    const int wk0 = pg_xcd_remap(blockIdx.x, gridDim.x);               // (cls_in_x: as in k_conv_bf16x)
    const int cls = cls_in_x ? (wk0 & (ncls - 1)) : blockIdx.z % ncls, slice = cls_in_x ? blockIdx.z : blockIdx.z / ncls;
    const int ah = (DRC == 1) ? (cls >> 1) : 0, aw = (DRC == 1) ? (cls & 1) : 0;
    const int Hc = (DRC == 0) ? g.Hs : g.Hb / 2, Wc = (DRC == 0) ? g.Ws : g.Wb / 2;      // host: Hc % R == 0, Wc % 16 == 0
    const int Cin = (DRC == 0) ? g.Cb : g.Ca, Cout = (DRC == 0) ? g.Ca : g.Cb;
    const int Hin = (DRC == 0) ? g.Hb : g.Hs, Win = (DRC == 0) ? g.Wb : g.Ws;
    const int tiles_c = Wc / 16, tiles_r = Hc / R;

    const int wk = (cls_in_x && ncls == 4) ? (wk0 >> 2) : wk0;
    const int tm = wk / tiles_n, tn = wk - tm * tiles_n;
    const int n0 = tn * BN;
    const int nsmp = tm / (tiles_r * tiles_c), trc = tm - nsmp * (tiles_r * tiles_c);
    const int r0 = (trc / tiles_c) * R, c0 = (trc % tiles_c) * 16;
    // K: macro chunks = (group, 64-channel chunk) with 4 taps each; cps counts 64-wide (tap, chunk) units and is a multiple of 4
    const int cch = Cin / 64, ngrp = (DRC == 0) ? 4 : 1;
    const int nmac = ngrp * cch;
    const int m_begin = slice * (cps >> 2), m_end = min(nmac, m_begin + (cps >> 2));

    // ---- window DMA: piece i of this wave covers window rows (wave * WP + i) * 8 .. + 7; lane -> (row, 16-byte slot)
    int w_y[WP], w_x[WP], w_cc[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int w = (wave * WP + i) * 8 + (lane >> 3);
        const int wr = w / WC, wc = w - wr * WC;
        // slot s of window row (wr, wc) holds the logical 8-channel group s ^ (wc & 7): with the 17-row pitch this keeps the tap-shifted
        // fragment reads (16 consecutive columns of row r in lanes 0-15, of row r + 1 in lanes 16-31) conflict-free for ds_read_b128's
        // lane groups (the flat kernel's (row >> 1) & 7 would be 2-way conflicted on EVERY read here: checked by enumeration)
        w_cc[i] = ((lane & 7) ^ (wc & 7)) * 8;
        w_y[i] = (w < NWR) ? wr : -0x10000;                        // beyond the window: never valid
        w_x[i] = wc;
    }
    // ---- weight fragments: the FRAGMENT-ORDERED pack (pg_bf16x_pack dir 4 / 5) W[tap][n tile of 32][64-channel chunk][k-step][lane][8]:
    //      the 64 lanes' operands of one MFMA are 1 KiB contiguous -- one fully coalesced buffer_load_dwordx4 per fragment (rows of a
    //      [n][k] pack would be a 64-line gather per instruction: tried, 1 ms slower per cfg4 step)
    const int NT = (Cout + 31) / 32;
    int b_row[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int nt = n0 / 32 + wn * NR + j;
        b_row[j] = (nt < NT) ? (nt * cch * 4 * 64 + lane) * 16 : (int)0x80000000u;
    }
    char* const w_dst = smem + wave * WP * 1024;

    // ---- fragment rows: MFMA tile i of this wave = tile rows (wm * MR + i) * 32 + lrow -> (rr, cc) -> window row rr * 17 + cc (+ tap shift)
    int wbase[MR], wcol[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int ml = (wm * MR + i) * 32 + lrow;
        wbase[i] = (ml >> 4) * WC + (ml & 15);
        wcol[i] = ml & 15;
    }

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int ST = (DRC == 0) ? 2 : 1;
    const int nsteps = 4 * (m_end - m_begin);
    auto issue_window = [&](int mi) {                              // window of macro chunk m_begin + mi into buffer mi & 1 (beyond the end: zeros)
        const int mc = m_begin + mi;
        const int grp = (DRC == 0) ? mc / cch : 0, cin0 = (mc - grp * cch) * 64;
        const int ph = grp >> 1, pw = grp & 1;
        const int oy = (DRC == 0) ? 2 * (r0 - 1 + ph) + 1 - ph : r0 + ah - 1;     // window origin in the source tensor's coordinates
        const int ox = (DRC == 0) ? 2 * (c0 - 1 + pw) + 1 - pw : c0 + aw - 1;
        const bool live = mc < m_end;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int y = oy + ST * w_y[i], x = ox + ST * w_x[i];
            const bool ok = live && (unsigned)y < (unsigned)Hin && (unsigned)x < (unsigned)Win;
            const int off = (((nsmp * Hin + y) * Win + x) * ld_in + cin0 + w_cc[i]) * 2;
            dma16(rin, w_dst + (mi & 1) * WBYTES + i * 1024, ok ? off : (int)0x80000000u);
        }
    };
    bf16x8 breg[4][NR][4];                                         // one register set per tap position: [tap][column tile][k-step]
    auto load_b = [&](int s, bf16x8 (&dst)[NR][4]) {               // step s = (macro chunk, tap); beyond the end: zeros (keeps the vmcnt counts)
        const int mc = m_begin + (s >> 2), t = s & 3;
        const int grp = (DRC == 0) ? mc / cch : 0, cin0 = (mc - grp * cch) * 64;
        const int ph = grp >> 1, pw = grp & 1;
        const int kh = (DRC == 0) ? 2 * (t >> 1) + ph : (1 - ah) + 2 * (t >> 1);
        const int kw = (DRC == 0) ? 2 * (t & 1) + pw : (1 - aw) + 2 * (t & 1);
        const unsigned kill = (s < nsteps) ? 0u : 0x80000000u;
        const int w_uni = (((kh * 4 + kw) * NT * cch + (cin0 >> 6)) * 4 * 64) * 16;     // (tap, ., chunk, k-step 0) of the fragment-ordered pack
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)((unsigned)(b_row[j] + w_uni + ks * 1024) | kill), 0, 0));
                dst[j][ks] = __builtin_bit_cast(bf16x8, v);
            }
    };
    auto multiply = [&](int s, const bf16x8 (&bsrc)[NR][4]) {
        const int t = s & 3;
        const int e_r = (DRC == 0) ? (t >> 1) : 1 - (t >> 1), e_c = (DRC == 0) ? (t & 1) : 1 - (t & 1);
        const int wsh = e_r * WC + e_c;
        const char* const win = smem + ((s >> 2) & 1) * WBYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[MR];
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int w = wbase[i] + wsh;
                af[i] = *reinterpret_cast<const bf16x8*>(win + w * 128 + (((ks * 2 + lh) ^ ((wcol[i] + e_c) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bsrc[j][ks], af[i], acc[i][j], 0, 0, 0);   // D[channel][pixel]
        }
    };
    // Schedule (loads and LDS-DMA pieces complete in issue order; NB = the NR * 4 fragment loads of one tap).  The fragments of step s
    // live in register set s & 3 and are requested THREE taps ahead (a tap is 16 MFMAs: one tap of distance left the MFMAs waiting on
    // L2); the next chunk's window is requested at tap 0.  Before the MFMAs of tap t the newest requests still allowed in flight:
    //   chunk start: fragments(s0 + 1), (s0 + 2)                                       -> <= 2 NB   (window(mi), fragments(s0) are in)
    //   t = 0: request fragments(s0 + 3), window(mi + 1)
    //   t >= 1: request fragments(s0 + t + 3); fragments(s0 + t) are in once <= 3 NB + WP remain
    constexpr int NB = NR * 4;
#ifdef PG_TRACE_R
    const unsigned long long tr_t0 = PG_TR_NOW();
    unsigned long long tr_t1 = 0, tr_w = 0, tr_m = 0;
#endif
    if (nsteps > 0) {
        issue_window(0);
        load_b(0, breg[0]);
        load_b(1, breg[1]);
        load_b(2, breg[2]);
    }
    const int nmc = nsteps >> 2;                                   // macro chunks of this tile
    const bool tpre = MUL && slab_stride == 0 && out_bf;          // t through LDS (bf16 t; an fp32 t keeps the loads in the epilogue)
    const int t_off = TDED ? 2 * WBYTES : (nmc & 1) * WBYTES;     // the buffer the last chunk does not read
    auto issue_t = [&]() {
        if constexpr (MUL) {
            const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)mul.t, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int k = 0; k < TP; ++k) {
                const int piece = wave * TP + k;
                const int pl = piece * PPP + lane / NSL;           // pixel of the tile; the lane's granule holds slot (lane % NSL) ^ (pl % NSL)
                const int sl = (lane % NSL) ^ (pl & (NSL - 1));
                const int ii = r0 + (pl >> 4), jj = c0 + (pl & 15);
                const long op = (DRC == 0) ? (long)((nsmp * g.Hs + ii) * g.Ws + jj) : (long)((nsmp * g.Hb + 2 * ii + ah) * g.Wb + 2 * jj + aw);
                const int ch = n0 + sl * 8;
                dma16(rt, smem + t_off + piece * 1024, ch < Cout ? (int)((op * mul.ld + ch) * 2) : (int)0x80000000u);
            }
        }
    };
    auto chunk = [&](int mi, auto xp_tag, auto t_tag) {            // XP: the pieces requested at tap 0 (the next window, or the slice of t)
        constexpr int XP = decltype(xp_tag)::value;
        constexpr bool TNOW = decltype(t_tag)::value;
        const int s0 = 4 * mi;
#ifdef PG_TRACE_R
        const unsigned long long tr_a = PG_TR_NOW();
#endif
        wait_vmcnt<2 * NB>();
        __builtin_amdgcn_s_barrier();                              // this chunk's window is complete; everyone is done with the other buffer
#ifdef PG_TRACE_R
        const unsigned long long tr_b = PG_TR_NOW();
        if (mi == 0) tr_t1 = tr_b;
        else tr_w += tr_b - tr_a;
#endif
        load_b(s0 + 3, breg[3]);
        if constexpr (TNOW) issue_t();
        else issue_window(mi + 1);
        multiply(s0, breg[0]);
        load_b(s0 + 4, breg[0]);
        wait_vmcnt<3 * NB + XP>();
        multiply(s0 + 1, breg[1]);
        load_b(s0 + 5, breg[1]);
        wait_vmcnt<3 * NB + XP>();
        multiply(s0 + 2, breg[2]);
        load_b(s0 + 6, breg[2]);
        wait_vmcnt<3 * NB + XP>();
        multiply(s0 + 3, breg[3]);
#ifdef PG_TRACE_R
        tr_m += PG_TR_NOW() - tr_b;
#endif
    };
#pragma unroll 1
    for (int mi = 0; mi + 1 < nmc; ++mi) chunk(mi, pg_ic<WP>{}, std::false_type{});
    if (nmc > 0) {
        if (MUL && tpre) chunk(nmc - 1, pg_ic<(MUL ? TP : WP)>{}, std::integral_constant<bool, MUL>{});
        else chunk(nmc - 1, pg_ic<WP>{}, std::false_type{});
    }
    wait_vmcnt<0>();                                               // the killed tail loads / pieces drain before the kernel ends
    __syncthreads();
#ifdef PG_TRACE_R
    const unsigned long long tr_t2 = PG_TR_NOW();
#endif

    // ---- epilogue (as k_conv_bf16x): lane = pixel lrow of tile i; register r = channel (r & 3) + 8 * (r >> 2) + 4 * lh of tile j
    const bool fin = (slab_stride == 0);
    const int ldo = fin ? ld_out : Cout;
    char* const obase = (char*)out + (fin ? 0L : (long)slice * slab_stride * 4);
    const bool obf = fin && out_bf;
    long opix[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int ml = (wm * MR + i) * 32 + lrow;
        const int ii = r0 + (ml >> 4), jj = c0 + (ml & 15);
        opix[i] = (DRC == 0) ? (long)((nsmp * g.Hs + ii) * g.Ws + jj) : (long)((nsmp * g.Hb + 2 * ii + ah) * g.Wb + 2 * jj + aw);
    }
    if constexpr (STATS) __syncthreads();
    auto body = [&](auto act_tag, auto mact_tag) {                 // (one body per activation: see epi_dispatch)
    constexpr int ACT = decltype(act_tag)::value, MACT = decltype(mact_tag)::value;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int nb = n0 + (wn * NR + j) * 32;
        float s1[STATS ? 16 : 1], s2[STATS ? 16 : 1];
        if constexpr (STATS) {
#pragma unroll
            for (int k = 0; k < 16; ++k) s1[k] = s2[k] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const long orow = opix[i] * ldo;
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = nb + 8 * q + 4 * lh;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (fin && bias != nullptr && ch < Cout) bv = *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[i][j][4 * q + e];
                    v[q][e] = fin ? act_sel<ACT>(x + bv[e], act) : x;
                }
                if constexpr (MUL) {
                    const long tidx = opix[i] * mul.ld + min(ch, Cout - 4);
                    f32x4 tv;
                    if (obf) {
                        const int pl = (wm * MR + i) * 32 + lrow, sl = (wn * NR + j) * 4 + q;       // tile pixel, 16-byte slot of the slice in LDS
                        const u32x2 h = *reinterpret_cast<const u32x2*>(smem + t_off + ((pl * NSL + (sl ^ (pl & (NSL - 1)))) << 4) + lh * 8);
                        tv = f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
                                   __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
                    } else {
                        tv = *reinterpret_cast<const f32x4*>((const char*)mul.t + tidx * 4);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q][e] *= act_grad_c<MACT>(tv[e], mul.act);
                }
                if constexpr (STATS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r = (float)(__bf16)v[q][e];
                        s1[4 * q + e] += r;
                        s2[4 * q + e] += r * r;
                    }
                }
            }
            if (obf) {
                char* const stg = smem + (MUL && !TDED ? ((nmc - 1) & 1) * WBYTES : STG_OFF) + wave * STG_W;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x2 pk = {pack2(v[q][0], v[q][1]), pack2(v[q][2], v[q][3])};
                    *reinterpret_cast<u32x2*>(stg + lrow * STG_RS + (8 * q + 4 * lh) * 2) = pk;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {                      // 128 pieces of 16 bytes: lane -> (pixel, quarter of its 64 bytes)
                    const int c = lane + 64 * k, pp = c >> 2, part = c & 3;
                    const u32x4 o4 = *reinterpret_cast<const u32x4*>(stg + pp * STG_RS + part * 16);
                    const int ml = (wm * MR + i) * 32 + pp;
                    const int ii = r0 + (ml >> 4), jj = c0 + (ml & 15);
                    const long op = (DRC == 0) ? (long)((nsmp * g.Hs + ii) * g.Ws + jj) : (long)((nsmp * g.Hb + 2 * ii + ah) * g.Wb + 2 * jj + aw);
                    const int ch = nb + part * 8;
#ifdef PG_TRACE_NOSTORE
                    asm volatile("" ::"v"(o4), "v"(ch), "v"(op));
#else
                    if (ch < Cout) *reinterpret_cast<u32x4*>(obase + (op * ldo + ch) * 2) = o4;
#endif
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = nb + 8 * q + 4 * lh;
                    if (ch < Cout) *reinterpret_cast<f32x4*>(obase + (orow + ch) * 4) = v[q];
                }
            }
        }
        if constexpr (STATS) {
            float* const tr = reinterpret_cast<float*>(smem) + wave * 64 * 33;
            float* const res = reinterpret_cast<float*>(smem) + 4 * 64 * 33 + wave * 64;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                tr[lane * 33 + k] = s1[k];
                tr[lane * 33 + 16 + k] = s2[k];
            }
            __syncthreads();
            float t = 0.f;
#pragma unroll
            for (int l = 0; l < 32; ++l) t += tr[(lh * 32 + l) * 33 + lrow];
            res[lane] = t;
            __syncthreads();
            if (wm == 0) {
                float tot = t;
#pragma unroll
                for (int w2 = 1; w2 < WM; ++w2) tot += reinterpret_cast<const float*>(smem)[4 * 64 * 33 + (w2 * WN + wn) * 64 + lane];
                const int which = lrow >> 4, k = lrow & 15;
                const int ch = nb + 8 * (k >> 2) + 4 * lh + (k & 3);
                const int chunk = cls * (tiles_r * tiles_c) + trc;     // this tile's slot among the sample's `chunks` partial sums
                if (ch < Cout) part[(((long)nsmp * chunks + chunk) * Cout + ch) * 2 + which] = (double)tot;
            }
            __syncthreads();
        }
    }
    };
    epi_dispatch<MUL>(fin ? act : PG_ACT_NONE, mul.act, body);
#ifdef PG_TRACE_R
    if (pg_trace_buf != nullptr && lane == 0) {
        const unsigned long long tr_t3 = PG_TR_NOW();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* const o = pg_trace_buf + ((size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 4 + wave) * 8;
        o[0] = tr_t0;
        o[1] = tr_t1;
        o[2] = tr_w;
        o[3] = tr_m;
        o[4] = tr_t2;
        o[5] = tr_t3;
        o[6] = ((unsigned long long)xcc << 32) | hw;
        o[7] = (unsigned long long)((nsteps + 3) / 4);
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient on bf16 tensors:  dP[tap][a][b] = sum_m small[m][a] * big[pix(m, tap)][b]  -- per tap a GEMM whose K index is
// the pixel m, the STRIDED index of both operands (channels are the contiguous one).  The tiles therefore sit in LDS as
// [pixel][channel] rows exactly as they lie in HBM (LDS-DMA, 16 B per lane, zero padding by the descriptor's range check) and the
// MFMA fragments -- 8 consecutive pixels of one channel -- come out of ds_read_b64_tr_b16, the transposing read (two per fragment).
// Bank conflicts of the transposed reads: a 32-lane half reads 4 pixel rows x one 64-byte segment (32 channels); rows are 128 /
// 256 / 512 bytes, so the segment index is XOR-swizzled with the row (on the DMA source address and on the read address) to spread
// the four rows over the four quarters of the 256-byte bank window.
// Output transposed (D[b][a]): a lane owns 4 consecutive b of one a per accumulator quad = one 16-byte fp32 store.
// Grid: flat, XCD-remapped so that the 16 taps of one (tile, K slice) run back to back on one XCD: they share the `small` tile and
// read shifted windows of the same `big` pixels from that XCD's L2.
// TAPN: `big` has 8 channels per pixel (ld_big == 8, g.Cb <= 8 real ones): the 16 taps x 8 channels are the 128 b-columns of ONE GEMM
// (no tap dimension in the grid); one DMA lane fetches one tap's pixel (16 bytes), its own (kh, kw) offset instead of the block's.
template <int MR, int NR, int WM, int WN, bool TAPN = false, int OCC = (MR * NR <= 4) ? 4 : 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_wgrad_bf16x(
    const __bf16* __restrict__ small, int ld_small, const __bf16* __restrict__ big, int ld_big, float* __restrict__ out, long slab_stride,
    XGeom g, int cps, int small_bytes, int big_bytes, int tiles_b, int ntiles, float inv_hw, float inv_w) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BA = WM * MR * 32, BB = WN * NR * 32;            // a (small-side) and b (big-side) channels per workgroup
    constexpr int KP = 64;                                         // pixels per chunk
    constexpr int RBA = BA * 2, RBB = BB * 2;                      // LDS row bytes
    constexpr int RPA = 1024 / RBA, RPB = 1024 / RBB;              // pixel rows per DMA piece
    constexpr int AP = KP / RPA / 4, BP = KP / RPB / 4;            // pieces per wave and chunk
    __shared__ __attribute__((aligned(1024))) char smem[KP * (RBA + RBB)];
    char* const As = smem;
    char* const Bs = smem + KP * RBA;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)small, 0, small_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)big, 0, big_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lh = lane >> 5;

    const int wk = pg_xcd_remap(blockIdx.x, gridDim.x);
    static_assert(!TAPN || BB == 128, "taps in N: 16 taps x 8 channels");
    const int tap = TAPN ? 0 : (wk & 15), tile = TAPN ? wk % ntiles : (wk >> 4) % ntiles, slice = TAPN ? wk / ntiles : (wk >> 4) / ntiles;
    const int ta = tile / tiles_b, tb = tile - ta * tiles_b;
    const int a0 = ta * BA, b0 = TAPN ? 0 : tb * BB;
    const int kh = tap >> 2, kw = tap & 3;
    const int HW = g.Hs * g.Ws, M = g.N * HW;
    const int nchunks = (M + KP - 1) / KP;
    const int c_begin = slice * cps, c_end = min(nchunks, c_begin + cps);

    // swizzle of the 64-byte segment index by the pixel row (see above)
    auto swz = [](int r, int rowbytes) { return rowbytes >= 256 ? (r & 3) << 2 : ((r >> 1) & 1) << 2; };
    // ---- DMA sources.  A: pixel rows are consecutive in `small`; B: the tap's window of `big`, decoded per chunk
    int a_off[AP], a_row[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        constexpr int LPR = RBA / 16;
        const int r = (wave * AP + i) * RPA + lane / LPR;
        const int ch = (lane % LPR) ^ swz(r, RBA);
        const int a = a0 + ch * 8;
        a_row[i] = (a < g.Ca) ? r : 0x40000000;                    // beyond the channel count: never valid
        a_off[i] = (r * ld_small + a) * 2;
    }
    int b_col[BP], b_row[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        constexpr int LPR = RBB / 16;
        const int r = (wave * BP + j) * RPB + lane / LPR;
        const int ch = (lane % LPR) ^ swz(r, RBB);
        const int b = b0 + ch * 8;
        b_row[j] = (TAPN || b < g.Cb) ? r : 0x40000000;
        b_col[j] = TAPN ? ch : b * 2;                              // TAPN: the lane's tap
    }
    char* const a_dst = As + wave * AP * 1024;
    char* const b_dst = Bs + wave * BP * 1024;

    // ---- transposed-read addresses: lane -> (pixel row q of its 4-row block, 4 channels), see T10 of the CDNA4 guide
    const int q = (lane & 15) >> 2, g4 = lane >> 4;
    const int chl = (g4 & 1) * 2 + ((lane & 3) >> 1), half8 = (lane & 1) * 8;    // 16-byte chunk within the 32-channel segment
    int a_rd[MR], b_rd[NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int seg = (wm * MR + i);                             // 32-channel segment of the tile
        a_rd[i] = (lh * 8 + q) * RBA + (((seg * 4 + chl) ^ swz(q, RBA)) << 4) + half8;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int seg = (wn * NR + j);
        b_rd[j] = (lh * 8 + q) * RBB + (((seg * 4 + chl) ^ swz(q, RBB)) << 4) + half8;
    }

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef PG_TRACE_R
    const unsigned long long tr_t0 = PG_TR_NOW();
    unsigned long long tr_w = 0, tr_m = 0, tr_b2 = 0, tr_i = 0;
#endif
    for (int c = c_begin; c < c_end; ++c) {
#ifdef PG_TRACE_R
        const unsigned long long tr_a = PG_TR_NOW();
#endif
        const int pix0 = c * KP;
        const int a_uni = pix0 * ld_small * 2;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool ok = pix0 + a_row[i] < M;
            dma16(rs, a_dst + i * 1024, ok ? a_off[i] + a_uni : (int)0x80000000u);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int m = pix0 + b_row[j];
            // m -> (n, p, qq) by float reciprocals, exact after one correction step for m < 2^24
            int n = (int)((float)m * inv_hw);
            int rem = m - n * HW;
            n += (rem >= HW) - (rem < 0);
            rem = m - n * HW;
            int pp = (int)((float)rem * inv_w);
            int qq = rem - pp * g.Ws;
            pp += (qq >= g.Ws) - (qq < 0);
            qq = rem - pp * g.Ws;
            const int h = g.s * pp - 1 + (TAPN ? b_col[j] >> 2 : kh), w = g.s * qq - 1 + (TAPN ? b_col[j] & 3 : kw);
            const bool ok = m < M && (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
            const int off = ((n * g.Hb + h) * g.Wb + w) * ld_big * 2 + (TAPN ? 0 : b_col[j]);
            dma16(rb, b_dst + j * 1024, ok ? off : (int)0x80000000u);
        }
#ifdef PG_TRACE_R
        const unsigned long long tr_i1 = PG_TR_NOW();
        tr_i += tr_i1 - tr_a;
#endif
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#ifdef PG_TRACE_R
        const unsigned long long tr_b = PG_TR_NOW();
        tr_w += tr_b - tr_i1;
#endif
#pragma unroll
        for (int ks = 0; ks < KP / 16; ++ks) {
            bf16x8 af[MR], bf[NR];
#pragma unroll
            for (int i = 0; i < MR; ++i) af[i] = tr_frag2(As + ks * 16 * RBA + a_rd[i], 4 * RBA);
#pragma unroll
            for (int j = 0; j < NR; ++j) bf[j] = tr_frag2(Bs + ks * 16 * RBB + b_rd[j], 4 * RBB);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);   // D[b][a]
        }
#ifdef PG_TRACE_R
        const unsigned long long tr_c = PG_TR_NOW();
        tr_m += tr_c - tr_b;
#endif
        __syncthreads();
#ifdef PG_TRACE_R
        tr_b2 += PG_TR_NOW() - tr_c;
#endif
    }
#ifdef PG_TRACE_R
    const unsigned long long tr_t2 = PG_TR_NOW();
#endif

    // ---- epilogue: lane = channel a (lane & 31) of tile i; register r = channel b (r & 3) + 8 * (r >> 2) + 4 * lh of tile j
    float* const o = out + (long)slice * slab_stride + (long)tap * g.Ca * g.Cb;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int a = a0 + (wm * MR + i) * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int bb = b0 + (wn * NR + j) * 32 + 4 * lh;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int b = bb + 8 * qd;
                if (TAPN) {                                        // column b = tap * 8 + channel: dP[tap][a][channel < Cb]
                    const int tp = b >> 3, c0 = b & 7;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (a < g.Ca && c0 + e < g.Cb) o[((long)tp * g.Ca + a) * g.Cb + c0 + e] = acc[i][j][4 * qd + e];
                } else if (a < g.Ca && b < g.Cb) {
                    const f32x4 v = {acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
                    *reinterpret_cast<f32x4*>(o + (long)a * g.Cb + b) = v;
                }
            }
        }
    }
#ifdef PG_TRACE_R
    if (pg_trace_buf != nullptr && lane == 0 && blockIdx.x < 16384) {
        unsigned long long* const ob = pg_trace_buf + ((size_t)blockIdx.x * 4 + wave) * 8;
        ob[0] = tr_t0;
        ob[1] = tr_i;            // issuing a chunk's DMA pieces (address arithmetic included)
        ob[2] = tr_w;            // waiting for them + the barrier
        ob[3] = tr_m;            // transposed reads + MFMAs
        ob[4] = tr_t2;
        ob[5] = PG_TR_NOW();
        ob[6] = tr_b2;           // the barrier after the MFMAs
        ob[7] = (unsigned long long)(c_end - c_begin);
    }
#endif
}

// P[tap][a][b] fp32 -> bf16, optionally transposing each tap to [b][a] (32 x 32 tiles through LDS).  blk / nblk: this workgroup's index
// in / the size of the (grid-stride) block range that covers the layer -- the whole grid, or one item's share of k_pack_batch's
__device__ __forceinline__ void pack_w_body(const float* __restrict__ P, __bf16* __restrict__ W, int Ca, int Cb, int transpose, int blk,
                                            int nblk, float (*tile)[33]) {
    if (!transpose) {
        const long total4 = 4L * Ca * Cb;                           // float4 groups
        for (long i = blk * 256L + threadIdx.x; i < total4; i += (long)nblk * 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(P + 4 * i);
            u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(W + 4 * i) = o;
        }
        return;
    }
    const int tb = (Cb + 31) / 32, ta = (Ca + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    for (long t = blk; t < 16L * ta * tb; t += nblk) {
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
__global__ __launch_bounds__(256) void k_pack_w_bf16(const float* __restrict__ P, __bf16* __restrict__ W, int Ca, int Cb, int transpose) {
    __shared__ float tile[32][33];
    pack_w_body(P, W, Ca, Cb, transpose, blockIdx.x, gridDim.x, tile);
}

// P[tap][a][b] (b < Cb <= 8) -> W8[a][tap][8] bf16, zero padded: the K = 128 row of output channel a in tap order
__device__ __forceinline__ void pack_w8_body(const float* __restrict__ P, __bf16* __restrict__ W, int Ca, int Cb, int blk, int nblk) {
    for (long i = blk * 256L + threadIdx.x; i < 16L * Ca; i += (long)nblk * 256) {
        const int tap = (int)(i / Ca), a = (int)(i - (long)tap * Ca);
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (c < Cb) ? P[((long)tap * Ca + a) * Cb + c] : 0.f;
        const u32x4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        *reinterpret_cast<u32x4*>(W + ((long)a * 16 + tap) * 8) = o;
    }
}
__global__ __launch_bounds__(256) void k_pack_w8_bf16(const float* __restrict__ P, __bf16* __restrict__ W, int Ca, int Cb) {
    pack_w8_body(P, W, Ca, Cb, blockIdx.x, gridDim.x);
}

// P[tap][a][b] -> the fragment-ordered pack of k_conv_bf16r: W[tap][n tile][64-k chunk][k-step][lane][8], element (n = 32 nt + (lane & 31),
// k = 64 kc + 16 ks + 8 (lane >> 5) + e).  swap = 0: n = a, k = b (big -> small); swap = 1: n = b, k = a (small -> big).  One thread per
// fragment; channels beyond the count are zero.
__device__ __forceinline__ void pack_frag_body(const float* __restrict__ P, __bf16* __restrict__ W, int Ca, int Cb, int swap, int blk, int nblk) {
    const int Cn = swap ? Cb : Ca, Ck = swap ? Ca : Cb;
    const int NT = (Cn + 31) / 32, KC = Ck / 64;
    const long total = 16L * NT * KC * 4 * 64;
    for (long i = blk * 256L + threadIdx.x; i < total; i += (long)nblk * 256) {
        const int lane = (int)(i & 63);
        long r = i >> 6;
        const int ks = (int)(r & 3);
        r >>= 2;
        const int kc = (int)(r % KC);
        r /= KC;
        const int nt = (int)(r % NT), tap = (int)(r / NT);
        const int n = nt * 32 + (lane & 31), k0 = kc * 64 + ks * 16 + (lane >> 5) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (n < Cn) ? (swap ? P[((long)tap * Ca + k0 + e) * Cb + n] : P[((long)tap * Ca + n) * Cb + k0 + e]) : 0.f;
        const u32x4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        *reinterpret_cast<u32x4*>(W + i * 8) = o;
    }
}
__global__ __launch_bounds__(256) void k_pack_frag_bf16(const float* __restrict__ P, __bf16* __restrict__ W, int Ca, int Cb, int swap) {
    pack_frag_body(P, W, Ca, Cb, swap, blockIdx.x, gridDim.x);
}

// The packs of several layers in ONE launch (pg_bf16x_pack_batch): item i owns the blocks [block0_i, block0_{i+1})
struct PackItem {
    const float* P;
    __bf16* W;
    int Ca, Cb, dir, block0, nblk;
};
struct PackBatch {
    int n;
    PackItem it[PG_BF16X_PACK_MAX];
};
__global__ __launch_bounds__(256) void k_pack_batch(const PackBatch b) {
    __shared__ float tile[32][33];
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.it[i + 1].block0) ++i;
    const PackItem& t = b.it[i];
    const int blk = blockIdx.x - t.block0;
    if (t.dir == 2)
        pack_w8_body(t.P, t.W, t.Ca, t.Cb, blk, t.nblk);
    else if (t.dir >= 4)
        pack_frag_body(t.P, t.W, t.Ca, t.Cb, t.dir == 5, blk, t.nblk);
    else
        pack_w_body(t.P, t.W, t.Ca, t.Cb, t.dir != 0, blk, t.nblk, tile);
}

}  // namespace

bool pg_bf16x_geom_ok(int dir, int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride) {
    (void)stride;
    if (dir == 2) {                    // big has <= 8 channels in 8-channel pixels
        if (Cb > 8 || Ca % 8 != 0 || Ca < 32) return false;
        return (long)N * Hs * Ws * Ca < 0x7fffffffL && (long)N * Hb * Wb * 16 < 0x60000000L;
    }
    if (dir == 3) {                    // rows of `small` (N*Hs*Ws x Ca) times W[Cb][Ca]
        if (Ca % BK != 0 || Cb % 8 != 0) return false;
        return (long)N * Hs * Ws * Cb < 0x7fffffffL && (long)Ca * Cb * 2 < 0x40000000L;
    }
    const int Cin = dir == 0 ? Cb : Ca, Cout = dir == 0 ? Ca : Cb;
    if (Cin % BK != 0 || Cout % 8 != 0 || Cout < 32) return false;
    if (16L * Ca * Cb * 2 >= 0x40000000L) return false;
    const long pix = (long)N * (dir == 0 ? Hs * Ws : Hb * Wb);
    if (pix * Cout >= 0x7fffffffL) return false;
    return true;
}

pg_bf16x_plan pg_bf16x_plan_of(int dir, int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride, int ring) {
    pg_bf16x_plan p;
    p.win = 0;
    const int Cin = (dir == 0) ? Cb : Ca, Cout = (dir == 0 || dir == 2) ? Ca : Cb;
    p.ncls = (dir == 1 && stride == 2) ? 4 : 1;
    const long Mc = (dir != 1) ? (long)N * Hs * Ws : (p.ncls == 4 ? (long)N * ((Hb + 1) / 2) * ((Wb + 1) / 2) : (long)N * Hb * Wb);
    const int taps = (p.ncls == 4) ? 4 : 16;
    p.nchunks = (dir == 2) ? 2 : (dir == 3) ? Cin / BK : taps * Cin / BK;
    p.out_elems = (long)N * (dir != 1 ? Hs * Ws : Hb * Wb) * Cout;
    auto blocks = [&](int bm, int bn) { return ((Mc + bm - 1) / bm) * ((Cout + bn - 1) / bn) * p.ncls; };
    static const int forced = pg_exp_env("PATCHGAN_BF16X_TILE") ? atoi(pg_exp_env("PATCHGAN_BF16X_TILE")) : -1;
    static const int target = pg_exp_env("PATCHGAN_BF16X_TARGET") ? atoi(pg_exp_env("PATCHGAN_BF16X_TARGET")) : 512;
    // Measured on the cfg4 layers (tools/layer_bench_bf16.py, one device): the 256 x 128 tile wins only where it alone fills the chip
    // twice over (>= ~480 workgroups = two per CU) on a long K; one workgroup per CU loses to 128 x 128 tiles at two to four per CU,
    // and a split-K pass (slab write + reduce) costs more than it returns once ~400 workgroups exist without it.
    // Round 4: at four workgroups per CU (OCC above) the 128 x 128 tile is as fast or faster everywhere the 256 x 128 one used to
    // win (dec4 88 -> 79 us, d2 at 2N 87 -> 79, d3 at 2N 275 -> 265 forward / 265 -> 272 data gradient): the wide tile stays reachable
    // through PATCHGAN_BF16X_TILE=0 only.
    // The 256 x 128 tile on EIGHT waves (tile 3, PATCHGAN_BF16X_TILE=3: 64 x 64 per wave like the 128 x 128 tile, two workgroups per CU,
    // 0.375 DMA pieces per MFMA instead of 0.5) wins the back-to-back layer benchmark wherever it fills the chip without a K split
    // (d3 at 2N 258 -> 239 us, dec4 77 -> 71) but LOSES inside the training step, where its operands come from HBM rather than from
    // a warm L2 (cfg4 bf16 step 6.28 -> 6.50 ms with it on every such layer): not selected.
    (void)blocks;
    if (forced >= 0 && forced <= 3 && !(forced == 3 && (Cout <= 64 || ring > 0))) p.tile = forced;
    else if (Cout <= 64) p.tile = 2;
    else p.tile = 1;
    p.bm = (p.tile == 1) ? 128 : 256;
    p.bn = (p.tile == 2) ? 64 : 128;
    p.tiles_m = (int)((Mc + p.bm - 1) / p.bm);
    p.tiles_n = (Cout + p.bn - 1) / p.bn;
    const long nb = (long)p.tiles_m * p.tiles_n * p.ncls;
    long s = (nb >= target * 25 / 32) ? 1 : (target + nb - 1) / nb;      // (400 of 512)
    const long smax = std::max<long>(1, p.nchunks / 4);
    if (s > smax) s = smax;
    p.split = (int)s;
    p.cps = (p.nchunks + p.split - 1) / p.split;
    p.split = (p.nchunks + p.cps - 1) / p.cps;
    p.ring = (ring >= 0 && dir < 2) ? (ring ? 1 : 0) : 0;
    // window-staged kernel (k_conv_bf16r): stride-2 layers whose (class) maps tile into whole R x 16 rectangles
    p.win = 0;
    static const bool nowin = pg_exp_env("PATCHGAN_BF16X_NOWIN") != nullptr;
    if (!nowin && dir <= 1 && stride == 2 && p.ring == 0 && ring <= 0 && (p.tile == 1 || p.tile == 2) && Cin % 64 == 0 &&
        !((Hb | Wb) & 1) && Hb == 2 * Hs && Wb == 2 * Ws) {
        const int R = p.bm / 16;
        if (Hs % R == 0 && Ws % 16 == 0) p.win = 1;
    }
    if (p.win) {            // its K loop runs in units of four taps: whole macro chunks per slice
        const int nm = p.nchunks / 4, cpm = (nm + p.split - 1) / p.split;
        p.cps = 4 * cpm;
        p.split = (nm + cpm - 1) / cpm;
    }
    return p;
}

void pg_bf16x_clamp(pg_bf16x_plan* p, size_t avail) {
    const long smax = (long)(avail / (sizeof(float) * (size_t)p->out_elems));
    if (p->split > 1 && smax < p->split) p->split = smax < 2 ? 1 : (int)smax;
    if (p->win) {
        const int nm = p->nchunks / 4, cpm = (nm + p->split - 1) / p->split;
        p->cps = 4 * cpm;
        p->split = (nm + cpm - 1) / cpm;
        return;
    }
    p->cps = (p->nchunks + p->split - 1) / p->split;
    p->split = (p->nchunks + p->cps - 1) / p->cps;
}

// chunks per sample of the statistics the STATS epilogue emits for this plan; 0: not available (split K, tiles straddling samples)
int pg_bf16x_stats_chunks(int dir, const pg_bf16x_plan* p, int N, int Hb, int Wb, int Hs, int Ws) {
    (void)N;
    if (dir > 1 || p->split != 1) return 0;
    if (dir == 1 && p->ncls == 4 && ((Hb | Wb) & 1)) return 0;   // the four parity classes must have one size
    const long hw = (dir == 0) ? (long)Hs * Ws : (p->ncls == 4 ? (long)(Hb / 2) * (Wb / 2) : (long)Hb * Wb);
    if (hw % p->bm != 0) return 0;
    return (int)(p->ncls * (hw / p->bm));
}

const char* pg_bf16x_kernel_name(int dir, int tile, int ring) {
    static const char* const tiles[4] = {"4,2,2,2", "2,2,2,2", "2,2,4,1", "2,2,4,2"};
    static thread_local char buf[64];
    if (ring == 2) snprintf(buf, sizeof buf, "k_conv_bf16r<%s,%d>", tile == 1 ? "4,1,1,4" : "4,1,2,2", dir);      // window-staged
    else snprintf(buf, sizeof buf, "k_conv_bf16x<%s,%d,%d>", tiles[tile < 0 || tile > 3 ? 0 : tile], dir, (ring && dir < 2) ? 32 : 64);
    return buf;
}

size_t pg_bf16x_w_bytes(int Ca, int Cb) {       // (channel counts padded to 32: the fragment-ordered pack holds whole 32-channel tiles)
    if (Cb <= 8) return ((size_t)16 * Ca * 8 * 2 + 255) & ~(size_t)255;      // 8-channel-pixel form: no fragment pack
    return ((size_t)16 * ((Ca + 31) / 32 * 32) * ((Cb + 31) / 32 * 32) * 2 + 255) & ~(size_t)255;
}

// workgroups of one layer's pack (0: this layout / channel count has no pack)
static int pack_blocks(int Ca, int Cb, int dir) {
    if (dir == 4 || dir == 5) {        // fragment-ordered (k_conv_bf16r): K = Cb (4) / Ca (5) in whole 64-chunks
        const int Cn = dir == 5 ? Cb : Ca, Ck = dir == 5 ? Ca : Cb;
        if (Ck % 64) return 0;
        return (int)std::min<long>((16L * ((Cn + 31) / 32) * (Ck / 64) * 4 * 64 + 255) / 256, 4096);
    }
    if (dir == 2) return (int)std::min<long>((16L * Ca + 255) / 256, 2048);        // W8[a][tap][8]: the Cb <= 8 channels of each tap, zero padded
    if (dir == 0) return (Cb & 3) ? 0 : (int)std::min<long>((4L * Ca * Cb + 255) / 256, 2048);
    return (int)std::min<long>(16L * ((Ca + 31) / 32) * ((Cb + 31) / 32), 4096);    // dir 1 and 3: each tap transposed to [b][a]
}

int pg_bf16x_pack(const float* P, void* W, int Ca, int Cb, int dir, hipStream_t st) {
    const int blocks = pack_blocks(Ca, Cb, dir);
    if (blocks == 0) return PG_EINVAL;
    if (dir == 4 || dir == 5)
        hipLaunchKernelGGL(k_pack_frag_bf16, dim3(blocks), dim3(256), 0, st, P, (__bf16*)W, Ca, Cb, dir == 5 ? 1 : 0);
    else if (dir == 2)
        hipLaunchKernelGGL(k_pack_w8_bf16, dim3(blocks), dim3(256), 0, st, P, (__bf16*)W, Ca, Cb);
    else
        hipLaunchKernelGGL(k_pack_w_bf16, dim3(blocks), dim3(256), 0, st, P, (__bf16*)W, Ca, Cb, dir == 0 ? 0 : 1);
    return pg_launch_status();
}

int pg_bf16x_pack_batch(int n, const pg_bf16x_pack_item* items, hipStream_t st) {
    if (n <= 0) return PG_OK;
    if (n > PG_BF16X_PACK_MAX || !items) return PG_EINVAL;
    PackBatch b;
    b.n = n;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const pg_bf16x_pack_item& s = items[i];
        const int nb = (s.P && s.W && s.Ca > 0 && s.Cb > 0 && s.dir >= 0 && s.dir <= 5) ? pack_blocks(s.Ca, s.Cb, s.dir) : 0;
        if (nb == 0) return PG_EINVAL;
        b.it[i] = PackItem{s.P, (__bf16*)s.W, s.Ca, s.Cb, s.dir, (int)blocks, nb};
        blocks += nb;
    }
    hipLaunchKernelGGL(k_pack_batch, dim3((unsigned)blocks), dim3(256), 0, st, b);
    return pg_launch_status();
}

int pg_bf16x_conv(int dir, const void* in, int ld_in, long in_bytes, const void* W, void* out, int ld_out, long slab_stride,
                  int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride, const pg_bf16x_plan* p, const float* bias,
                  int act, int out_bf, hipStream_t st, pg_epi_mul mul, double* part, int chunks, int bt) {
    if (part && (mul.t || dir > 1 || p->split != 1 || !out_bf || slab_stride != 0)) return PG_EINVAL;
    const XGeom g{N, Hb, Wb, Hs, Ws, Ca, Cb, stride};
    static const bool clsz = pg_exp_env("PATCHGAN_BF16X_CLSZ") != nullptr;      // A/B: the parity classes as grid z (one class after the other)
    const int cls_in_x = (p->ncls == 4 && !clsz) ? 1 : 0;
    const dim3 grid((unsigned)(p->tiles_m * p->tiles_n * (cls_in_x ? 4 : 1)), 1, (unsigned)(cls_in_x ? p->split : p->ncls * p->split));
    const int w_bytes = (dir == 2) ? 16 * Ca * 8 * 2 : (dir == 3) ? Ca * Cb * 2 : 16 * Ca * Cb * 2;
    const __bf16* I = (const __bf16*)in;
    const __bf16* Wp = (const __bf16*)W;
#define PG_BF16X_ARGS I, ld_in, Wp, out, ld_out, slab_stride, g, p->cps, bias, act, (int)in_bytes, w_bytes, out_bf, p->tiles_n, mul, part, chunks, cls_in_x
    if (p->win) {
        if ((dir == 1 && bt) || dir > 1 || (p->tile != 1 && p->tile != 2)) return PG_EINVAL;      // (weights: the fragment-ordered pack, dir 4 / 5)
        const int w_bytes = 16 * (((dir == 0 ? Ca : Cb) + 31) / 32 * 32) * (dir == 0 ? Cb : Ca) * 2;
#define PG_BF16W_K(MR, NR, WM, WN, D, MUL, ST) \
    hipLaunchKernelGGL((k_conv_bf16r<MR, NR, WM, WN, D, MUL, ST>), grid, dim3(256), 0, st, PG_BF16X_ARGS)
#define PG_BF16W_LAUNCH(MR, NR, WM, WN)                                        \
    do {                                                                       \
        if (dir == 0 && part) PG_BF16W_K(MR, NR, WM, WN, 0, false, true);      \
        else if (dir == 0 && !mul.t) PG_BF16W_K(MR, NR, WM, WN, 0, false, false); \
        else if (dir == 0) return PG_EINVAL;                                   \
        else if (part) PG_BF16W_K(MR, NR, WM, WN, 1, false, true);             \
        else if (mul.t) PG_BF16W_K(MR, NR, WM, WN, 1, true, false);            \
        else PG_BF16W_K(MR, NR, WM, WN, 1, false, false);                      \
    } while (0)
        // wave tiling: every wave owns ALL pixel rows of its 32 output channels (128 x 32 / 128 x 32 of a 256 x 64 tile), so that the weight
        // fragments, which each wave loads for itself, are not loaded twice per workgroup (64 x 64 per wave: the fragment loads alone
        // would fill the 64 B / clk path into the registers for as long as the MFMAs run)
        if (p->tile == 1) PG_BF16W_LAUNCH(4, 1, 1, 4);
        else PG_BF16W_LAUNCH(4, 1, 2, 2);
#undef PG_BF16W_LAUNCH
#undef PG_BF16W_K
        return pg_launch_status();
    }
    // (the 64-accumulator tiles at four workgroups per CU: the two-per-CU A/B of round 4, 6.94 vs 6.35 ms per cfg4 step, is in EXPERIMENTS.md)
#define PG_BF16X_K(MR, NR, WM, WN, D, KB, MUL, ST, BT) \
    hipLaunchKernelGGL((k_conv_bf16x<MR, NR, WM, WN, D, KB, MUL, ST, BT>), grid, dim3(WM * WN * 64), 0, st, PG_BF16X_ARGS)
#define PG_BF16X_LAUNCH(MR, NR, WM, WN)                                                                                      \
    do {                                                                                                                     \
        if (dir == 1 && !p->ring && bt) {                                                                                    \
            if (part) PG_BF16X_K(MR, NR, WM, WN, 1, 64, false, true, true);                                                  \
            else if (mul.t) PG_BF16X_K(MR, NR, WM, WN, 1, 64, true, false, true);                                            \
            else PG_BF16X_K(MR, NR, WM, WN, 1, 64, false, false, true);                                                      \
        } else if (part && dir == 0) PG_BF16X_K(MR, NR, WM, WN, 0, 64, false, true, false);                                  \
        else if (part) PG_BF16X_K(MR, NR, WM, WN, 1, 64, false, true, false);                                                \
        else if (dir == 0 && !p->ring) PG_BF16X_K(MR, NR, WM, WN, 0, 64, false, false, false);                               \
        else if (dir == 0) PG_BF16X_K(MR, NR, WM, WN, 0, 32, false, false, false);                                           \
        else if (dir == 1 && mul.t) PG_BF16X_K(MR, NR, WM, WN, 1, 64, true, false, false);                                   \
        else if (dir == 1 && !p->ring) PG_BF16X_K(MR, NR, WM, WN, 1, 64, false, false, false);                               \
        else if (dir == 1) PG_BF16X_K(MR, NR, WM, WN, 1, 32, false, false, false);                                           \
        else if (dir == 2) PG_BF16X_K(MR, NR, WM, WN, 2, 64, false, false, false);                                           \
        else PG_BF16X_K(MR, NR, WM, WN, 3, 64, false, false, false);                                                         \
    } while (0)
    switch (p->tile) {
        case 0: PG_BF16X_LAUNCH(4, 2, 2, 2); break;
        case 1: PG_BF16X_LAUNCH(2, 2, 2, 2); break;
        case 3: PG_BF16X_LAUNCH(2, 2, 4, 2); break;        // 256 x 128 on eight waves
        default: PG_BF16X_LAUNCH(2, 2, 4, 1); break;
    }
#undef PG_BF16X_LAUNCH
#undef PG_BF16X_K
#undef PG_BF16X_ARGS
    return pg_launch_status();
}

// ---- weight gradient -------------------------------------------------------------------------------------------------------
bool pg_bf16x_wgrad_geom_ok(int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride) {
    (void)Hb; (void)Wb; (void)stride;
    if ((long)N * Hs * Ws >= (1L << 24)) return false;             // pixel decode by float reciprocals
    if (Cb <= 8) return Ca % 32 == 0;                              // taps in N: `big` in 8-channel pixels (ld_big == 8, checked by the caller)
    if (Ca % 32 != 0 || Cb % 32 != 0 || Ca < 64 || Cb < 32) return false;
    return true;
}

pg_bf16x_plan pg_bf16x_wgrad_plan(int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride) {
    (void)Hb; (void)Wb; (void)stride;
    pg_bf16x_plan p;
    p.win = 0;
    static const int forced = pg_exp_env("PATCHGAN_BF16X_WTILE") ? atoi(pg_exp_env("PATCHGAN_BF16X_WTILE")) : -1;
    // tiles (a x b): 0: 256 x 128, 1: 128 x 128, 2: 256 x 64, 3: 128 x 64
    if (Cb <= 8) {                     // taps in N: tiles 4 / 5 = 64 / 128 a-channels x (16 taps x 8)
        p.tile = (Ca % 128 == 0) ? 5 : 4;
        p.bm = (p.tile == 5) ? 128 : 64;
        p.bn = 128;
        p.tiles_m = (Ca + p.bm - 1) / p.bm;
        p.tiles_n = 1;
        p.ncls = 1;
        const long Mp = (long)N * Hs * Ws;
        p.nchunks = (int)((Mp + 63) / 64);
        p.out_elems = 16L * Ca * Cb;
        long sp = std::max<long>(1, 1024 / p.tiles_m);             // HBM-bound streaming of every pixel: many short slices
        const long spmax = std::max<long>(1, p.nchunks / 8);
        if (sp > spmax) sp = spmax;
        p.split = (int)sp;
        p.cps = (p.nchunks + p.split - 1) / p.split;
        p.split = (p.nchunks + p.cps - 1) / p.cps;
        p.ring = 0;
        return p;
    }
    // (measured, tools/layer_bench_bf16.py: with <= 4096 pixels the 128 x 128 tile's shorter slices win: 20 vs 44 us on 1024 x 512 at 16 x 16)
    const long Mpix = (long)N * Hs * Ws;
    if (forced >= 0 && forced <= 3) p.tile = forced;
    else if (Cb % 128 == 0) p.tile = (Ca % 256 == 0 && Mpix > 4096) ? 0 : 1;
    else p.tile = (Ca % 256 == 0) ? 2 : 3;
    p.bm = (p.tile == 0 || p.tile == 2) ? 256 : 128;
    p.bn = (p.tile <= 1) ? 128 : 64;
    p.tiles_m = (Ca + p.bm - 1) / p.bm;
    p.tiles_n = (Cb + p.bn - 1) / p.bn;
    p.ncls = 16;
    const long M = (long)N * Hs * Ws;
    p.nchunks = (int)((M + 63) / 64);
    p.out_elems = 16L * Ca * Cb;
    const long nb = (long)p.tiles_m * p.tiles_n * 16;
    // workgroups to aim for: what the chip holds at once -- the 64-channel-wide tiles (2, 3) run four workgroups per CU.  Measured in the
    // cfg4 step (tools/step_launch_table.py): 128 -> 64 channels at 256 x 256, 2N: 158 -> 116 us with 64 slices instead of 32; the
    // 128 x 128 tile (also four per CU) gains nothing from it (23 -> 28 us on 1024 x 512 at 16 x 16), the 256 x 128 one (two per CU) loses
    static const int wtarget = pg_exp_env("PATCHGAN_BF16X_WTARGET") ? atoi(pg_exp_env("PATCHGAN_BF16X_WTARGET")) : 0;
    const int target = wtarget > 0 ? wtarget : (p.tile >= 2 ? 1024 : 512);
    long s = (nb >= target) ? 1 : (target + nb - 1) / nb;
    const long smax = std::max<long>(1, p.nchunks / 4);
    if (s > smax) s = smax;
    p.split = (int)s;
    p.cps = (p.nchunks + p.split - 1) / p.split;
    p.split = (p.nchunks + p.cps - 1) / p.cps;
    p.ring = 0;
    return p;
}

const char* pg_bf16x_wgrad_kernel_name(int tile) {
    static const char* const names[6] = {"k_wgrad_bf16x<4,2,2,2>", "k_wgrad_bf16x<2,2,2,2>", "k_wgrad_bf16x<2,2,4,1>", "k_wgrad_bf16x<1,2,4,1>",
                                         "k_wgrad_bf16x<2,1,1,4,true>", "k_wgrad_bf16x<4,1,1,4,true>"};
    return names[tile < 0 || tile > 5 ? 0 : tile];
}

int pg_bf16x_wgrad(const void* small, int ld_small, long small_bytes, const void* big, int ld_big, long big_bytes, float* out,
                   long slab_stride, int N, int Hb, int Wb, int Hs, int Ws, int Ca, int Cb, int stride, const pg_bf16x_plan* p,
                   hipStream_t st) {
    const XGeom g{N, Hb, Wb, Hs, Ws, Ca, Cb, stride};
    const int ntiles = p->tiles_m * p->tiles_n;
    const dim3 grid((unsigned)(ntiles * p->ncls * p->split), 1, 1);
    const float inv_hw = 1.0f / (float)(Hs * Ws), inv_w = 1.0f / (float)Ws;
    const __bf16* S = (const __bf16*)small;
    const __bf16* B = (const __bf16*)big;
#define PG_BF16X_WG(MR, NR, WM, WN, TN)                                                                                      \
    hipLaunchKernelGGL((k_wgrad_bf16x<MR, NR, WM, WN, TN>), grid, dim3(256), 0, st, S, ld_small, B, ld_big, out, slab_stride, g, p->cps, \
                       (int)small_bytes, (int)big_bytes, p->tiles_n, ntiles, inv_hw, inv_w)
    switch (p->tile) {
        case 0: PG_BF16X_WG(4, 2, 2, 2, false); break;
        case 1: PG_BF16X_WG(2, 2, 2, 2, false); break;
        case 2: PG_BF16X_WG(2, 2, 4, 1, false); break;
        case 3: PG_BF16X_WG(1, 2, 4, 1, false); break;
        case 4: PG_BF16X_WG(2, 1, 1, 4, true); break;
        default: PG_BF16X_WG(4, 1, 1, 4, true); break;
    }
#undef PG_BF16X_WG
    return pg_launch_status();
}

#ifdef PG_TRACE_R
extern "C" int pg_debug_trace_set(void* buf) {
    unsigned long long* p = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(pg_trace_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
