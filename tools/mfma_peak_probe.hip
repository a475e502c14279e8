// What the bf16 MFMA pipe sustains with nothing else going on: every CU, W waves per SIMD, four independent accumulator chains of
// v_mfma_f32_32x32x16_bf16 per wave, for a few hundred microseconds (the clock under this load is what prices every bf16 kernel).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak_probe.hip -o tools/mfma_peak_probe && tools/mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same with the 16x16 shapes (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32): equal FLOP per cycle on paper (16 / 32 cycles per
// instruction at half the FLOP); MI355X_MICROARCH.md "DVFS give-back" (7) reports the chip holding a HIGHER clock under the 16x16x32 loop
__device__ inline unsigned pk_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
template <bool BF>
__global__ __launch_bounds__(256) void k_peak16(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {      // random operands of magnitude ~1 (the clock depends on the data)
        a[e] = (__bf16)((float)(pk_hash(threadIdx.x * 16 + e) & 0xffff) / 32768.f - 1.f);
        b[e] = (__bf16)((float)(pk_hash(threadIdx.x * 16 + e + 8) & 0xffff) / 32768.f - 1.f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (BF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, a)[0], __builtin_bit_cast(f32x4, b)[0], acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <bool BF>
__global__ __launch_bounds__(256) void k_peak(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {      // random operands of magnitude ~1 (the clock depends on the data)
        a[e] = (__bf16)((float)(pk_hash(threadIdx.x * 16 + e) & 0xffff) / 32768.f - 1.f);
        b[e] = (__bf16)((float)(pk_hash(threadIdx.x * 16 + e + 8) & 0xffff) / 32768.f - 1.f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, a)[0], __builtin_bit_cast(f32x4, b)[0], acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
int main() {
    float* d;
    (void)hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
    for (int bf = 1; bf >= 0; --bf)
        for (int wps : {1, 2, 4})
            for (int iters : {20000, 100000}) {
                const int wgs = 256 * wps;      // four waves per workgroup = one per SIMD
                float best = 1e9f, last = 0.f;
                for (int r = 0; r < 4; ++r) {
                    (void)hipEventRecord(e0);
                    // (per iteration and wave: 16 MFMAs of the 32x32 shape = 32 of the 16x16 shape = the same FLOP)
                    if (shape == 0 && bf) hipLaunchKernelGGL(k_peak<true>, dim3(wgs), dim3(256), 0, 0, d, iters);
                    else if (shape == 0) hipLaunchKernelGGL(k_peak<false>, dim3(wgs), dim3(256), 0, 0, d, iters / 4);
                    else if (bf) hipLaunchKernelGGL(k_peak16<true>, dim3(wgs), dim3(256), 0, 0, d, iters);
                    else hipLaunchKernelGGL(k_peak16<false>, dim3(wgs), dim3(256), 0, 0, d, iters / 4);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                    last = ms;
                }
                const double flop = (double)wgs * 4 * (bf ? iters : iters / 4) * 16.0 * (bf ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2);
                printf("%s waves/SIMD %d iters %6d: best %8.3f ms %7.1f TFLOP/s   last %8.3f ms %7.1f TFLOP/s\n",
                       shape ? (bf ? "bf16 16x16x32" : "f32  16x16x4 ") : (bf ? "bf16 32x32x16" : "f32  32x32x2 "), wps,
                       bf ? iters : iters / 4, best, flop / best / 1e9, last, flop / last / 1e9);
            }
    return 0;
}
