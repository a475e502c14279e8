// Probe of ds_read_b64_tr_b16 (gfx950): which (row, column) each lane receives.  LDS image: rows of PITCH 16-bit elements,
// element value = row * 256 + col.  Each lane passes the address of (row = lane_row(lane), col = lane_col(lane)) as the ISA
// describes (per 16-lane group: lane 4q+p -> row q, columns 4p..4p+3) and prints what it got.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 72;
__global__ void k(short* out) {
    __shared__ short sm[64 * PITCH];
    for (int i = threadIdx.x; i < 64 * PITCH; i += 64) sm[i] = (short)((i / PITCH) * 256 + (i % PITCH));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // group g reads the 4-row x 16-column block at rows 4g.., columns 0..15
    const short* addr = &sm[(4 * g + q) * PITCH + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
// the fragment formula of k_wgrad_bf16: LDS tile [k][m] of pitch LDA; lane l must receive (k = 8 * (l / 32) + j, m = l % 32)
constexpr int LDA = 136;
__global__ void k2(short* out) {
    __shared__ short As[32 * LDA];
    for (int i = threadIdx.x; i < 32 * LDA; i += 64) As[i] = (short)((i / LDA) * 256 + (i % LDA));
    __syncthreads();
    const int lane = threadIdx.x;
    const int tr_row = ((lane >> 4) >> 1) * 8 + ((lane & 15) >> 2), tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const short* p = &As[(0 * 16 + tr_row) * LDA + 0 * 32 + tr_col];
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * LDA));
    for (int e = 0; e < 4; ++e) {
        out[lane * 8 + e] = lo[e];
        out[lane * 8 + 4 + e] = hi[e];
    }
}
int main() {
    {
        short* d2;
        hipMalloc(&d2, 64 * 8 * 2);
        hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d2);
        short h2[512];
        hipMemcpy(h2, d2, sizeof h2, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * (l / 32) + j, m = l % 32;
                if (h2[l * 8 + j] != (short)(k * 256 + m)) {
                    if (bad < 12) printf("lane %d j %d: got (k%d,m%d) want (k%d,m%d)\n", l, j, h2[l * 8 + j] / 256, h2[l * 8 + j] % 256, k, m);
                    ++bad;
                }
            }
        printf("fragment formula mismatches: %d\n", bad);
    }
    short* d;
    hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / 256, h[l * 4 + e] % 256);
        printf("\n");
    }
    return 0;
}
