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
int main() {
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
