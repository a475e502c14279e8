// Fills the LDS of every CU with NaN patterns for a while (a second process next to a test run): a kernel that reads LDS it never wrote
// then sees NaNs instead of its own predecessor's leftovers.  usage: lds_polluter [seconds]   (hipcc --offload-arch=gfx950 -O2)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_pollute(int words, unsigned pattern) {
    extern __shared__ unsigned s[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) s[i] = pattern ^ (unsigned)(i & 0xff);
    __syncthreads();
    if (s[(threadIdx.x * 7) % words] == 0x12345678u) printf("x");     // keeps the stores alive
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 30.0;
    const int bytes = 64 * 1024;       // per workgroup: two to three of them resident per CU sweep the whole 160 KB over time
    hipFuncSetAttribute((const void*)k_pollute, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int i = 0; i < 64; ++i) {
            hipLaunchKernelGGL(k_pollute, dim3(1024), dim3(256), (i & 1) ? 160 * 1024 : bytes, 0, ((i & 1) ? 160 * 1024 : bytes) / 4,
                               0x7fc00000u);      // quiet NaN (fp32); as bf16 pairs 0x7fc0 is a NaN too
            ++n;
        }
        hipDeviceSynchronize();
    }
    printf("lds_polluter: %ld launches\n", n);
    return 0;
}
