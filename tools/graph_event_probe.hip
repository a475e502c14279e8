// Does this HIP runtime time a kernel INSIDE a captured graph with external event-record nodes (hipEventRecordWithFlags(...,
// hipEventRecordExternal) during stream capture), replay after replay?  bench.py's roofline leg wants HIP events around the dominant
// kernel while the timed steps are hipGraph replays.  hipcc --offload-arch=gfx950 tools/graph_event_probe.hip -o tools/graph_event_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(float* p, int n) {
    float a = p[threadIdx.x];
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    p[threadIdx.x] = a;
}
int main() {
    float* d; CK(hipMalloc(&d, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, d, 1000);
    CK(hipEventRecordWithFlags(e0, st, hipEventRecordExternal));
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, d, 200000);
    CK(hipEventRecordWithFlags(e1, st, hipEventRecordExternal));
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, d, 1000);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) {
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        float ms = -1; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("replay %d: %.3f ms between the external event nodes\n", r, ms);
    }
    // eager reference
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, d, 200000);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = -1; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager: %.3f ms\n", ms);
    return 0;
}
