// Developer tool: per-node cost of a linear hipGraph of kernel nodes against the same kernels launched on a stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void empty_kernel(double* p, int i) { if (p && threadIdx.x == 9999) p[0] = i; }
__global__ void touch_kernel(double* p, int i) { p[(size_t)blockIdx.x * 256 + threadIdx.x] += 1.0; }
int main()
{
    double* d; CK(hipMalloc(&d, (size_t)4096 * 256 * 8)); CK(hipMemset(d, 0, (size_t)4096 * 256 * 8));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 200, R = 25;
    for (int grid : {1, 1536}) {
        for (int which = 0; which < 2; ++which) {
            auto launch = [&](int i) { if (which) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, s, d, i); else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, d, i); };
            for (int i = 0; i < 2000; ++i) launch(i);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N * R; ++i) launch(i);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const float stream_us = ms * 1e3f / (N * R);
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) launch(i);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("grid %5d x 256  %-22s: stream %.2f us per launch, graph of %d nodes %.2f us per node\n", grid,
                   which ? "load+store 8 B/thread" : "empty", stream_us, N, ms * 1e3f / (N * R));
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
