// Developer tool: back-to-back launch cost of kernels that do (almost) nothing, for the fixed part of a sweep launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(double* p) { if (p && threadIdx.x == 9999) p[0] = 1.0; }
__global__ void touch_kernel(double* p) { p[(size_t)blockIdx.x * 256 + threadIdx.x] += 1.0; }
int main()
{
    double* d; hipMalloc(&d, (size_t)4096 * 256 * 8); hipMemset(d, 0, (size_t)4096 * 256 * 8);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1, 256, 1536, 2048}) {
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < 2000; ++i) { if (which) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, s, d); else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, d); }
            hipStreamSynchronize(s);
            hipEventRecord(e0, s);
            for (int i = 0; i < 5000; ++i) { if (which) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, s, d); else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, d); }
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("grid %5d x 256  %s: %.2f us per launch\n", grid, which ? "load+store 8 B/thread" : "empty", ms * 1e3 / 5000);
        }
    }
    return 0;
}
