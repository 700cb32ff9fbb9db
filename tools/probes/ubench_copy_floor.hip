// Developer tool: the memory-only floor of a single-sweep launch -- the sweep kernel's loop (16-byte buffer loads one
// tile ahead, delayed write-through stores, grid-stride tiles of 256 pairs) with the arithmetic removed (x -> x + 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32v4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 ld(const double* base)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const u32v4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 0);
    double2 d;
    d.x = __longlong_as_double((long long)(((uint64_t)v.y << 32) | v.x));
    d.y = __longlong_as_double((long long)(((uint64_t)v.w << 32) | v.z));
    return d;
}
__device__ __forceinline__ void st(double* base, double2 d)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const uint64_t a = (uint64_t)__double_as_longlong(d.x), b = (uint64_t)__double_as_longlong(d.y);
    const u32v4_t v = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, threadIdx.x * 16, 0, 16);
}
__global__ __launch_bounds__(256) void touch_kernel(double* x, int64_t n_pairs)
{
    const int64_t stride = (int64_t)gridDim.x * 256, first = (int64_t)blockIdx.x * 256;
    double2 nxt = {0, 0}, done = {0, 0};
    if (first < n_pairs) nxt = ld(x + 2 * first);
    int64_t base_done = -1, base = first;
    for (; base + stride < n_pairs; base += stride) {
        double2 v = nxt;
        nxt = ld(x + 2 * (base + stride));
        if (base_done >= 0) st(x + 2 * base_done, done);
        v.x += 1.0; v.y += 1.0;
        done = v; base_done = base;
    }
    if (base < n_pairs) {
        double2 v = nxt;
        if (base_done >= 0) st(x + 2 * base_done, done);
        v.x += 1.0; v.y += 1.0;
        if (base + threadIdx.x < n_pairs) st(x + 2 * base, v);
    }
}
int main()
{
    const int64_t M = 10000000, n_pairs = M / 2;
    double* d; (void)hipMalloc(&d, (size_t)(M + 1024) * 8); (void)hipMemset(d, 0, (size_t)(M + 1024) * 8);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc : {4, 6, 8, 16}) {
        const int grid = 256 * bpc;
        for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, s, d, n_pairs);
        hipStreamSynchronize(s);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, s, d, n_pairs);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%2d blocks/CU: %.2f us per launch (160 MB -> %.2f TB/s)\n", bpc, best * 1e3f / 3000, 160e6 / (best * 1e-3 / 3000) * 1e-12);
    }
    return 0;
}
