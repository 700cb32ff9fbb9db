// Test helper (GPU box): Philox4x32-10 words from the REAL rocRAND device API, for cross-checking the engine's own
// Philox (amc_math.h) and the oracle's.  rocrand_init(seed, subsequence, offset) places the 128-bit counter at
// (offset / 4 [64 bit], subsequence [64 bit]); the engine's draw counter is (x, y, z, w) = (step lo,
// step hi | draw << 16 | stream << 28, pair lo, pair hi), so subsequence = pair and offset = 4 * (x | y << 32).
// Reads lines "seed pair counter_xy" from stdin, prints the four words of rocrand4() per line.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>

#include <cstdio>
#include <vector>

__global__ void words_kernel(const unsigned long long* seed, const unsigned long long* pair,
                             const unsigned long long* xy, uint4* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed[i], pair[i], 0ull, &st);
    // offset = 4 * xy can exceed 64 bits: advance the counter in two steps of whole Philox blocks
    skipahead(2ull * xy[i], &st);
    skipahead(2ull * xy[i], &st);
    out[i] = rocrand4(&st);
}

int main()
{
    std::vector<unsigned long long> seed, pair, xy;
    unsigned long long a, b, c;
    while (std::scanf("%llu %llu %llu", &a, &b, &c) == 3) { seed.push_back(a); pair.push_back(b); xy.push_back(c); }
    const int n = (int)seed.size();
    if (n == 0) return 0;
    unsigned long long *ds, *dp, *dx;
    uint4* dout;
    hipMalloc(&ds, n * 8); hipMalloc(&dp, n * 8); hipMalloc(&dx, n * 8); hipMalloc(&dout, n * sizeof(uint4));
    hipMemcpy(ds, seed.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(dp, pair.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(dx, xy.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(words_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, ds, dp, dx, dout, n);
    std::vector<uint4> out(n);
    if (hipMemcpy(out.data(), dout, n * sizeof(uint4), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (int i = 0; i < n; ++i) std::printf("%u %u %u %u\n", out[i].x, out[i].y, out[i].z, out[i].w);
    return 0;
}
