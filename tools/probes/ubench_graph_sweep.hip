// Developer tool: single-sweep launches of libamc at M = 1e7, submitted one by one against replayed from a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "amc.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define AK(x) do { if ((x) != 0) { printf("%s: %s\n", #x, amc_last_error()); return 1; } } while (0)
int main()
{
    double sigma[1] = {0.1}, weight[1] = {1.0};
    amc_config cfg = {sizeof(amc_config), 0, 10000000, 0, 10000000, AMC_POTENTIAL_HARMONIC, 1, 2.0, sigma, weight, 1, 1, 0, NULL, 0, 0};
    amc_handle* h; AK(amc_create(&cfg, &h));
    AK(amc_init_uniform(h, -2.0, 2.0));
    void* sp = nullptr; AK(amc_get_stream(h, &sp));
    hipStream_t s = (hipStream_t)sp;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20000; ++i) AK(amc_sweep(h, 1));      // clocks
    CK(hipStreamSynchronize(s));
    const int N = 100, R = 20;
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N * R; ++i) AK(amc_sweep(h, 1));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream: %.2f us per sweep\n", ms * 1e3f / (N * R));
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) AK(amc_sweep(h, 1));
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph of %d: %.2f us per sweep\n", N, ms * 1e3f / (N * R));
    }
    amc_destroy(h);
    return 0;
}
