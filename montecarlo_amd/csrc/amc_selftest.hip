// amc_selftest.hip -- parity-test hooks of the C ABI (amc_selftest_*): the device's arithmetic primitives, Philox words, the accept
// filter's float estimate and the wave totals, evaluated on the device for the tests to compare with the oracle.
#define AMC_KERNEL_LINKAGE static      // this object's own copies of the selftest kernels; none of the others is emitted
#include "amc_internal.h"

extern "C" {

// ---- parity-test hooks ----------------------------------------------------------------
int amc_selftest_math(int device, int fn, const double* a, const double* b_or_null, double* out, int64_t n)
{
    if (!a || !out || n < 0 || fn < 0 || fn > 11 || ((fn == 5 || fn == 6 || fn == 9 || fn == 10 || fn == 11) && !b_or_null))
        return fail(AMC_ERR_BAD_ARG, "amc_selftest_math: bad argument");
    if (n == 0) return AMC_OK;
    AMC_HIP(hipSetDevice(device));
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    AMC_HIP(hipMalloc(&da, (size_t)n * sizeof(double)));
    AMC_HIP(hipMalloc(&db, (size_t)n * sizeof(double)));
    AMC_HIP(hipMalloc(&dout, (size_t)n * sizeof(double)));
    AMC_HIP(hipMemcpy(da, a, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    AMC_HIP(hipMemcpy(db, b_or_null ? b_or_null : a, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(amc::selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn, da, db, dout, n);
    AMC_HIP(hipGetLastError());
    AMC_HIP(hipMemcpy(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return AMC_OK;
}

int amc_selftest_accept_filter(int device, float t_from, float t_to, double* max_rel_err)
{
    if (!max_rel_err || !(t_from <= 0.0f) || !(t_to <= t_from) || !(t_to >= -1e30f))
        return fail(AMC_ERR_BAD_ARG, "amc_selftest_accept_filter: need 0 >= t_from >= t_to (negative floats, from the one nearer zero)");
    AMC_HIP(hipSetDevice(device));
    // negative floats order like their bit patterns: -0.0 = 0x80000000 < ... ; walk from t_from down to t_to
    uint32_t b0, b1;
    float f0 = t_from == 0.0f ? -0.0f : t_from;
    std::memcpy(&b0, &f0, 4);
    std::memcpy(&b1, &t_to, 4);
    const uint64_t count = (uint64_t)b1 - (uint64_t)b0 + 1;
    unsigned long long* d_max = nullptr;
    AMC_HIP(hipMalloc(&d_max, sizeof(unsigned long long)));
    AMC_HIP(hipMemset(d_max, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(amc::selftest_filter_kernel, dim3(4096), dim3(256), 0, 0, b0, count, d_max);
    AMC_HIP(hipGetLastError());
    unsigned long long bits = 0;
    AMC_HIP(hipMemcpy(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
    (void)hipFree(d_max);
    std::memcpy(max_rel_err, &bits, sizeof(double));
    return AMC_OK;
}

int amc_selftest_philox(int device, uint64_t seed, const uint64_t* pair, const uint64_t* t, uint32_t draw,
                        uint32_t stream, uint32_t* out4, int64_t n)
{
    if (!pair || !t || !out4 || n < 0) return fail(AMC_ERR_BAD_ARG, "amc_selftest_philox: bad argument");
    if (n == 0) return AMC_OK;
    AMC_HIP(hipSetDevice(device));
    uint64_t *dp = nullptr, *dt = nullptr;
    uint32_t* dout = nullptr;
    AMC_HIP(hipMalloc(&dp, (size_t)n * sizeof(uint64_t)));
    AMC_HIP(hipMalloc(&dt, (size_t)n * sizeof(uint64_t)));
    AMC_HIP(hipMalloc(&dout, (size_t)n * 4 * sizeof(uint32_t)));
    AMC_HIP(hipMemcpy(dp, pair, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice));
    AMC_HIP(hipMemcpy(dt, t, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(amc::selftest_philox_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t)seed,
                       (uint32_t)(seed >> 32), dp, dt, draw, stream, dout, n);
    AMC_HIP(hipGetLastError());
    AMC_HIP(hipMemcpy(out4, dout, (size_t)n * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    (void)hipFree(dp); (void)hipFree(dt); (void)hipFree(dout);
    return AMC_OK;
}

int amc_selftest_wave_totals(int device, const int64_t* values, int64_t* totals, int64_t* totals_plain)
{
    if (!values || !totals || !totals_plain) return fail(AMC_ERR_BAD_ARG, "amc_selftest_wave_totals: NULL argument");
    AMC_HIP(hipSetDevice(device));
    long long *din = nullptr, *dout = nullptr, *dref = nullptr;
    AMC_HIP(hipMalloc(&din, 6 * 64 * sizeof(long long)));
    AMC_HIP(hipMalloc(&dout, 13 * sizeof(long long)));
    AMC_HIP(hipMalloc(&dref, 6 * sizeof(long long)));
    AMC_HIP(hipMemcpy(din, values, 6 * 64 * sizeof(long long), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(amc::selftest_wave_totals_kernel, dim3(1), dim3(64), 0, 0, din, dout, dref);
    AMC_HIP(hipGetLastError());
    AMC_HIP(hipMemcpy(totals, dout, 13 * sizeof(long long), hipMemcpyDeviceToHost));
    AMC_HIP(hipMemcpy(totals_plain, dref, 6 * sizeof(long long), hipMemcpyDeviceToHost));
    (void)hipFree(din); (void)hipFree(dout); (void)hipFree(dref);
    return AMC_OK;
}

}  // extern "C"
