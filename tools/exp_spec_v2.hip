// Developer experiment (timing only, no parity): how fast would the pair-step be with table-driven
// exp / log / sincospi and a 52-bit Julia-style accept uniform?  Stand-alone: own kernel, runs
// nsteps fused steps over M chains and prints ns per pair-step.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox(u32x4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t m1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = __builtin_amdgcn_bitop3_b32((uint32_t)(m1 >> 32), c.y, k0, 0x96);
        n.y = (uint32_t)m1;
        n.z = __builtin_amdgcn_bitop3_b32((uint32_t)(m0 >> 32), c.w, k1, 0x96);
        n.w = (uint32_t)m0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// tables in LDS: exp2[32], {inv_c, log_c}[128], {sin, cos}[128]
struct Tables { double e2[32]; double invc[128]; double logc[128]; double sn[128]; double cs[128]; };

__device__ __forceinline__ double exp_tab(double x, const Tables* T)
{
    const double SHIFT = 0x1.8p52;
    const double t = __builtin_fma(x, 46.16624130844683, SHIFT);        // 32/ln2
    const double kd = t - SHIFT;
    const uint32_t ki = (uint32_t)__double_as_longlong(t);
    double r = __builtin_fma(-kd, 0x1.62e42fee00000p-6, x);
    r = __builtin_fma(-kd, 0x1.a39ef35793c76p-38, r);
    double p = 1.0 / 720;
    p = __builtin_fma(p, r, 1.0 / 120);
    p = __builtin_fma(p, r, 1.0 / 24);
    p = __builtin_fma(p, r, 1.0 / 6);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double y = T->e2[ki & 31u] * p;
    const int32_t m = (int32_t)ki >> 5;
    return __longlong_as_double(__double_as_longlong(y) + ((long long)m << 52));
}

__device__ __forceinline__ double log_tab(double x, const Tables* T)
{
    const uint64_t ux = (uint64_t)__double_as_longlong(x);
    const uint32_t hx = (uint32_t)(ux >> 32);
    const int32_t k = (int32_t)(hx >> 20) - 1023;
    const uint32_t j = (hx >> 13) & 127u;
    const double m = __longlong_as_double((long long)((ux & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    const double r = __builtin_fma(m, T->invc[j], -1.0);
    double p = 1.0 / 7;
    p = __builtin_fma(p, r, -1.0 / 6);
    p = __builtin_fma(p, r, 0.2);
    p = __builtin_fma(p, r, -0.25);
    p = __builtin_fma(p, r, 1.0 / 3);
    p = __builtin_fma(p, r, -0.5);
    p = __builtin_fma(p, r, 1.0);
    const double dk = (double)k;
    const double hi = __builtin_fma(dk, 0x1.62e42fee00000p-1, T->logc[j]);
    return __builtin_fma(p, r, __builtin_fma(dk, 0x1.a39ef35793c76p-33, hi));
}

__device__ __forceinline__ void sincospi_tab(double w, double& s, double& c, const Tables* T)
{
    const double SHIFT = 0x1.8p52;
    const double t = __builtin_fma(w, 64.0, SHIFT);
    const double nd = t - SHIFT;
    const uint32_t j = (uint32_t)__double_as_longlong(t) & 127u;
    const double r = __builtin_fma(nd, -0.015625, w);
    const double z = r * r;
    double ps = -0x1.32d2cce62bd86p-1;                 // -pi^7/7!
    ps = __builtin_fma(ps, z, 0x1.466bc6775aae2p+1);   //  pi^5/5!
    ps = __builtin_fma(ps, z, -0x1.4abbce625be53p+2);  // -pi^3/3!
    ps = __builtin_fma(ps, z, 0x1.921fb54442d18p+1);   //  pi
    const double sr = ps * r;
    double pc = -0x1.55d3c7e3cbffap+0;                 // -pi^6/6!
    pc = __builtin_fma(pc, z, 0x1.03c1f081b5ac4p+2);   //  pi^4/4!
    pc = __builtin_fma(pc, z, -0x1.3bd3cc9be45dep+2);  // -pi^2/2
    const double cr = __builtin_fma(pc, z, 1.0);
    const double S = T->sn[j], C = T->cs[j];
    s = __builtin_fma(S, cr, C * sr);
    c = __builtin_fma(C, cr, -(S * sr));
}

__device__ __forceinline__ double div_by_const(double a, double b, double y)
{
    const double q0 = a * y;
    const double r0 = __builtin_fma(-q0, b, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-q1, b, a);
    return __builtin_fma(r1, y, q1);
}

__device__ __forceinline__ double uni52(uint32_t lo, uint32_t hi)
{
    const uint64_t v = ((uint64_t)hi << 32 | lo) >> 12;
    return __longlong_as_double((long long)(v | 0x3ff0000000000000ull)) - 1.0;
}

template <bool TAB>
__device__ __forceinline__ bool mh(double& x, double beta, double sigma, double den, double rden, double logc, double z,
                                   double u, const Tables* T)
{
    const double delta = 0.0 + sigma * z;
    const double logq = div_by_const(-(delta * delta), den, rden) - logc;
    const double e1 = x * x;
    const double xn = x + delta;
    const double e2 = xn * xn;
    const double dlogp = ((-e2) * beta) - ((-e1) * beta);
    const double arg = (dlogp + logq) - logq;
    const bool accept = (arg >= 0.0) | ((arg >= -708.0) & (exp_tab(arg, T) > u));
    const double xr = xn + (-delta);
    x = accept ? xn : xr;
    return accept;
}

__global__ __launch_bounds__(256) void sweep(double* x, const Tables* gT, int64_t n_pairs, int nsteps, uint64_t t0,
                                             double sigma, double beta, unsigned long long* slots)
{
    __shared__ Tables T;
    {
        const double* src = reinterpret_cast<const double*>(gT);
        double* dst = reinterpret_cast<double*>(&T);
        for (int i = threadIdx.x; i < (int)(sizeof(Tables) / 8); i += 256) dst[i] = src[i];
        __syncthreads();
    }
    const double den = 2.0 * (sigma * sigma), rden = 1.0 / den, logc = 0.5 * log(6.283185307179586 * sigma * sigma);
    const int64_t stride = (int64_t)gridDim.x * 256;
    unsigned long long wacc = 0;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < n_pairs; base += stride) {
        const int64_t p = base + threadIdx.x;
        const bool v = p < n_pairs;
        const int64_t pc = v ? p : 0;
        double2 xv = *reinterpret_cast<const double2*>(x + 2 * pc);
        for (int s = 0; s < nsteps; ++s) {
            const uint64_t t = t0 + s;
            u32x4 c0 = {(uint32_t)t, (uint32_t)(t >> 32) | (1u << 28), (uint32_t)pc, (uint32_t)(pc >> 32)};
            const u32x4 pn = philox(c0, 1, 0);
            c0.y |= 1u << 16;
            const u32x4 pu = philox(c0, 1, 0);
            const uint64_t v1 = (uint64_t)pn.x ^ ((uint64_t)pn.y << 21);
            const double u = 0x1.0p-53 + (double)v1 * 0x1.0p-53;
            const uint64_t v2 = (uint64_t)pn.z ^ ((uint64_t)pn.w << 21);
            const double w = 0x1.0p-52 + (double)v2 * 0x1.0p-52;
            const double sr = __builtin_sqrt(-2.0 * log_tab(u, &T));
            double sn, cs;
            sincospi_tab(w, sn, cs, &T);
            const bool a0 = mh<true>(xv.x, beta, sigma, den, rden, logc, sn * sr, uni52(pu.x, pu.y), &T);
            const bool a1 = mh<true>(xv.y, beta, sigma, den, rden, logc, cs * sr, uni52(pu.z, pu.w), &T);
            wacc += __popcll(__ballot(a0 && v)) + __popcll(__ballot(a1 && v));
        }
        if (v) *reinterpret_cast<double2*>(x + 2 * p) = xv;
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&slots[blockIdx.x], wacc);
}

int main()
{
    const int64_t M = 10000000, n_pairs = M / 2;
    Tables h;
    for (int j = 0; j < 32; ++j) h.e2[j] = exp2(j / 32.0);
    for (int j = 0; j < 128; ++j) {
        const double c = 1.0 + (j + 0.5) / 128.0;
        h.invc[j] = 1.0 / c; h.logc[j] = log(c);
        h.sn[j] = sin(M_PI * j / 64.0); h.cs[j] = cos(M_PI * j / 64.0);
    }
    Tables* dT; double* dx; unsigned long long* slots;
    hipMalloc(&dT, sizeof(Tables)); hipMemcpy(dT, &h, sizeof(Tables), hipMemcpyHostToDevice);
    hipMalloc(&dx, (M + 2) * sizeof(double));
    std::vector<double> hx(M + 2);
    for (int64_t i = 0; i < M; ++i) hx[i] = -2.0 + 4.0 * ((i * 2654435761u) % 1000003) / 1000003.0;
    hipMemcpy(dx, hx.data(), (M + 2) * sizeof(double), hipMemcpyHostToDevice);
    hipMalloc(&slots, 4096 * 8); hipMemset(slots, 0, 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 2048;
    for (int mode = 0; mode < 2; ++mode) {
        const int nsteps = mode ? 200 : 1, launches = mode ? 1 : 200;
        hipLaunchKernelGGL(sweep, dim3(grid), dim3(256), 0, 0, dx, dT, n_pairs, 20, 0ull, 0.1, 2.0, slots);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int l = 0; l < launches; ++l)
            hipLaunchKernelGGL(sweep, dim3(grid), dim3(256), 0, 0, dx, dT, n_pairs, nsteps, (uint64_t)(100 + l * nsteps), 0.1, 2.0, slots);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per sweep\n", mode ? "fused" : "step1", ms * 1e3 / 200);
    }
    std::vector<unsigned long long> hs(4096);
    hipMemcpy(hs.data(), slots, 4096 * 8, hipMemcpyDeviceToHost);
    unsigned long long tot = 0; for (auto v : hs) tot += v;
    hipMemcpy(hx.data(), dx, M * sizeof(double), hipMemcpyDeviceToHost);
    double s2 = 0; for (int64_t i = 0; i < M; ++i) s2 += hx[i] * hx[i];
    printf("acceptance %.5f  <x^2> %.5f\n", (double)tot / (420.0 * M), s2 / M);
    return 0;
}
