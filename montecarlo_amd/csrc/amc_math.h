// amc_math.h -- device arithmetic of the engine (gfx950).
//
// Implements the arithmetic spec of DESIGN.md §3: Philox4x32-10, the uniform /
// Box-Muller maps and own exp / log / sincospi in f64.  Everything here is a fixed
// sequence of IEEE-754 operations (+, -, *, /, sqrt, explicit fma, integer bit
// moves), compiled with -ffp-contract=off, so the GPU result is reproducible bit
// for bit by any IEEE host.  ocml's exp/log/sincospi are NOT used: they differ
// from every host libm in the last ulp, which would break accept-count parity.
//
// Constants: tools/gen_math_constants.py (Taylor 1/n!, pi^n/n!) and the published
// fdlibm e_log.c minimax set Lg1..Lg7.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amc {

enum : uint32_t { STREAM_INIT = 0, STREAM_METROPOLIS = 1, STREAM_ESTIMATOR = 2 };
enum : uint32_t { DRAW_NORMAL = 0, DRAW_ACCEPT = 1, DRAW_CATEGORICAL = 2 };

struct u32x4 { uint32_t x, y, z, w; };

// Philox4x32-10 (Salmon et al. SC'11), same rounds/constants as rocRAND's engine
// (rocrand_philox4x32_10.h:270-303).  The key schedule is wave-uniform (SALU).
__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t m1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = __builtin_amdgcn_bitop3_b32((uint32_t)(m1 >> 32), c.y, k0, 0x96);   // a ^ b ^ c in one VALU op
        n.y = (uint32_t)m1;
        n.z = __builtin_amdgcn_bitop3_b32((uint32_t)(m0 >> 32), c.w, k1, 0x96);
        n.w = (uint32_t)m0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// Counter of one draw of one chain pair: x = step[31:0],
// y = step[47:32] | draw<<16 | stream<<28, (z,w) = global pair id.
__device__ __forceinline__ u32x4 draw_counter(uint64_t pair, uint64_t t, uint32_t draw, uint32_t stream)
{
    u32x4 c;
    c.x = (uint32_t)t;
    c.y = ((uint32_t)(t >> 32) & 0xFFFFu) | ((draw & 0xFFFu) << 16) | ((stream & 0xFu) << 28);
    c.z = (uint32_t)pair;
    c.w = (uint32_t)(pair >> 32);
    return c;
}

// rand(rng)::Float64 in [0,1) (metropolis.jl:184): 53 bits.
__device__ __forceinline__ double uniform53(uint32_t lo, uint32_t hi)
{
    const uint64_t v = (uint64_t)lo | ((uint64_t)(hi >> 11) << 32);
    return (double)v * 0x1.0p-53;
}

__device__ __forceinline__ double uniform32(uint32_t v) { return (double)v * 0x1.0p-32; }

// exp(x) for x in [-708, 709] WITHOUT the range / NaN guards (callers mask those cases).
__device__ __forceinline__ double exp_core_f64(double x)
{
    const double LOG2E = 0x1.71547652b82fep+0;
    const double LN2_HI = 0x1.62e42fee00000p-1;
    const double LN2_LO = 0x1.a39ef35793c76p-33;
    const double SHIFT = 0x1.8p52;
    const double t = x * LOG2E + SHIFT;
    const double kd = t - SHIFT;
    const int64_t ki = (int64_t)(int32_t)(uint32_t)(uint64_t)__double_as_longlong(t);
    double r = __builtin_fma(-kd, LN2_HI, x);
    r = __builtin_fma(-kd, LN2_LO, r);
    double p = 0x1.6124613a86d09p-33;
    p = __builtin_fma(p, r, 0x1.1eed8eff8d898p-29);
    p = __builtin_fma(p, r, 0x1.ae64567f544e4p-26);
    p = __builtin_fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);
    p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0x1.0000000000000p-1);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __longlong_as_double(__double_as_longlong(p) + (long long)((uint64_t)ki << 52));
}

// Full-domain exp of the arithmetic spec: 0 below -708 (no subnormals), +inf above 709, NaN -> NaN.
__device__ __forceinline__ double exp_f64(double x)
{
    double y = exp_core_f64(x);
    y = (x < -708.0) ? 0.0 : y;
    y = (x > 709.0) ? __builtin_huge_val() : y;
    y = (x != x) ? x : y;
    return y;
}

// log(x) for POSITIVE NORMAL FINITE x (the Box-Muller radius argument, u in [2^-53, 1]): the spec's
// log without its subnormal pre-scaling and without the 0 / inf / negative / NaN guards.
__device__ __forceinline__ double log_pos_normal_f64(double x)
{
    const double LN2_HI = 0x1.62e42fee00000p-1;
    const double LN2_LO = 0x1.a39ef35793c76p-33;
    const uint64_t ux = (uint64_t)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(ux >> 32);
    int32_t k = (int32_t)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    const uint32_t i = (hx + 0x95f64u) & 0x100000u;
    k += (int32_t)(i >> 20);
    const uint64_t um = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffull);
    const double f = __longlong_as_double((long long)um) - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 0x1.39a09d078c69fp-3, 0x1.c71c51d8e78afp-3),
                                        0x1.999999997fa04p-2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 0x1.2f112df3e5244p-3,
                                                                         0x1.7466496cb03dep-3),
                                                         0x1.2492494229359p-2),
                                        0x1.5555555555593p-1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * LN2_HI - ((hfsq - __builtin_fma(s, hfsq + R, dk * LN2_LO)) - f);
}

__device__ __forceinline__ double log_f64(double x)
{
    const double LN2_HI = 0x1.62e42fee00000p-1;
    const double LN2_LO = 0x1.a39ef35793c76p-33;
    const double x0 = x;
    int64_t k = 0;
    uint64_t ux = (uint64_t)__double_as_longlong(x);
    if (ux < 0x0010000000000000ull) {
        x = x * 0x1.0p54;
        ux = (uint64_t)__double_as_longlong(x);
        k = -54;
    }
    uint32_t hx = (uint32_t)(ux >> 32);
    k += (int64_t)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    const uint32_t i = (hx + 0x95f64u) & 0x100000u;
    k += (int64_t)(i >> 20);
    const uint64_t um = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffull);
    const double f = __longlong_as_double((long long)um) - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 0x1.39a09d078c69fp-3, 0x1.c71c51d8e78afp-3),
                                        0x1.999999997fa04p-2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 0x1.2f112df3e5244p-3,
                                                                         0x1.7466496cb03dep-3),
                                                         0x1.2492494229359p-2),
                                        0x1.5555555555593p-1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    double y = dk * LN2_HI - ((hfsq - __builtin_fma(s, hfsq + R, dk * LN2_LO)) - f);
    y = (x0 == __builtin_huge_val()) ? x0 : y;
    y = (x0 == 0.0) ? -__builtin_huge_val() : y;
    y = (x0 != x0 || x0 < 0.0) ? __builtin_nan("") : y;
    return y;
}

__device__ __forceinline__ void sincospi_f64(double w, double& sp, double& cp)
{
    const double SHIFT = 0x1.8p52;
    const double t = (w + w) + SHIFT;
    const double nd = t - SHIFT;
    const uint32_t n = (uint32_t)(uint64_t)__double_as_longlong(t);
    const double r = w - 0.5 * nd;
    const double z = r * r;
    double ps = -0x1.6fadb9f155744p-16;
    ps = __builtin_fma(ps, z, 0x1.e8f434d018d63p-12);
    ps = __builtin_fma(ps, z, -0x1.e3074fde8871fp-8);
    ps = __builtin_fma(ps, z, 0x1.50783487ee782p-4);
    ps = __builtin_fma(ps, z, -0x1.32d2cce62bd86p-1);
    ps = __builtin_fma(ps, z, 0x1.466bc6775aae2p+1);
    ps = __builtin_fma(ps, z, -0x1.4abbce625be53p+2);
    double s = __builtin_fma(r, 0x1.1a62633145c07p-53, (r * z) * ps);
    s = __builtin_fma(r, 0x1.921fb54442d18p+1, s);
    double pc = 0x1.20c62c2f2d7f5p-18;
    pc = __builtin_fma(pc, z, -0x1.b6e24f44b128fp-14);
    pc = __builtin_fma(pc, z, 0x1.f9d38a3763cc3p-10);
    pc = __builtin_fma(pc, z, -0x1.a6d1f2a204a8cp-6);
    pc = __builtin_fma(pc, z, 0x1.e1f506891babbp-3);
    pc = __builtin_fma(pc, z, -0x1.55d3c7e3cbffap+0);
    pc = __builtin_fma(pc, z, 0x1.03c1f081b5ac4p+2);
    pc = __builtin_fma(pc, z, -0x1.3bd3cc9be45dep+2);
    const double c = __builtin_fma(pc, z, 1.0);
    const bool swap = (n & 1u) != 0u;
    const double a = swap ? c : s;      // |sin| carrier
    const double b = swap ? s : c;      // |cos| carrier
    // n&3: 0 (s, c); 1 (c, -s); 2 (-s, -c); 3 (-c, s): negation = flipping the sign bit
    const uint64_t sa = (uint64_t)(n & 2u) << 62;
    const uint64_t sb = (uint64_t)((n + 1u) & 2u) << 62;
    sp = __longlong_as_double((long long)((uint64_t)__double_as_longlong(a) ^ sa));
    cp = __longlong_as_double((long long)((uint64_t)__double_as_longlong(b) ^ sb));
}

// rocRAND box_muller_double(uint4) map (rocrand_normal.h:78-98) on own log/sincospi:
// one Philox result -> two standard normals, one per chain of the pair.
__device__ __forceinline__ void box_muller(u32x4 v, double& z0, double& z1)
{
    const uint64_t v1 = (uint64_t)v.x ^ ((uint64_t)v.y << 21);
    const double u = 0x1.0p-53 + (double)v1 * 0x1.0p-53;
    const uint64_t v2 = (uint64_t)v.z ^ ((uint64_t)v.w << 21);
    const double w = 0x1.0p-52 + (double)v2 * 0x1.0p-52;
    const double s = __builtin_sqrt(-2.0 * log_pos_normal_f64(u));   // u in [2^-53, 1]: positive normal
    double sn, cs;
    sincospi_f64(w, sn, cs);
    z0 = sn * s;
    z1 = cs * s;
}

// Julia's min(a, b): NaN if either operand is NaN.
__device__ __forceinline__ double julia_min(double a, double b)
{
    double m = (b < a) ? b : a;
    m = (b != b) ? b : m;
    m = (a != a) ? a : m;
    return m;
}

}  // namespace amc
