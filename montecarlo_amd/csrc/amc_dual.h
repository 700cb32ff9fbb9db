// amc_dual.h -- forward-mode differentiation of a script-defined log_proposal_density: derivatives the user does not write.
//
// The reference never asks a user for d logq / d theta: withgrad_log_proposal_density! (src/PolicyGuided/gradients.jl:28-33)
// evaluates the model's own log_proposal_density over ForwardDiff's dual numbers -- ForwardDiff.gradient(p -> logq(action,
// policy, p, system), parameters): the PARAMETERS carry unit partials, the action's delta and the system's x are constants --,
// ext/EnzymeExt.jl:8-19 and ext/ZygoteExt.jl do the same with their engines.  Here the expressions are C++ text
// (AMC_USER_LOGQ), so the same thing is an evaluation of that text over Dual<P>: operator overloads give forward mode, one
// pass yields logq (the value parts go through exactly the operations of the plain evaluation: the same bits) AND all P
// partials.  A handle created without dlogq expressions (amc_create_proposal_model / _vector_policy_model / _mixed_model with
// NULL there) takes this route; hand-written derivatives, where given, are used as given.
//
// The rules are ForwardDiff 0.10's (src/dual.jl, src/partials.jl; DiffRules for log), written out so that the oracle's twin
// (the test suite's CPU restatement, its own implementation) and this header perform the same IEEE operations in the same order:
//   x + y, x - y, -x     componentwise
//   x * y                (vx * vy,  (vy * px_i) + (vx * py_i))                       _mul_partials(px, py, vy, vx)
//   x * c, c * x         (vx * c,   px_i * c)
//   x / y                (vx / vy,  (inv(vy) * px_i) + (-(vx / (vy * vy)) * py_i))   _div_partials
//   x / c                (vx / c,   px_i / c)
//   c / y                (q = c / vy,  (-(q / vy)) * py_i)
//   log(x)               (log vx,   inv(vx) * px_i)                                 DiffRules: inv(x)
//   exp(x)               (e = exp vx,  e * px_i)
//   sqrt(x)              (s = sqrt vx,  inv(s + s) * px_i)
//   fabs(x)              signbit(vx) ? -x : x
//   fma(x, y, z)         (fma(vx, vy, vz),  ((vy * px_i) + (vx * py_i)) + pz_i); a constant argument has no partials:
//                        fma(x, c, z) -> (px_i * c) + pz_i,  fma(x, c, c') -> px_i * c,  fma(c, c', z) -> pz_i
// inv(v) = 1.0 / v, an IEEE division.  Comparisons look at the values.  No contraction (-ffp-contract=off).
// What the vocabulary does not cover (a conditional expression whose two arms have different types, say) is a compile error
// that names the operator; such a policy gives its derivatives by hand.
#pragma once

namespace amc {

// sqrt / fabs / fma below would hide the global functions of the same names from every unqualified call inside this namespace
// (a script's `sqrt(fabs(x))` on plain numbers): keep both sets visible
using ::sqrt;
using ::fabs;
using ::fma;

template <int N>
struct Dual {
    double v;
    double d[N];
};

template <int N> __device__ __forceinline__ Dual<N> dual_const(double v)
{
    Dual<N> r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = 0.0;
    return r;
}
// the p-th parameter: unit partial in slot p
template <int N> __device__ __forceinline__ Dual<N> dual_var(double v, int p)
{
    Dual<N> r = dual_const<N>(v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (i == p) ? 1.0 : 0.0;
    return r;
}
// what an expression evaluated to, as a dual: an expression that does not mention the parameters is a constant
template <int N> __device__ __forceinline__ Dual<N> as_dual(const Dual<N>& a) { return a; }
template <int N> __device__ __forceinline__ Dual<N> as_dual(double a) { return dual_const<N>(a); }

#define AMC_DUAL_FN template <int N> __device__ __forceinline__

AMC_DUAL_FN Dual<N> operator+(const Dual<N>& a) { return a; }
AMC_DUAL_FN Dual<N> operator-(const Dual<N>& a)
{
    Dual<N> r;
    r.v = -a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
    return r;
}
AMC_DUAL_FN Dual<N> operator+(const Dual<N>& a, const Dual<N>& b)
{
    Dual<N> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
AMC_DUAL_FN Dual<N> operator+(const Dual<N>& a, double c) { Dual<N> r = a; r.v = a.v + c; return r; }
AMC_DUAL_FN Dual<N> operator+(double c, const Dual<N>& a) { Dual<N> r = a; r.v = c + a.v; return r; }
AMC_DUAL_FN Dual<N> operator-(const Dual<N>& a, const Dual<N>& b)
{
    Dual<N> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
AMC_DUAL_FN Dual<N> operator-(const Dual<N>& a, double c) { Dual<N> r = a; r.v = a.v - c; return r; }
AMC_DUAL_FN Dual<N> operator-(double c, const Dual<N>& a)
{
    Dual<N> r;
    r.v = c - a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
    return r;
}
AMC_DUAL_FN Dual<N> operator*(const Dual<N>& a, const Dual<N>& b)
{
    Dual<N> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (b.v * a.d[i]) + (a.v * b.d[i]);
    return r;
}
AMC_DUAL_FN Dual<N> operator*(const Dual<N>& a, double c)
{
    Dual<N> r;
    r.v = a.v * c;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c;
    return r;
}
AMC_DUAL_FN Dual<N> operator*(double c, const Dual<N>& a)
{
    Dual<N> r;
    r.v = c * a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c;
    return r;
}
AMC_DUAL_FN Dual<N> operator/(const Dual<N>& a, const Dual<N>& b)
{
    Dual<N> r;
    r.v = a.v / b.v;
    const double fa = 1.0 / b.v, fb = -(a.v / (b.v * b.v));
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (fa * a.d[i]) + (fb * b.d[i]);
    return r;
}
AMC_DUAL_FN Dual<N> operator/(const Dual<N>& a, double c)
{
    Dual<N> r;
    r.v = a.v / c;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / c;
    return r;
}
AMC_DUAL_FN Dual<N> operator/(double c, const Dual<N>& b)
{
    Dual<N> r;
    const double q = c / b.v;
    r.v = q;
    const double f = -(q / b.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = f * b.d[i];
    return r;
}

AMC_DUAL_FN bool operator<(const Dual<N>& a, const Dual<N>& b) { return a.v < b.v; }
AMC_DUAL_FN bool operator<(const Dual<N>& a, double b) { return a.v < b; }
AMC_DUAL_FN bool operator<(double a, const Dual<N>& b) { return a < b.v; }
AMC_DUAL_FN bool operator>(const Dual<N>& a, const Dual<N>& b) { return a.v > b.v; }
AMC_DUAL_FN bool operator>(const Dual<N>& a, double b) { return a.v > b; }
AMC_DUAL_FN bool operator>(double a, const Dual<N>& b) { return a > b.v; }
AMC_DUAL_FN bool operator<=(const Dual<N>& a, const Dual<N>& b) { return a.v <= b.v; }
AMC_DUAL_FN bool operator<=(const Dual<N>& a, double b) { return a.v <= b; }
AMC_DUAL_FN bool operator<=(double a, const Dual<N>& b) { return a <= b.v; }
AMC_DUAL_FN bool operator>=(const Dual<N>& a, const Dual<N>& b) { return a.v >= b.v; }
AMC_DUAL_FN bool operator>=(const Dual<N>& a, double b) { return a.v >= b; }
AMC_DUAL_FN bool operator>=(double a, const Dual<N>& b) { return a >= b.v; }

// the vocabulary's functions (amc_log / amc_exp are macros over log_f64 / exp_f64: amc_model.h)
AMC_DUAL_FN Dual<N> log_f64(const Dual<N>& a)
{
    Dual<N> r;
    r.v = log_f64(a.v);
    const double f = 1.0 / a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = f * a.d[i];
    return r;
}
AMC_DUAL_FN Dual<N> exp_f64(const Dual<N>& a, const double* T)
{
    Dual<N> r;
    r.v = exp_f64(a.v, T);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = r.v * a.d[i];
    return r;
}
AMC_DUAL_FN Dual<N> sqrt(const Dual<N>& a)
{
    Dual<N> r;
    r.v = __builtin_sqrt(a.v);
    const double f = 1.0 / (r.v + r.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = f * a.d[i];
    return r;
}
AMC_DUAL_FN Dual<N> fabs(const Dual<N>& a) { return __builtin_signbit(a.v) ? -a : a; }
AMC_DUAL_FN Dual<N> fma(const Dual<N>& a, const Dual<N>& b, const Dual<N>& c)
{
    Dual<N> r;
    r.v = __builtin_fma(a.v, b.v, c.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = ((b.v * a.d[i]) + (a.v * b.d[i])) + c.d[i];
    return r;
}
// fma with constants among its arguments: a constant has no partials (no products with zeros: 0 * Inf would poison a partial)
AMC_DUAL_FN Dual<N> fma(const Dual<N>& a, const Dual<N>& b, double c)
{
    Dual<N> r;
    r.v = __builtin_fma(a.v, b.v, c);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (b.v * a.d[i]) + (a.v * b.d[i]);
    return r;
}
AMC_DUAL_FN Dual<N> fma(const Dual<N>& a, double b, const Dual<N>& c)
{
    Dual<N> r;
    r.v = __builtin_fma(a.v, b, c.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] * b) + c.d[i];
    return r;
}
AMC_DUAL_FN Dual<N> fma(double a, const Dual<N>& b, const Dual<N>& c) { return fma(b, a, c); }
AMC_DUAL_FN Dual<N> fma(const Dual<N>& a, double b, double c)
{
    Dual<N> r;
    r.v = __builtin_fma(a.v, b, c);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b;
    return r;
}
AMC_DUAL_FN Dual<N> fma(double a, const Dual<N>& b, double c) { return fma(b, a, c); }
AMC_DUAL_FN Dual<N> fma(double a, double b, const Dual<N>& c)
{
    Dual<N> r = c;
    r.v = __builtin_fma(a, b, c.v);
    return r;
}

#undef AMC_DUAL_FN

}  // namespace amc
