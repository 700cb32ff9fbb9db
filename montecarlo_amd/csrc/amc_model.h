// amc_model.h -- the model on the path: configuration of a translation unit (state type, policy parameters, classes), the script-defined
// hooks (potential, reward, proposal, action), potential / proposal / acceptance of the built-in Gaussian displacement, the accept
// filter, one mc_step! of a chain pair (mh_pair) and the 16-byte loads / write-through stores of a pair.
// Part of the kernel sources of the many-chain Metropolis engine (gfx950 / CDNA4); amc_kernels.h includes all of them, in order.
#pragma once

#include "amc_math.h"
#include "amc_xsum.h"

#define AMC_MAX_MOVES 64
#define AMC_MAX_LEARN 8
#ifndef AMC_BLOCK
#define AMC_BLOCK 256
#endif
#define AMC_PAD_DOUBLES (2 * AMC_BLOCK + 8)   // readable padding behind every per-chain array: a ragged last
                                 // block-iteration may load up to AMC_BLOCK - 1 pairs past the end without a clamp

namespace amc {

enum { POT_HARMONIC = 0, POT_DOUBLE_WELL = 1, POT_CUSTOM = 2 };

// Linkage of the kernels that are no templates.  The header is compiled into more than one object (amc_pg_fused.hip holds some
// instantiations of pg_estimate_kernel, built with other code-generation options); an object that only wants template
// instantiations defines this as `static` and, not using them, emits none of these kernels.
// AMC_PLAIN_KERNELS 0 (the run-time compiler, building ONE template instantiation): the headers' plain kernels -- initial ensemble,
// parameter tables, accumulate / update, selftest hooks, some 9 000 instructions of ISA -- are left out of the translation unit
#ifndef AMC_PLAIN_KERNELS
#define AMC_PLAIN_KERNELS 1
#endif
#ifndef AMC_KERNEL_LINKAGE
#define AMC_KERNEL_LINKAGE
#endif

// State type.  The reference's Particle{T} / Displacement{T} are generic in T <: AbstractFloat (particle_1d.jl:9,26);
// Float64 is what its scripts use and what the offline build of this header compiles.  A handle created with
// state_dtype = AMC_DTYPE_F32 gets the SAME kernel sources compiled at run time with AMC_STATE_F32 defined: x, beta, e,
// delta and dlogp are then Float32 exactly where Julia's promotion rules keep them Float32 -- the policy parameters,
// the normal variate, log_proposal_density, the acceptance probability and the uniforms stay Float64
// (ComponentArray(sigma = 0.1) is Float64; Normal(0f0, sigma) promotes; rand(rng) is Float64).  DESIGN.md section 3.7.
#ifdef AMC_STATE_F32
typedef float real_t;
typedef float2 real2;
#else
typedef double real_t;
typedef double2 real2;
#endif

// Rows of the per-move parameter table.
enum { PT_SIGMA = 0, PT_DEN = 1, PT_LOGC = 2, PT_CUM = 3, PT_DDEN = 4, PT_DLHALF = 5, PT_WEIGHT = 6,
       PT_RDEN = 7, PT_C3HI = 8, PT_C3LO = 9,
       PT_THETA1 = 10, PT_THETA2 = 11, PT_THETA3 = 12,      // parameters 1..3 of a script-defined policy with several (AMC_NP)
       PT_CLASS = 13,                                       // the move's policy / action class (pools that mix them: AMC_NCLASS)
       PT_ROWS = 14 };

// Parameters of a move's policy (Move.parameters, src/metropolis.jl:140-147: an array; GradientData keeps grad j and
// grad logq as arrays of that shape and g as their outer product, PolicyGuided/gradients.jl:41-61).  The built-in Gaussian
// displacement has one (sigma); a script-defined policy (amc_create_vector_policy_model) may have up to four, theta0 (= PT_SIGMA's
// row, `sigma` in the expressions) .. theta3: the translation unit hiprtc compiles for it defines AMC_NP.  Everything that
// serves AMC_NP > 1 is behind `#if AMC_NP > 1` or a constant that is 4 for AMC_NP == 1: the one-parameter kernels are the
// code they were.
#ifndef AMC_NP
#define AMC_NP 1
#endif
// Pools that MIX policy / action types: every Move carries its own `action` and `policy` (metropolis.jl:140-162), and the
// generic functions dispatch on their types.  A handle of amc_create_mixed_model has up to AMC_MAX_CLASSES expression sets
// ("classes": sample, logq, dlogq, perform, invert), one class per move (PT_CLASS); the translation unit defines AMC_NCLASS and,
// for the classes 1 .. 3, the macros suffixed _1 .. _3 (class 0: the unsuffixed ones).  One parameter per move.
#ifndef AMC_NCLASS
#define AMC_NCLASS 1
#endif
#define AMC_MAX_CLASSES 4
#define AMC_MAX_NP 4
// GradientData columns of one learnable move: j, grad j [NP], grad logq [NP], g [upper triangle, row by row]
#define AMC_PG_NC (1 + 2 * AMC_NP + AMC_NP * (AMC_NP + 1) / 2)

// a / b, correctly rounded, for a divisor b whose reciprocal y = RN(1/b) is precomputed
// (b = 2 sigma^2 is one value per move).  Two Markstein corrections: q1 is a faithful rounding of
// a/b (|q0 - a/b| < 1.5 ulp, so q0 + r0*y is within 2^-52 relative of a/b before its one rounding),
// and Markstein's theorem (Muller et al., Handbook of FP Arithmetic, 2nd ed., Thm 4.10: q faithful,
// |y - 1/b| < 2^-53 |1/b|, r = a - bq exact  =>  RN(q + r y) = RN(a/b)) makes q2 the IEEE quotient.
// Needs a, b, 1/b normal or a == 0: guaranteed by the 1e-100 <= sigma <= 1e100 check in the C ABI.
// (a == -0.0 returns +0.0; both consumers subtract a non-zero constant next, so it never shows.)
// 5 f64 ops instead of v_div_scale x2 + v_rcp_f64 + 7 fma + v_div_fmas + v_div_fixup.
__device__ __forceinline__ double div_by_const(double a, double b, double y)
{
    const double q0 = a * y;
    const double r0 = __builtin_fma(-q0, b, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-q1, b, a);
    return __builtin_fma(r1, y, q1);
}

// POT_CUSTOM: `potential` is a free function the driver SCRIPT defines in the reference
// (harmonic_oscillator/MC_harmonic_oscillator.jl:4; docs/src/man/system.md).  A Julia closure cannot cross the
// C ABI, so amc_create_custom takes the body as a C expression in `x` and the kernels of this header are
// compiled for it at run time (hiprtc, amc_api.hip: the translation unit defines AMC_USER_POTENTIAL before
// including this file).  The expression sees IEEE + - * / (no contraction: -ffp-contract=off), sqrt, fabs, fma,
// and the arithmetic spec's own amc_exp / amc_log (bit-reproducible on any IEEE host).
#ifndef AMC_USER_POTENTIAL
#define AMC_USER_POTENTIAL(x) (x)        // offline build: POT_CUSTOM kernels are never instantiated
#endif
// reward(action, system) (gradients.jl:20; the model's is particle_1d.jl:42-44, delta^2): likewise script-defined in the
// reference, evaluated right after perform_action! (gradients.jl:100) -- an expression in `delta` and the NEW position
// `x`, same vocabulary as the potential.
#ifndef AMC_USER_REWARD
#define AMC_USER_REWARD(delta, x) ((delta) * (delta))
#endif
// A script-defined POLICY of the Gaussian-displacement family (sample_action! / log_proposal_density are the model's,
// particle_1d.jl:48-59; the reference hands them `system`, so the width may depend on the state): the proposal width is
// sigma * scale(x) with AMC_USER_SCALE an expression in the CURRENT position x,
//   sample_action!        delta = rand(rng, Normal(0, sigma*scale(x)))            = 0 + (sigma*scale(x)) * z
//   log_proposal_density  -(delta)^2 / (2 (sigma*scale(x))^2) - log(2pi (sigma*scale(x))^2) / 2
// The forward density is evaluated at the old state, the backward one at the new state (mc_step! metropolis.jl:178,182
// call it before and after perform_action!), so logq_b != logq_f and the proposal ratio is real.  Undefined (offline
// build, every handle without a scale expression): scale == 1, the StandardGaussian policy of the reference.
#define amc_exp(v) (::amc::exp_f64((v), amc_tables_))
#define amc_log(v) (::amc::log_f64((v)))
__device__ __forceinline__ real_t user_potential(real_t x, const double* amc_tables_)
{
    return (real_t)(AMC_USER_POTENTIAL(x));       // Particle.e is a field of type T: the value is converted on assignment
}

__device__ __forceinline__ double user_reward(real_t delta, real_t x, const double* amc_tables_)
{
    return (double)(AMC_USER_REWARD(delta, x));
}
#ifdef AMC_USER_SCALE
__device__ __forceinline__ double user_scale(real_t x, const double* amc_tables_)
{
    return (double)(real_t)(AMC_USER_SCALE(x));      // a function of the system returns T; sigma * scale promotes
}
#endif
// A script-defined PROPOSAL in full (amc_create_proposal_model): the model's own sample_action! and
// log_proposal_density (example/particle_1d/particle_1d.jl:52-59 are the particle_1d model's; src/metropolis.jl:35-62
// only declare the generic functions), each as one expression:
//   AMC_USER_SAMPLE(z, x, sigma)      delta, from ONE standard normal variate z (the engine's Box-Muller draw of the
//                                     step), the current position x and the move's parameter sigma
//   AMC_USER_LOGQ(delta, x, sigma)    log q(delta | x, sigma): the density of what AMC_USER_SAMPLE returns
//   AMC_USER_DLOGQ(delta, x, sigma)   its derivative with respect to sigma (what the reference gets from ForwardDiff /
//                                     Enzyme / Zygote, gradients.jl:28-33); optional, needed by the estimator only
// mc_step! (metropolis.jl:176-190) evaluates the forward density at the old state and the backward one, of the
// inverted action, at the new state; nothing cancels, arg is formed in full in the reference's operations (and since round 5 decided
// through the accept filter like the built-in path's: accept_filter_arg).
// ... and a script-defined ACTION (the reference's Action interface, src/metropolis.jl:15-119: perform_action!,
// invert_action!, perform_action_cached!; example/particle_1d/particle_1d.jl:30-40 are the displacement's methods), for a
// one-parameter action on the position:
//   AMC_USER_PERFORM(x, delta)    the position after perform_action!(system, action)        (displacement: x + delta)
//   AMC_USER_INVERT(delta, x)     the parameter of the inverted action, given the NEW state (displacement: -delta)
// perform_action_cached! (the revert) re-applies the inverted action, as the reference does (metropolis.jl:119,187).
// `sigma` as the expressions see it in a K > 1 sweep: the lane's sigma with log(sigma) beside it.  There the lanes of a wave hold
// different moves, so no function of sigma alone leaves the loop the way it does where the move is wave-uniform -- but sigma takes
// only K values: the block forms amc_log(sigma_k) once per launch (row 5 of the sweep's LDS table, the same log_f64 on the same
// operand: the same bits) and `amc_log(sigma)` in an expression -- the Langevin proposal's normalisation, say -- reads it instead of
// spending fifty vector instructions per lane and step.  Everywhere else SigmaArg is a double: it converts, and only this one call
// knows it.  AMC_SIGMA_MEMO: the row exists (script-defined proposals; AMC_NO_SIGMA_MEMO, set by the run-time compiler for A/B, drops it).
#if defined(AMC_USER_LOGQ) && !defined(AMC_NO_SIGMA_MEMO)
#define AMC_SIGMA_MEMO 1
#else
#define AMC_SIGMA_MEMO 0
#endif
struct SigmaArg {
    double v, logv;
    __device__ __forceinline__ operator double() const { return v; }
};
__device__ __forceinline__ double log_f64(const SigmaArg& s) { return s.logv; }
}  // namespace amc
#include "amc_dual.h"          // forward-mode differentiation of script-defined densities (dual numbers over the same vocabulary)
namespace amc {

struct UserTheta {          // a script-defined policy's parameters theta1 .. theta3 of one move (see AMC_USER_THETAS), their logs beside them
    double t1, t2, t3;
    double l1, l2, l3;      // (AMC_SIGMA_MEMO: what amc_log(theta_p) reads -- per move from the block's table in a K > 1 sweep)
};
#ifdef AMC_USER_LOGQ
#ifndef AMC_USER_PERFORM
#define AMC_USER_PERFORM(x, delta) ((x) + (delta))
#endif
#ifndef AMC_USER_INVERT
#define AMC_USER_INVERT(delta, x) (-(delta))
#endif
#if AMC_NCLASS > 1
__shared__ int s_user_class[AMC_MAX_MOVES];
// the expression of move k's class: class 0 the unsuffixed macro, classes 1 .. 3 (where the pool has them) the suffixed ones
// (k carries the class in its bits 8 and up -- user_move_key: per lane in the sweep, where the lanes of a wave hold different
// moves; read through the scalar unit in the estimator, whose move is the launch's)
#define AMC_BY_CLASS(k, T_, E0, E1, E2, E3)                                                                            \
    do {                                                                                                               \
        const int cls_ = (k) >> 8;                                                                                     \
        if (cls_ == 1) return (T_)(E1);                                                                                \
        if (AMC_NCLASS > 2 && cls_ == 2) return (T_)(E2);                                                              \
        if (AMC_NCLASS > 3 && cls_ == 3) return (T_)(E3);                                                              \
        return (T_)(E0);                                                                                               \
    } while (0)
#if AMC_NCLASS < 3
#define AMC_USER_SAMPLE_2 AMC_USER_SAMPLE
#define AMC_USER_LOGQ_2 AMC_USER_LOGQ
#define AMC_USER_PERFORM_2 AMC_USER_PERFORM
#define AMC_USER_INVERT_2 AMC_USER_INVERT
#endif
#if AMC_NCLASS < 4
#define AMC_USER_SAMPLE_3 AMC_USER_SAMPLE
#define AMC_USER_LOGQ_3 AMC_USER_LOGQ
#define AMC_USER_PERFORM_3 AMC_USER_PERFORM
#define AMC_USER_INVERT_3 AMC_USER_INVERT
#endif
// d logq / d sigma of a class: AMC_USER_DLOGQ (class 0) / AMC_USER_DLOGQ_c where the script gives the expression; a class without
// one gets its derivative by forward-mode differentiation of its logq (user_logq_dual below, amc_dual.h)
#endif
__device__ __forceinline__ real_t user_perform(real_t x, real_t delta, const double* amc_tables_, int k)
{
#if AMC_NCLASS > 1
    AMC_BY_CLASS(k, real_t, AMC_USER_PERFORM(x, delta), AMC_USER_PERFORM_1(x, delta), AMC_USER_PERFORM_2(x, delta), AMC_USER_PERFORM_3(x, delta));
#else
    (void)k;
    return (real_t)(AMC_USER_PERFORM(x, delta));
#endif
}
__device__ __forceinline__ real_t user_invert(real_t delta, real_t x, const double* amc_tables_, int k)
{
#if AMC_NCLASS > 1
    AMC_BY_CLASS(k, real_t, AMC_USER_INVERT(delta, x), AMC_USER_INVERT_1(delta, x), AMC_USER_INVERT_2(delta, x), AMC_USER_INVERT_3(delta, x));
#else
    (void)k;
    return (real_t)(AMC_USER_INVERT(delta, x));
#endif
}
// The move's further parameters (AMC_NP > 1): the expressions see them as theta1 .. theta3, and theta0 is another name of sigma.
// They reach the user_* functions as a UserTheta VALUE.  Where the move is the same for the whole wave -- the K == 1 sweep, the
// estimator (its launch's learnable move) -- the kernel reads them once, at its start, through the scalar unit
// (user_theta_uniform): every subexpression of the script that depends on the parameters alone (log(theta1), 1/theta1,
// theta1*theta1*theta1, the reciprocal refinements of a division by them) is then a loop invariant the compiler forms ONCE per
// wave.  (Round 5: read from the LDS copy at every use, as the K > 1 sweep must -- its lanes hold different moves --, none of it
// could leave the loop, and the two-parameter drift + width policy paid ~1070 vector instructions per wave-trip.)
#if AMC_NP > 1
__shared__ double s_user_theta[AMC_MAX_NP - 1][AMC_MAX_MOVES];
#if AMC_SIGMA_MEMO
__shared__ double s_user_theta_log[AMC_MAX_NP - 1][AMC_MAX_MOVES];       // log(theta_p) of every move (stage_user_theta), see SigmaArg
// theta0 is sigma as the function got it (a SigmaArg in a K > 1 sweep); theta1 .. theta3 carry their logs the same way
#define AMC_USER_THETAS(th)                                                                                            \
    const auto theta0 = sigma;                                                                                         \
    const SigmaArg theta1 = {(th).t1, (th).l1}, theta2 = {(th).t2, (th).l2}, theta3 = {(th).t3, (th).l3};              \
    (void)theta0; (void)theta1; (void)theta2; (void)theta3
#else
#define AMC_USER_THETAS(th)                                                                                            \
    const double theta0 = (double)sigma, theta1 = (th).t1, theta2 = (th).t2, theta3 = (th).t3;                                  \
    (void)theta0; (void)theta1; (void)theta2; (void)theta3
#endif
#else
#define AMC_USER_THETAS(th) const double theta0 = (double)sigma; (void)theta0; (void)th
#endif
// k: the move key (user_move_key).  Per lane, from the LDS copy staged by stage_user_theta:
__device__ __forceinline__ UserTheta user_theta_lds(int k)
{
    UserTheta th = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#if AMC_NP > 1
    th.t1 = s_user_theta[0][k & 0xFF];
    if (AMC_NP > 2) th.t2 = s_user_theta[1][k & 0xFF];
    if (AMC_NP > 3) th.t3 = s_user_theta[2][k & 0xFF];
#if AMC_SIGMA_MEMO
    th.l1 = s_user_theta_log[0][k & 0xFF];
    if (AMC_NP > 2) th.l2 = s_user_theta_log[1][k & 0xFF];
    if (AMC_NP > 3) th.l3 = s_user_theta_log[2][k & 0xFF];
#endif
#else
    (void)k;
#endif
    return th;
}
// ... and of a move the whole wave shares (k wave-uniform), from the parameter table itself: scalar loads
__device__ __forceinline__ UserTheta user_theta_uniform(const double* ptab, int k)
{
    UserTheta th = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#if AMC_NP > 1
    th.t1 = ptab[PT_THETA1 * AMC_MAX_MOVES + (k & 0xFF)];
    if (AMC_NP > 2) th.t2 = ptab[(PT_THETA1 + 1) * AMC_MAX_MOVES + (k & 0xFF)];
    if (AMC_NP > 3) th.t3 = ptab[(PT_THETA1 + 2) * AMC_MAX_MOVES + (k & 0xFF)];
#if AMC_SIGMA_MEMO
    // wave-uniform values formed once, at the kernel's start (and not at all where no expression asks for a log)
    th.l1 = log_f64(th.t1);
    if (AMC_NP > 2) th.l2 = log_f64(th.t2);
    if (AMC_NP > 3) th.l3 = log_f64(th.t3);
#endif
#else
    (void)ptab; (void)k;
#endif
    return th;
}
// with_logs: the kernel reads the per-lane copy (a K > 1 sweep) -- the block forms log(theta_p) of every move beside it
__device__ __forceinline__ void stage_user_theta(const double* ptab, bool with_logs = false)       // before a barrier the caller already has
{
#if AMC_NCLASS > 1
    for (int i = threadIdx.x; i < AMC_MAX_MOVES; i += AMC_BLOCK) s_user_class[i] = (int)ptab[PT_CLASS * AMC_MAX_MOVES + i];
#endif
#if AMC_NP > 1
    for (int i = threadIdx.x; i < (AMC_NP - 1) * AMC_MAX_MOVES; i += AMC_BLOCK)
        s_user_theta[i / AMC_MAX_MOVES][i % AMC_MAX_MOVES] = ptab[(PT_THETA1 + i / AMC_MAX_MOVES) * AMC_MAX_MOVES + i % AMC_MAX_MOVES];
#if AMC_SIGMA_MEMO
    if (with_logs)
        for (int i = threadIdx.x; i < (AMC_NP - 1) * AMC_MAX_MOVES; i += AMC_BLOCK)
            s_user_theta_log[i / AMC_MAX_MOVES][i % AMC_MAX_MOVES] = log_f64(ptab[(PT_THETA1 + i / AMC_MAX_MOVES) * AMC_MAX_MOVES + i % AMC_MAX_MOVES]);
#else
    (void)with_logs;
#endif
#else
    (void)with_logs;
    (void)ptab;
#endif
}
// what the user_* functions take as `k`: the move, with its class above bit 8 in pools that mix classes
__device__ __forceinline__ int user_move_key(int k)                    // the sweep: the lane's move, class from the LDS copy
{
#if AMC_NCLASS > 1
    return k | (s_user_class[k] << 8);
#else
    return k;
#endif
}
__device__ __forceinline__ int user_move_key_uniform(int k, const double* ptab)      // the estimator: the launch's move
{
#if AMC_NCLASS > 1
    return k | ((int)ptab[PT_CLASS * AMC_MAX_MOVES + k] << 8);
#else
    (void)ptab;
    return k;
#endif
}
template <class S>
__device__ __forceinline__ real_t user_sample(double z, real_t x, S sigma, const double* amc_tables_, int k, const UserTheta& th)
{
    AMC_USER_THETAS(th);
#if AMC_NCLASS > 1
    AMC_BY_CLASS(k, real_t, AMC_USER_SAMPLE(z, x, sigma), AMC_USER_SAMPLE_1(z, x, sigma), AMC_USER_SAMPLE_2(z, x, sigma), AMC_USER_SAMPLE_3(z, x, sigma));
#else
    return (real_t)(AMC_USER_SAMPLE(z, x, sigma));    // Displacement.delta::T
#endif
}
template <class S>
__device__ __forceinline__ double user_logq(real_t delta, real_t x, S sigma, const double* amc_tables_, int k, const UserTheta& th)
{
    AMC_USER_THETAS(th);
#if AMC_NCLASS > 1
    AMC_BY_CLASS(k, double, AMC_USER_LOGQ(delta, x, sigma), AMC_USER_LOGQ_1(delta, x, sigma), AMC_USER_LOGQ_2(delta, x, sigma), AMC_USER_LOGQ_3(delta, x, sigma));
#else
    return (double)(AMC_USER_LOGQ(delta, x, sigma));
#endif
}
// Forward-mode differentiation (amc_dual.h): the class's log_proposal_density evaluated over dual numbers whose partials belong to
// the move's parameters -- ForwardDiff.gradient(p -> log_proposal_density(action, policy, p, system), parameters),
// src/PolicyGuided/gradients.jl:28-33; delta and x are constants there and here.  Returns logq (the value parts go through the
// plain evaluation's operations: the same bits as user_logq) and leaves d[p] = d logq / d theta_p.
template <class S>
__device__ __forceinline__ double user_logq_dual(real_t delta, real_t x, S sigma_in, const double* amc_tables_, int k, const UserTheta& th,
                                                 double (&d)[AMC_NP])
{
    typedef Dual<AMC_NP> D_;
    const D_ sigma = dual_var<AMC_NP>((double)sigma_in, 0), theta0 = sigma;
    // theta1 .. theta3: parameters 1 .. 3 where the policy has them (a slot beyond AMC_NP carries no partial: a constant)
    const D_ theta1 = dual_var<AMC_NP>(th.t1, 1), theta2 = dual_var<AMC_NP>(th.t2, 2), theta3 = dual_var<AMC_NP>(th.t3, 3);
    (void)theta0; (void)theta1; (void)theta2; (void)theta3; (void)k;
    D_ r;
#if AMC_NCLASS > 1
    const int cls_ = k >> 8;
    if (cls_ == 1) r = as_dual<AMC_NP>(AMC_USER_LOGQ_1(delta, x, sigma));
    else if (AMC_NCLASS > 2 && cls_ == 2) r = as_dual<AMC_NP>(AMC_USER_LOGQ_2(delta, x, sigma));
    else if (AMC_NCLASS > 3 && cls_ == 3) r = as_dual<AMC_NP>(AMC_USER_LOGQ_3(delta, x, sigma));
    else r = as_dual<AMC_NP>(AMC_USER_LOGQ(delta, x, sigma));
#else
    r = as_dual<AMC_NP>(AMC_USER_LOGQ(delta, x, sigma));
#endif
#pragma unroll
    for (int p = 0; p < AMC_NP; ++p) d[p] = r.d[p];
    return r.v;
}
// no class of the pool brings a derivative expression: ONE dual evaluation serves log_proposal_density and its gradient
#if !defined(AMC_USER_DLOGQ) && !defined(AMC_USER_DLOGQ_1) && !defined(AMC_USER_DLOGQ_2) && !defined(AMC_USER_DLOGQ_3)
#define AMC_DLOGQ_ALL_AUTO 1
#else
#define AMC_DLOGQ_ALL_AUTO 0
#endif
// grad log_proposal_density with respect to the parameters, d[p] = d logq / d theta_p
template <class S>
__device__ __forceinline__ void user_dlogq(real_t delta, real_t x, S sigma, const double* amc_tables_, int k, const UserTheta& th,
                                           double (&d)[AMC_NP])
{
#if AMC_NCLASS > 1
    // one parameter per move; the class's own expression, or its logq differentiated
    const int cls_ = k >> 8;
    double da_[AMC_NP];
    (void)da_;
    AMC_USER_THETAS(th);
    if (cls_ == 1) {
#ifdef AMC_USER_DLOGQ_1
        d[0] = (double)(AMC_USER_DLOGQ_1(delta, x, sigma));
#else
        (void)user_logq_dual(delta, x, sigma, amc_tables_, k, th, da_); d[0] = da_[0];
#endif
        return;
    }
#if AMC_NCLASS > 2
    if (cls_ == 2) {
#ifdef AMC_USER_DLOGQ_2
        d[0] = (double)(AMC_USER_DLOGQ_2(delta, x, sigma));
#else
        (void)user_logq_dual(delta, x, sigma, amc_tables_, k, th, da_); d[0] = da_[0];
#endif
        return;
    }
#endif
#if AMC_NCLASS > 3
    if (cls_ == 3) {
#ifdef AMC_USER_DLOGQ_3
        d[0] = (double)(AMC_USER_DLOGQ_3(delta, x, sigma));
#else
        (void)user_logq_dual(delta, x, sigma, amc_tables_, k, th, da_); d[0] = da_[0];
#endif
        return;
    }
#endif
#ifdef AMC_USER_DLOGQ
    d[0] = (double)(AMC_USER_DLOGQ(delta, x, sigma));
#else
    (void)user_logq_dual(delta, x, sigma, amc_tables_, k, th, da_); d[0] = da_[0];
#endif
#elif defined(AMC_USER_DLOGQ)
    AMC_USER_THETAS(th);
    d[0] = (double)(AMC_USER_DLOGQ(delta, x, sigma));
#if AMC_NP > 1
    d[1] = (double)(AMC_USER_DLOGQ1(delta, x, sigma));
#endif
#if AMC_NP > 2
    d[2] = (double)(AMC_USER_DLOGQ2(delta, x, sigma));
#endif
#if AMC_NP > 3
    d[3] = (double)(AMC_USER_DLOGQ3(delta, x, sigma));
#endif
#else
    (void)user_logq_dual(delta, x, sigma, amc_tables_, k, th, d);
#endif
}
// log_proposal_density and its gradient at one point (what withgrad_log_proposal_density! returns and leaves, gradients.jl:28-33)
template <class S>
__device__ __forceinline__ double user_logq_dlogq(real_t delta, real_t x, S sigma, const double* amc_tables_, int k, const UserTheta& th,
                                                  double (&d)[AMC_NP])
{
#if AMC_DLOGQ_ALL_AUTO
    return user_logq_dual(delta, x, sigma, amc_tables_, k, th, d);
#else
    const double lq = user_logq(delta, x, sigma, amc_tables_, k, th);
    user_dlogq(delta, x, sigma, amc_tables_, k, th, d);
    return lq;
#endif
}
#endif
#undef amc_exp
#undef amc_log

// potential(x): harmonic_oscillator/MC_harmonic_oscillator.jl:4 (x^2 == x*x);
// double well (x*x-1)^2 is BASELINE config 3's.  T: the block's LDS copy of the math tables (custom only).
template <int POT>
__device__ __forceinline__ real_t potential(real_t x, const double* T)
{
    if (POT == POT_CUSTOM) return user_potential(x, T);
    if (POT == POT_DOUBLE_WELL) {
        const real_t q = x * x - (real_t)1.0;
        return q * q;
    }
    return x * x;
}

// One mc_step! (metropolis.jl:176-190) on the particle_1d model, in the reference's
// operation order:
//   sample_action!        particle_1d.jl:56-59   delta = 0 + sigma*z
//   log_proposal_density  particle_1d.jl:52-54   logq = -(d*d)/(2 s^2) - log(2pi s^2)/2
//   perform_action!       particle_1d.jl:30-35   e1 = e; x += delta; e2 = potential(x)
//   delta_log_target      metropolis.jl:74 + particle_1d.jl:20-22   (-e2*b) - (-e1*b)
//   invert_action!        particle_1d.jl:37-40   logq_b == logq_f bit for bit
//   alpha = min(1, exp(dlogp + logq_b - logq_f)); accept iff alpha > u  (strict)
//   reject: perform_action_cached! re-applies the negated action: x = (x+d) + (-d)
// split in three: the part every chain needs in f64 (propose), the exact accept decision in the reference's
// arithmetic (accept_exact), and a floating-point FILTER that settles the decision from a float estimate whenever the
// estimate's rigorous error interval does not contain u (accept_filter) -- the same idea as the filtered exact
// predicates of computational geometry.  logq, arg and exp(arg) feed nothing but that one comparison.
struct Proposal {
    real_t delta, xn, dlogp;
};

template <int POT>
__device__ __forceinline__ Proposal propose(real_t x, real_t beta, double sigma, double z, const double* T)
{
    Proposal p;
    // Displacement.delta::T = rand(rng, Normal(zero(T), sigma::Float64)) = 0.0 + sigma*z.  fma(sigma, z, 0.0) is that value
    // bit for bit in every case: the product is rounded once either way and adding +0.0 changes nothing but the sign of a
    // zero product (-0.0 -> +0.0 in both forms; NaN and infinities pass through alike).  One instruction instead of two.
    p.delta = (real_t)__builtin_fma(sigma, z, 0.0);
    const real_t e1 = potential<POT>(x, T);
    p.xn = x + p.delta;
    const real_t e2 = potential<POT>(p.xn, T);
    p.dlogp = ((-e2) * beta) - ((-e1) * beta);
    return p;
}

// The reference-ordered decision.  alpha = min(1, exp(arg)); accept iff alpha > u, with u in [0, 1).  Decided
// without forming alpha:
//   arg >= 0           -> exp(arg) >= 1 -> alpha == 1 > u           : accept
//   -708 <= arg < 0    -> alpha == exp(arg) (<= 1)                  : accept iff exp(arg) > u
//   arg < -708 or NaN  -> alpha == 0 or NaN (Julia's min keeps NaN) : reject
// (bitwise | and & on purpose: no short-circuit branches)
__device__ __forceinline__ bool accept_exact(real_t delta, real_t dlogp, double den, double rden, double logc, double u,
                                             const double* T)
{
    // (delta)^2 and its negation are formed in T, the division by the Float64 2 sigma^2 promotes
    const double logq = div_by_const((double)(-(delta * delta)), den, rden) - logc;   // == (-(d*d)) / den - logc, bit for bit
    const double arg = ((double)dlogp + logq) - logq;
    const bool c_pos = arg >= 0.0, c_rng = arg >= -708.0, c_exp = exp_core_f64(arg, T) > u;
    return c_pos | (c_rng & c_exp);
}

// Filter.  Inputs: dlogp (exact, f64) and k = the top 12 bits of u's 52-bit significand -- the bits the step's normal
// draw supplies (spec v5) --, so k 2^-12 <= u < (k+1) 2^-12 with both ends exact floats.  Error budget of the estimate
// ex = v_exp_f32(log2e * float(dlogp)) against the spec's exp(arg), for -17 <= dlogp < 1e-12 (relative):
//   arg vs dlogp      arg = fl(fl(dlogp + logq) - logq), |arg - dlogp| <= 2^-53 (2|dlogp| + |logq|) with
//                     |logq| <= z^2/2 (1 + 2^-50) + |log(2 pi s^2)/2| <= 37 + 231 (|z| <= 8.5, 1e-100 <= s <= 1e100)
//                                                                                                       < 4e-14
//   float(dlogp)      2^-24 * 17                                                                          1.1e-6
//   * log2e (float)   constant 1.3e-8 rel + product rounding 6e-8, times |y| <= 24.6, times ln 2          1.3e-6
//   v_exp_f32         1 ulp by the ISA; amc_selftest_accept_filter measures it exhaustively               < 5e-7
//   spec exp vs exp   2 ulp f64                                                                            4e-16
//   * (1 -+ eps)      one float rounding                                                                    6e-8
// total < 3.1e-6; eps = 2^-16 = 1.5e-5 leaves a factor 5.  Outside the range: dlogp > 1e-12 -> arg > 0 -> accept;
// dlogp < -17 -> the clamped estimate e^-17 is an upper bound only, and serves as one: its "lower bound minus one" is
// negative, so it can never claim an accept.  dlogp >= 0 gives ex >= 1, whose upper bound 4096 (1 + eps) exceeds every
// k, so it can never claim a reject.  NaN compares false everywhere -> undecided.
// Undecided when u's cell touches the interval (~1.2e-4 per chain-step, ~1.5 % of wave-steps); then the whole wave
// forms the accept draw and takes accept_exact.
#define AMC_FILTER_EPS 0x1.0p-16f
// The three primitive comparisons of one chain; the decision masks are formed from their ballots on the scalar unit
// (a ballot of a COMPOUND bool goes through a 0/1 VGPR and a second compare).  The sign test uses the float t:
// t > 2e-12 implies dlogp > 1e-12 with room to spare (t = RN(dlogp), relative 6e-8); a dlogp that underflows to
// t = 0 simply is not settled by its sign.
struct FilterCmp {
    bool pos, lo, hi;
};

__device__ __forceinline__ FilterCmp accept_filter(real_t dlogp, uint32_t k)
{
    const float t = (float)dlogp;
    const float ex = __builtin_amdgcn_exp2f(__builtin_fmaxf(t, -17.0f) * 0x1.715476p+0f);
    const float kf = (float)k;                                             // exact: k < 2^12
    constexpr float SCALE = 4096.0f;
    // lower / upper bound of exp(arg) 2^12, the lower one already minus 1: one rounding each (in the budget)
    const float lo1 = __builtin_fmaf(ex, (1.0f - AMC_FILTER_EPS) * SCALE, -1.0f);
    const float hi = ex * ((1.0f + AMC_FILTER_EPS) * SCALE);
    FilterCmp c;
    c.pos = t > 2e-12f;                 // arg > 0: accept whatever u is
    c.lo = lo1 > kf;                    // exp(arg) > (k+1) 2^-12 > u   (never true below -17: lo1 < 0)
    c.hi = hi < kf;                     // exp(arg) < k 2^-12 <= u      (never true for arg >= 0: hi > 4096)
    return c;
}

// The same filter for a decision whose ARGUMENT is known in full -- the script-defined proposals below form arg = (dlogp + logq_b) -
// logq_f in the reference's operations, nothing cancels -- : what the filter saves there is exp(arg) in Float64 and the accept draw
// (a second Philox call per pair and step), for all but the ~1.5 % of wave-steps it leaves open.  pos is the exact comparison;
// t = RN_f32(arg) moves the estimate's argument by at most 17 * 2^-24 = 1.0e-6, inside AMC_FILTER_EPS with the 3.1e-6 of the estimate
// itself.  arg = NaN or -Inf: fmaxf returns -17, the filter rejects for k >= 1 -- as the exact form does (no comparison with a NaN
// holds, and below -708 the exact form rejects) -- and leaves k = 0 open.
__device__ __forceinline__ FilterCmp accept_filter_arg(double arg, uint32_t k)
{
    const float t = (float)arg;
    const float ex = __builtin_amdgcn_exp2f(__builtin_fmaxf(t, -17.0f) * 0x1.715476p+0f);
    const float kf = (float)k;
    constexpr float SCALE = 4096.0f;
    const float lo1 = __builtin_fmaf(ex, (1.0f - AMC_FILTER_EPS) * SCALE, -1.0f);
    const float hi = ex * ((1.0f + AMC_FILTER_EPS) * SCALE);
    FilterCmp c;
    c.pos = arg >= 0.0;
    c.lo = lo1 > kf;
    c.hi = hi < kf;
    return c;
}
// the reference-ordered decision from arg and the full uniform: alpha = min(1, exp(arg)) > u (metropolis.jl:183-185)
__device__ __forceinline__ bool accept_exact_arg(double arg, double u, const double* T)
{
    const bool c_pos = arg >= 0.0, c_rng = arg >= -708.0, c_exp = exp_core_f64(arg, T) > u;
    return c_pos | (c_rng & c_exp);
}
// What one mc_step! of a script-defined proposal leaves for the decision: the proposed state, the state after perform_action_cached!
// (the revert: the inverted action applied to the proposed state) and the argument of the acceptance probability.
struct ScriptStep {
    real_t xn, xr;
    double arg;
};

#ifdef AMC_USER_SCALE
// One mc_step! with the state-dependent proposal width above, in the reference's operation order, up to the decision (mh_pair:
// the proposal ratio does not cancel, arg is formed in full and the filter takes it as it is, accept_filter_arg).
template <int POT>
__device__ __forceinline__ ScriptStep mh_scaled(real_t x, real_t beta, double sigma, double z, const double* T)
{
    const double TWO_PI = 0x1.921fb54442d18p+2;
    const double sc = sigma * user_scale(x, T);
    const double sc2 = sc * sc;
    const real_t delta = (real_t)__builtin_fma(sc, z, 0.0);          // 0.0 + sc*z, bit for bit (see propose)
    const double logq_f = ((double)(-(delta * delta))) / (2.0 * sc2) - log_f64(TWO_PI * sc2) / 2.0;
    const real_t e1 = potential<POT>(x, T);
    const real_t xn = x + delta;
    const real_t e2 = potential<POT>(xn, T);
    const real_t dlogp = ((-e2) * beta) - ((-e1) * beta);
    const real_t nd = -delta;
    const double scn = sigma * user_scale(xn, T);
    const double scn2 = scn * scn;
    const double logq_b = ((double)(-(nd * nd))) / (2.0 * scn2) - log_f64(TWO_PI * scn2) / 2.0;
    ScriptStep st;
    st.arg = ((double)dlogp + logq_b) - logq_f;
    st.xn = xn;
    st.xr = (real_t)(xn + nd);
    return st;
}
#endif

#ifdef AMC_USER_LOGQ
// One mc_step! with a script-defined proposal (see user_sample / user_logq), in the reference's operation order
// (metropolis.jl:176-190), up to the decision (mh_pair).
// A class of a mixed pool that IS the particle_1d Gaussian displacement -- its expressions are, character for character, the ones
// the host mirror writes the built-in policy out as: sample `sigma*z`, logq `-(delta*delta)/(2.0*(sigma*sigma)) -
// amc_log(6.283185307179586*(sigma*sigma))/2.0`, the displacement's own perform / invert (amc_rtc.hip sets AMC_CLASS_GAUSS_MASK, a bit
// per class) -- takes its density from the move's table row in the sweep (K > 1): den = 2.0*(sigma*sigma) and logc =
// amc_log(6.283185307179586*(sigma*sigma))/2.0 are the SAME operations on the same operands, formed once per parameter change
// (derive_move_params) instead of per lane and step, and the backward density of -delta is the forward one bit for bit
// ((-d)*(-d) == d*d).  What the lanes of such a class save: a log (fifty vector instructions) and the second density.
struct GaussRow {
    bool on;
    double den, logc;
};
__device__ __forceinline__ GaussRow gauss_row_off() { return GaussRow{false, 0.0, 0.0}; }
// k: the move key (move | class << 8); s_tab: the sweep's LDS copy of the parameter table (rows sigma, den, logc, cum, rden)
__device__ __forceinline__ GaussRow gauss_row_of(int k, const double* s_tab)
{
#if AMC_NCLASS > 1 && defined(AMC_CLASS_GAUSS_MASK)
    GaussRow g;
    g.on = (((unsigned)AMC_CLASS_GAUSS_MASK >> (k >> 8)) & 1u) != 0u;
    g.den = s_tab[1 * AMC_MAX_MOVES + (k & 0xFF)];
    g.logc = s_tab[2 * AMC_MAX_MOVES + (k & 0xFF)];
    return g;
#else
    (void)k; (void)s_tab;
    return gauss_row_off();
#endif
}

template <int POT, class S>
__device__ __forceinline__ ScriptStep mh_script(real_t x, real_t beta, S sigma, double z, const double* T, int k, const UserTheta& th,
                                                const GaussRow& g)
{
    const real_t delta = user_sample(z, x, sigma, T, k, th);             // :177 sample_action!
    double logq_f;                                                        // :178
    if (g.on) logq_f = ((double)(-(delta * delta))) / g.den - g.logc;
    else logq_f = user_logq(delta, x, sigma, T, k, th);
    const real_t e1 = potential<POT>(x, T);
    const real_t xn = user_perform(x, delta, T, k);                      // :179 perform_action!
    const real_t e2 = potential<POT>(xn, T);
    const real_t dlogp = ((-e2) * beta) - ((-e1) * beta);                // :180
    const real_t nd = user_invert(delta, xn, T, k);                      // :181 invert_action!
    double logq_b = logq_f;                                               // :182
    if (!g.on) logq_b = user_logq(nd, xn, sigma, T, k, th);
    ScriptStep st;
    st.arg = ((double)dlogp + logq_b) - logq_f;                          // :183
    st.xn = xn;
    st.xr = user_perform(xn, nd, T, k);                                  // :187 perform_action_cached!
    return st;
}
#endif

// What the exact decision of a chain needs from its move besides sigma: den = 2 sigma^2, RN(1/den), log(2 pi sigma^2)/2.
// K == 1: the pool's only move, wave-uniform scalars.  K > 1: read from the LDS copy of the move table by the chain's
// move index -- inside the undecided arm only, the common path reads sigma alone.
struct MoveExact {
    double dn, rd, lc;
};

// One mc_step! of both chains of a pair.  force_mask (wave-uniform, all ones or zero; tests) sends every wave through
// accept_exact.  acc_bits: bit 0 = even chain accepted, bit 8 = odd chain accepted (the step-log word's accept bits).
// The accept draw is not formed up front: the top 12 bits of u come from the normal draw (spec v5), which brackets u
// to 2^-12, and the decision is settled without the second Philox call unless exp(arg) falls into u's cell (~1.2e-4
// per chain-step, ~1.5 % of wave-steps).  `pu` / `have_pu` (wave-uniform): the accept draw, if the move pick of this
// step already needed it (pair_steps).
template <int POT, bool MULTI>
__device__ __forceinline__ void mh_pair(real2& xv, real_t b0, real_t b1, double sg0, double sg1, int k0, int k1,
                                        const double* s_tab, MoveExact m1, double z0, double z1, u32x4 pn, u32x4 pu,
                                        bool have_pu, u32x4 accept_ctr, uint32_t key0, uint32_t key1, const double* T,
                                        unsigned long long force_mask, uint32_t& acc_bits, unsigned long long& m0,
                                        unsigned long long& m1_out, const UserTheta& th1)
{
    const uint32_t a0_12 = spare_accept12(pn, 0), a1_12 = spare_accept12(pn, 1);
#if defined(AMC_USER_SCALE) || defined(AMC_USER_LOGQ)
    {
#ifdef AMC_USER_LOGQ
        // K == 1: the pool's only move, its parameters wave-uniform values read at the kernel's start; K > 1: the lane's move
        const int mk0 = user_move_key(MULTI ? k0 : 0), mk1 = user_move_key(MULTI ? k1 : 0);
        ScriptStep s0, s1;
        if (MULTI && AMC_SIGMA_MEMO) {          // the lane's sigma with its log from the block's table (SigmaArg)
            const SigmaArg a0 = {sg0, s_tab[(AMC_SIGMA_MEMO ? 5 : 0) * AMC_MAX_MOVES + (MULTI ? k0 : 0)]}, a1 = {sg1, s_tab[(AMC_SIGMA_MEMO ? 5 : 0) * AMC_MAX_MOVES + (MULTI ? k1 : 0)]};
            s0 = mh_script<POT>(xv.x, b0, a0, z0, T, mk0, user_theta_lds(mk0), gauss_row_of(mk0, s_tab));
            s1 = mh_script<POT>(xv.y, b1, a1, z1, T, mk1, user_theta_lds(mk1), gauss_row_of(mk1, s_tab));
        } else {
            s0 = mh_script<POT>(xv.x, b0, sg0, z0, T, mk0, MULTI ? user_theta_lds(mk0) : th1, MULTI ? gauss_row_of(mk0, s_tab) : gauss_row_off());
            s1 = mh_script<POT>(xv.y, b1, sg1, z1, T, mk1, MULTI ? user_theta_lds(mk1) : th1, MULTI ? gauss_row_of(mk1, s_tab) : gauss_row_off());
        }
#else
        const ScriptStep s0 = mh_scaled<POT>(xv.x, b0, sg0, z0, T), s1 = mh_scaled<POT>(xv.y, b1, sg1, z1, T);
#endif
        // round 5: the 12-bit bracket of u settles these decisions too (accept_filter_arg); exp(arg) in Float64 and the accept draw
        // are formed by the waves in which some lane's bracket leaves its decision open, for all their lanes
        const FilterCmp c0 = accept_filter_arg(s0.arg, a0_12), c1 = accept_filter_arg(s1.arg, a1_12);
        const unsigned long long acc0 = __builtin_amdgcn_ballot_w64(c0.pos | c0.lo), rej0 = __builtin_amdgcn_ballot_w64(c0.hi);
        const unsigned long long acc1 = __builtin_amdgcn_ballot_w64(c1.pos | c1.lo), rej1 = __builtin_amdgcn_ballot_w64(c1.hi);
        const unsigned long long undecided = __builtin_amdgcn_ballot_w64(true) & ~((acc0 | rej0) & (acc1 | rej1));
        bool a0 = c0.pos | c0.lo, a1 = c1.pos | c1.lo;
        if ((undecided | force_mask) != 0ull) {
            if (!have_pu) {
                asm volatile("" : "+v"(accept_ctr.z));       // pins the second Philox call inside this arm (no speculation)
                pu = philox4x32_10(accept_ctr, key0, key1);
            }
            a0 = accept_exact_arg(s0.arg, uniform_accept(a0_12, pu.x, pu.y), T);
            a1 = accept_exact_arg(s1.arg, uniform_accept(a1_12, pu.z, pu.w), T);
        }
        xv.x = a0 ? s0.xn : s0.xr;
        xv.y = a1 ? s1.xn : s1.xr;
        m0 = __builtin_amdgcn_ballot_w64(a0);
        m1_out = __builtin_amdgcn_ballot_w64(a1);
        acc_bits = (a0 ? 1u : 0u) | (a1 ? 0x100u : 0u);
        return;
    }
#endif
    const Proposal p0 = propose<POT>(xv.x, b0, sg0, z0, T), p1 = propose<POT>(xv.y, b1, sg1, z1, T);
    const real_t xr0 = p0.xn + (-p0.delta), xr1 = p1.xn + (-p1.delta);
    const FilterCmp c0 = accept_filter(p0.dlogp, a0_12), c1 = accept_filter(p1.dlogp, a1_12);
#define AMC_B(c) __builtin_amdgcn_ballot_w64(c)
    const unsigned long long acc0 = AMC_B(c0.pos) | AMC_B(c0.lo), rej0 = AMC_B(c0.hi);
    const unsigned long long acc1 = AMC_B(c1.pos) | AMC_B(c1.lo), rej1 = AMC_B(c1.hi);
    const unsigned long long undecided = AMC_B(true) & ~((acc0 | rej0) & (acc1 | rej1));
#undef AMC_B
    if ((undecided | force_mask) != 0ull) {
        // the reference-ordered arithmetic decides (it agrees with the filter wherever the filter decided)
        if (!have_pu) {
            asm volatile("" : "+v"(accept_ctr.z));       // pins the second Philox call inside this arm (no speculation)
            pu = philox4x32_10(accept_ctr, key0, key1);
        }
        MoveExact e0 = m1, e1 = m1;
        if (MULTI) {
            e0.dn = s_tab[AMC_MAX_MOVES + k0]; e0.lc = s_tab[2 * AMC_MAX_MOVES + k0]; e0.rd = s_tab[4 * AMC_MAX_MOVES + k0];
            e1.dn = s_tab[AMC_MAX_MOVES + k1]; e1.lc = s_tab[2 * AMC_MAX_MOVES + k1]; e1.rd = s_tab[4 * AMC_MAX_MOVES + k1];
        }
        const bool a0 = accept_exact(p0.delta, p0.dlogp, e0.dn, e0.rd, e0.lc, uniform_accept(a0_12, pu.x, pu.y), T);
        const bool a1 = accept_exact(p1.delta, p1.dlogp, e1.dn, e1.rd, e1.lc, uniform_accept(a1_12, pu.z, pu.w), T);
        m0 = __builtin_amdgcn_ballot_w64(a0);
        m1_out = __builtin_amdgcn_ballot_w64(a1);
        xv.x = a0 ? p0.xn : xr0;
        xv.y = a1 ? p1.xn : xr1;
        acc_bits = (a0 ? 1u : 0u) | (a1 ? 0x100u : 0u);
    } else {
        const bool a0 = c0.pos | c0.lo, a1 = c1.pos | c1.lo;
        m0 = acc0;
        m1_out = acc1;
        xv.x = a0 ? p0.xn : xr0;
        xv.y = a1 ? p1.xn : xr1;
        acc_bits = (a0 ? 1u : 0u) | (a1 ? 0x100u : 0u);
    }
}

// 16-byte loads / stores of a chain pair.  Stores use the sc1 (write-through) policy: the line does not stay dirty in
// the XCD's L2, so the kernel boundary does not pay for writing back up to 32 MB of dirty lines
// (MI355X_MICROARCH.md, store flavours / "boundary" row: + B / 6 TB/s for B dirty bytes; plain, nt and sc0 sc1 stores
// were re-measured: sc1 is the fastest).  The address lives on the SCALAR unit: a buffer resource at the block's (uniform)
// base plus the lane's constant byte offset threadIdx.x * 16 -- no per-lane 64-bit address arithmetic in the loop
// (4 VALU instructions per load/store pair otherwise).  aux 16 = sc1.
typedef uint32_t u32v4_t __attribute__((ext_vector_type(4)));

typedef uint32_t u32v2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ real2 load_pair_block(const real_t* block_base)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)block_base, 0, 0x7fffffff, 0x00020000);
    real2 d;
#ifdef AMC_STATE_F32
    const u32v2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, threadIdx.x * 8, 0, 0);     // a Float32 pair: 8 bytes per lane
    d.x = __uint_as_float(v.x);
    d.y = __uint_as_float(v.y);
#else
    const u32v4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 0);
    d.x = __longlong_as_double((long long)(((uint64_t)v.y << 32) | v.x));
    d.y = __longlong_as_double((long long)(((uint64_t)v.w << 32) | v.z));
#endif
    return d;
}

__device__ __forceinline__ void store_pair_block_writethrough(real_t* block_base, real2 d)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)block_base, 0, 0x7fffffff, 0x00020000);
#ifdef AMC_STATE_F32
    const u32v2_t v = {__float_as_uint(d.x), __float_as_uint(d.y)};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, threadIdx.x * 8, 0, 16);
#else
    const uint64_t a = (uint64_t)__double_as_longlong(d.x), b = (uint64_t)__double_as_longlong(d.y);
    const u32v4_t v = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, threadIdx.x * 16, 0, 16);
#endif
}
}  // namespace amc
