// amc_kernels.h -- HIP kernels of the many-chain Metropolis engine (gfx950 / CDNA4).
//
// Data layout in HBM: SoA, one f64 per chain
//   x[M_pad]                      chain positions (Particle.x); e is NOT stored: it is
//                                 potential(x) by construction (particle_1d.jl:13-15,33)
//   beta[M_pad]      (optional)   per-chain Particle.beta
//   acc[K][M_pad], tot[K-1][M_pad]  u32 Move.accepted_calls / total_calls (when kept; the last move's total_calls is the
//                                 step count minus the other moves': every chain takes the same number of steps)
//   ptab[PT_ROWS][AMC_MAX_MOVES]  per-move derived parameters (device-computed)
// One lane owns TWO adjacent chains (one global "pair"): 16-byte loads/stores, one
// Box-Muller and one accept-uniform Philox call serve both chains.
// All kernels are HBM-streaming in shape but f64-VALU-bound in practice; no MFMA.
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
// inverted action, at the new state; nothing cancels, every decision takes the reference-ordered arithmetic.
// ... and a script-defined ACTION (the reference's Action interface, src/metropolis.jl:15-119: perform_action!,
// invert_action!, perform_action_cached!; example/particle_1d/particle_1d.jl:30-40 are the displacement's methods), for a
// one-parameter action on the position:
//   AMC_USER_PERFORM(x, delta)    the position after perform_action!(system, action)        (displacement: x + delta)
//   AMC_USER_INVERT(delta, x)     the parameter of the inverted action, given the NEW state (displacement: -delta)
// perform_action_cached! (the revert) re-applies the inverted action, as the reference does (metropolis.jl:119,187).
struct UserTheta {          // a script-defined policy's parameters theta1 .. theta3 of one move (see AMC_USER_THETAS)
    double t1, t2, t3;
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
// d logq / d sigma: given for every class or for none (the host refuses the estimator then)
#ifdef AMC_USER_DLOGQ
#define AMC_USER_DLOGQ_0 AMC_USER_DLOGQ
#if AMC_NCLASS < 3
#define AMC_USER_DLOGQ_2 AMC_USER_DLOGQ
#endif
#if AMC_NCLASS < 4
#define AMC_USER_DLOGQ_3 AMC_USER_DLOGQ
#endif
#else
#define AMC_USER_DLOGQ_0(delta, x, sigma) __builtin_nan("")
#define AMC_USER_DLOGQ_1(delta, x, sigma) __builtin_nan("")
#define AMC_USER_DLOGQ_2(delta, x, sigma) __builtin_nan("")
#define AMC_USER_DLOGQ_3(delta, x, sigma) __builtin_nan("")
#endif
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
#define AMC_USER_THETAS(th)                                                                                            \
    const double theta0 = sigma, theta1 = (th).t1, theta2 = (th).t2, theta3 = (th).t3;                                  \
    (void)theta0; (void)theta1; (void)theta2; (void)theta3
#else
#define AMC_USER_THETAS(th) const double theta0 = sigma; (void)theta0; (void)th
#endif
// k: the move key (user_move_key).  Per lane, from the LDS copy staged by stage_user_theta:
__device__ __forceinline__ UserTheta user_theta_lds(int k)
{
    UserTheta th = {0.0, 0.0, 0.0};
#if AMC_NP > 1
    th.t1 = s_user_theta[0][k & 0xFF];
    if (AMC_NP > 2) th.t2 = s_user_theta[1][k & 0xFF];
    if (AMC_NP > 3) th.t3 = s_user_theta[2][k & 0xFF];
#else
    (void)k;
#endif
    return th;
}
// ... and of a move the whole wave shares (k wave-uniform), from the parameter table itself: scalar loads
__device__ __forceinline__ UserTheta user_theta_uniform(const double* ptab, int k)
{
    UserTheta th = {0.0, 0.0, 0.0};
#if AMC_NP > 1
    th.t1 = ptab[PT_THETA1 * AMC_MAX_MOVES + (k & 0xFF)];
    if (AMC_NP > 2) th.t2 = ptab[(PT_THETA1 + 1) * AMC_MAX_MOVES + (k & 0xFF)];
    if (AMC_NP > 3) th.t3 = ptab[(PT_THETA1 + 2) * AMC_MAX_MOVES + (k & 0xFF)];
#else
    (void)ptab; (void)k;
#endif
    return th;
}
__device__ __forceinline__ void stage_user_theta(const double* ptab)       // before a barrier the caller already has
{
#if AMC_NCLASS > 1
    for (int i = threadIdx.x; i < AMC_MAX_MOVES; i += AMC_BLOCK) s_user_class[i] = (int)ptab[PT_CLASS * AMC_MAX_MOVES + i];
#endif
#if AMC_NP > 1
    for (int i = threadIdx.x; i < (AMC_NP - 1) * AMC_MAX_MOVES; i += AMC_BLOCK)
        s_user_theta[i / AMC_MAX_MOVES][i % AMC_MAX_MOVES] = ptab[(PT_THETA1 + i / AMC_MAX_MOVES) * AMC_MAX_MOVES + i % AMC_MAX_MOVES];
#else
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
__device__ __forceinline__ real_t user_sample(double z, real_t x, double sigma, const double* amc_tables_, int k, const UserTheta& th)
{
    AMC_USER_THETAS(th);
#if AMC_NCLASS > 1
    AMC_BY_CLASS(k, real_t, AMC_USER_SAMPLE(z, x, sigma), AMC_USER_SAMPLE_1(z, x, sigma), AMC_USER_SAMPLE_2(z, x, sigma), AMC_USER_SAMPLE_3(z, x, sigma));
#else
    return (real_t)(AMC_USER_SAMPLE(z, x, sigma));    // Displacement.delta::T
#endif
}
__device__ __forceinline__ double user_logq(real_t delta, real_t x, double sigma, const double* amc_tables_, int k, const UserTheta& th)
{
    AMC_USER_THETAS(th);
#if AMC_NCLASS > 1
    AMC_BY_CLASS(k, double, AMC_USER_LOGQ(delta, x, sigma), AMC_USER_LOGQ_1(delta, x, sigma), AMC_USER_LOGQ_2(delta, x, sigma), AMC_USER_LOGQ_3(delta, x, sigma));
#else
    return (double)(AMC_USER_LOGQ(delta, x, sigma));
#endif
}
// grad log_proposal_density with respect to the parameters, d[p] = d logq / d theta_p
__device__ __forceinline__ void user_dlogq(real_t delta, real_t x, double sigma, const double* amc_tables_, int k, const UserTheta& th,
                                           double (&d)[AMC_NP])
{
    AMC_USER_THETAS(th);
#if AMC_NCLASS > 1
    d[0] = [&]() -> double {
        AMC_BY_CLASS(k, double, AMC_USER_DLOGQ_0(delta, x, sigma), AMC_USER_DLOGQ_1(delta, x, sigma), AMC_USER_DLOGQ_2(delta, x, sigma),
                     AMC_USER_DLOGQ_3(delta, x, sigma));
    }();
#elif defined(AMC_USER_DLOGQ)
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
    for (int p = 0; p < AMC_NP; ++p) d[p] = __builtin_nan("");     // the host refuses the estimator for such a handle
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
template <int POT>
__device__ __forceinline__ ScriptStep mh_script(real_t x, real_t beta, double sigma, double z, const double* T, int k, const UserTheta& th)
{
    const real_t delta = user_sample(z, x, sigma, T, k, th);             // :177 sample_action!
    const double logq_f = user_logq(delta, x, sigma, T, k, th);          // :178
    const real_t e1 = potential<POT>(x, T);
    const real_t xn = user_perform(x, delta, T, k);                      // :179 perform_action!
    const real_t e2 = potential<POT>(xn, T);
    const real_t dlogp = ((-e2) * beta) - ((-e1) * beta);                // :180
    const real_t nd = user_invert(delta, xn, T, k);                      // :181 invert_action!
    const double logq_b = user_logq(nd, xn, sigma, T, k, th);            // :182
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
        const ScriptStep s0 = mh_script<POT>(xv.x, b0, sg0, z0, T, mk0, MULTI ? user_theta_lds(mk0) : th1);
        const ScriptStep s1 = mh_script<POT>(xv.y, b1, sg1, z1, T, mk1, MULTI ? user_theta_lds(mk1) : th1);
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

// ---- reproducible cross-chain sums (amc_xsum.h): the device side ---------------------------
// A lane keeps one f64 accumulator per kind-Q column and two per kind-R column; a wave keeps the integer totals of what its
// lanes have flushed in LDS ("slots", written by its lane 0 only: no atomics); at the end of the kernel thread 0 of the block
// merges the block's four wave slots into one row of 64-bit words that the next level (the tail of the estimator kernel, or
// the host) adds up -- integers throughout, so no order of additions enters any result.
typedef unsigned long long xs_word;
enum { XS_ROW_Q = 2, XS_ROW_R = 6 };      // words per column of a block row: (lo, hi) / (top | flags << 32, k1.lo, k1.hi, k2.lo, k2.hi, 0)
#define AMC_XS_POISON_HI ((long long)0x8000000000000000ull)   // kind-Q row whose sum is NaN: hi = INT64_MIN, lo = 0

// Wave-wide totals of N 64-bit integers per lane, valid in EVERY lane afterwards.  Data-parallel-primitive moves instead of
// __shfl_down: a shuffle of a 64-bit value is two ds_bpermute_b32 through the LDS crossbar (~100 cycles each way, six rounds),
// a DPP move is a vector-unit instruction.  Rounds: row_shr 1, 2, 4, 8 inside the rows of 16 lanes (lanes without a source
// add 0), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15), then lane 31 into rows 2 and 3 (row_bcast31): lane 63
// holds the total, read back through the scalar unit.  The N values share the rounds (independent instructions back to back).
// (On the estimator kernel's tail and flushes: 1.1 us per launch of the 8 us the tail took with shuffles.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_move_i64(long long v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(unsigned long long)v, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROW_MASK, 0xF, false);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
template <int N, int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_round_i64(long long (&v)[N])
{
    long long o[N];
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = dpp_move_i64<CTRL, ROW_MASK>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += o[i];
}
template <int N>
__device__ __forceinline__ void wave_total_i64_dpp(long long (&v)[N])      // the plain form (round 4): kept as the selftest's reference
{
    dpp_round_i64<N, 0x111, 0xF>(v);      // row_shr:1
    dpp_round_i64<N, 0x112, 0xF>(v);      // row_shr:2
    dpp_round_i64<N, 0x114, 0xF>(v);      // row_shr:4
    dpp_round_i64<N, 0x118, 0xF>(v);      // row_shr:8
    dpp_round_i64<N, 0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
    dpp_round_i64<N, 0x143, 0xC>(v);      // row_bcast:31 into rows 2 and 3
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(unsigned long long)v[i], 63);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(v[i] >> 32), 63);
        v[i] = (long long)(((unsigned long long)hi << 32) | (unsigned long long)lo);
    }
}

// Wave totals by FOLDING (round 5).  The form above costs seven vector instructions per value and round (hipcc keeps a 64-bit
// DPP move as two v_mov_b32_dpp into zeroed registers plus the add: 93 instructions for two values), and every wave of a launch
// that forms callback sums pays it once per column.  gfx950 can swap half-waves and rows of two registers in ONE instruction
// (v_permlane32_swap: lanes 32..63 of the first operand with lanes 0..31 of the second; v_permlane16_swap: the odd rows of the
// first with the even rows of the second), so two values fold into one register whose halves (rows) hold one value each, with
// half the lanes left to add up -- a transposing reduction: 3 instructions per PAIR of values and step instead of 14.
//   fold32(a, b)   lanes 0..31: a[l] + a[l + 32]      lanes 32..63: b[l - 32] + b[l]
//   fold16(a, b)   row 0: a.row0 + a.row1   row 1: b.row0 + b.row1   row 2: a.row2 + a.row3   row 3: b.row2 + b.row3
// The last four steps, inside a row of 16 lanes, are an in-place scan (lane 15 of the row ends with the row's total): a 64-bit
// add whose first operand comes through DPP is v_add_co_u32_dpp + v_addc_co_u32_dpp, which hipcc does not form from C++
// (inline assembly; lanes without a source keep their value: bound_ctrl is off and the destination is the second operand).
// s_nop 1: a DPP operand must not be read within two wait states of the vector instruction that wrote it, and the compiler's
// hazard pass does not look inside an asm statement.
__device__ __forceinline__ long long fold32_i64(long long a, long long b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned int)(unsigned long long)a, (unsigned int)(unsigned long long)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned int)((unsigned long long)a >> 32), (unsigned int)((unsigned long long)b >> 32), false, false);
    return (long long)(((unsigned long long)hi[0] << 32) | lo[0]) + (long long)(((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ long long fold16_i64(long long a, long long b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned int)(unsigned long long)a, (unsigned int)(unsigned long long)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned int)((unsigned long long)a >> 32), (unsigned int)((unsigned long long)b >> 32), false, false);
    return (long long)(((unsigned long long)hi[0] << 32) | lo[0]) + (long long)(((unsigned long long)hi[1] << 32) | lo[1]);
}
#define AMC_ROW_STEP_I64(LO, HI, CTRL)                                                         \
    "s_nop 1\n\t"                                                                               \
    "v_add_co_u32_dpp " LO ", vcc, " LO ", " LO " " CTRL " row_mask:0xf bank_mask:0xf\n\t"      \
    "v_addc_co_u32_dpp " HI ", vcc, " HI ", " HI ", vcc " CTRL " row_mask:0xf bank_mask:0xf\n\t"
// lane 15 of every row: the total of the row's 16 lanes
__device__ __forceinline__ long long row_total_i64(long long v)
{
    unsigned int lo = (unsigned int)(unsigned long long)v, hi = (unsigned int)((unsigned long long)v >> 32);
    asm volatile(AMC_ROW_STEP_I64("%0", "%1", "row_shr:1") AMC_ROW_STEP_I64("%0", "%1", "row_shr:2")
                 AMC_ROW_STEP_I64("%0", "%1", "row_shr:4") AMC_ROW_STEP_I64("%0", "%1", "row_shr:8")
                 : "+v"(lo), "+v"(hi) : : "vcc");
    return (long long)(((unsigned long long)hi << 32) | lo);
}
// ... and lane 31 / lane 63: the totals of rows 0 + 1 / rows 2 + 3 (one more step: row_bcast:15 into rows 1 and 3)
__device__ __forceinline__ long long half_total_i64(long long v)
{
    unsigned int lo = (unsigned int)(unsigned long long)v, hi = (unsigned int)((unsigned long long)v >> 32);
    asm volatile(AMC_ROW_STEP_I64("%0", "%1", "row_shr:1") AMC_ROW_STEP_I64("%0", "%1", "row_shr:2")
                 AMC_ROW_STEP_I64("%0", "%1", "row_shr:4") AMC_ROW_STEP_I64("%0", "%1", "row_shr:8")
                 "s_nop 1\n\t"
                 "v_add_co_u32_dpp %0, vcc, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 : "+v"(lo), "+v"(hi) : : "vcc");
    return (long long)(((unsigned long long)hi << 32) | lo);
}
#undef AMC_ROW_STEP_I64
__device__ __forceinline__ long long read_lane_i64(long long v, int lane)
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(unsigned long long)v, lane);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), lane);
    return (long long)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
// Wave-wide totals of N 64-bit integers per lane, valid in EVERY lane afterwards (read back through the scalar unit): four values
// per register (fold32 twice, fold16, the row scan: rows hold v0, v2, v1, v3), a remaining pair in the halves of one, a single
// value folded onto itself.
template <int N>
__device__ __forceinline__ void wave_total_i64(long long (&v)[N])
{
    int i = 0;
#pragma unroll
    for (; i + 4 <= N; i += 4) {
        const long long u = row_total_i64(fold16_i64(fold32_i64(v[i], v[i + 1]), fold32_i64(v[i + 2], v[i + 3])));
        v[i] = read_lane_i64(u, 15); v[i + 2] = read_lane_i64(u, 31); v[i + 1] = read_lane_i64(u, 47); v[i + 3] = read_lane_i64(u, 63);
    }
    if (N - i == 3) {
        const long long u = row_total_i64(fold16_i64(fold32_i64(v[i], v[i + 1]), fold32_i64(v[i + 2], 0ll)));
        v[i] = read_lane_i64(u, 15); v[i + 2] = read_lane_i64(u, 31); v[i + 1] = read_lane_i64(u, 47);
    } else if (N - i == 2) {
        const long long u = half_total_i64(fold32_i64(v[i], v[i + 1]));
        v[i] = read_lane_i64(u, 31); v[i + 1] = read_lane_i64(u, 63);
    } else if (N - i == 1) {
        const long long u = half_total_i64(fold32_i64(v[i], 0ll));
        v[i] = read_lane_i64(u, 31);
    }
}
// the wave's largest value, valid in every lane (v_max_u32 through DPP: lanes without a source see 0)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#define AMC_MAX_STEP(CTRL, MASK)                                                                                       \
    {                                                                                                                  \
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, MASK, 0xF, false);                   \
        v = o > v ? o : v;                                                                                             \
    }
    AMC_MAX_STEP(0x111, 0xF) AMC_MAX_STEP(0x112, 0xF) AMC_MAX_STEP(0x114, 0xF) AMC_MAX_STEP(0x118, 0xF)
    AMC_MAX_STEP(0x142, 0xA) AMC_MAX_STEP(0x143, 0xC)
#undef AMC_MAX_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// The same for N 32-bit limbs whose wave totals fit 32 bits: ONE instruction per limb and round (the compiler folds the DPP
// move into the add: v_add_u32_dpp), where a 64-bit value costs two moves and a two-instruction add.  These kernels are bound
// by vector-instruction issue and every wave runs its flush once per launch: 27 instructions per wave-trip over round 3
// (PMC SQ_INSTS_VALU, 368 -> 395) came from the 64-bit rounds.
template <int N, int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_round_u32(uint32_t (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[i], CTRL, ROW_MASK, 0xF, false);
}
template <int N>
__device__ __forceinline__ void wave_total_u32(uint32_t (&v)[N])      // valid in every lane
{
    dpp_round_u32<N, 0x111, 0xF>(v);
    dpp_round_u32<N, 0x112, 0xF>(v);
    dpp_round_u32<N, 0x114, 0xF>(v);
    dpp_round_u32<N, 0x118, 0xF>(v);
    dpp_round_u32<N, 0x142, 0xA>(v);
    dpp_round_u32<N, 0x143, 0xC>(v);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (uint32_t)__builtin_amdgcn_readlane((int)v[i], 63);
}
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ void xs_store_q_row(xs_word* row, const xs::PartQ& p)
{
    const bool bad = p.flags != 0;
    row[0] = bad ? 0ull : (xs_word)p.k.lo;
    row[1] = bad ? (xs_word)AMC_XS_POISON_HI : (xs_word)p.k.hi;
}
__device__ __forceinline__ void xs_store_r_row(xs_word* row, const xs::PartR& p)
{
    row[0] = (xs_word)(uint32_t)p.top | ((xs_word)p.flags << 32);
    row[1] = (xs_word)p.k1.lo; row[2] = (xs_word)p.k1.hi;
    row[3] = (xs_word)p.k2.lo; row[4] = (xs_word)p.k2.hi;
    row[5] = 0ull;
}
__host__ __device__ inline xs::PartQ xs_load_q_row(const xs_word* row)
{
    xs::PartQ p;
    p.k.lo = (uint64_t)row[0]; p.k.hi = (int64_t)row[1];
    p.flags = 0u;
    if ((long long)row[1] == AMC_XS_POISON_HI && row[0] == 0ull) { p.flags = xs::XS_F_NAN; p.k = xs::i128{0, 0}; }
    return p;
}
__host__ __device__ inline xs::PartR xs_load_r_row(const xs_word* row)
{
    xs::PartR p;
    p.top = (int32_t)(uint32_t)row[0]; p.flags = (uint32_t)(row[0] >> 32);
    p.k1.lo = (uint64_t)row[1]; p.k1.hi = (int64_t)row[2];
    p.k2.lo = (uint64_t)row[3]; p.k2.hi = (int64_t)row[4];
    return p;
}

// Kind Q: NC lane accumulators s[] with wave-uniform constants cbits[] (bits of 1.5 * 2^(E + 52)).  A flush adds every lane's
// integer (bits(s) - cbits) into the wave's slot -- its low 32 bits and its high part as two separate 64-bit words, so no
// carry has to travel (k = hi 2^32 + lo) -- and restarts the accumulators at their constants.  An accumulator that has left
// its binade met a NaN or an infinity (or, never with the quanta of amc_xsum.h, too large a sum): the column is NaN.
// All 64 lanes take part (wave shuffles; lane 0 updates the slot): a flush never sits inside divergent control flow.  (64 lanes
// adding to one LDS address with atomics serialise: that form cost the fused time step 20 us per launch.)
struct QSlot {
    unsigned long long lo, hi;          // hi: two's complement
    unsigned int flags, pad_;
};
__device__ __forceinline__ void q_slot_clear(QSlot& s) { s.lo = 0ull; s.hi = 0ull; s.flags = 0u; s.pad_ = 0u; }
template <int NC>
__device__ __forceinline__ void q_flush(double (&s)[NC], const uint64_t (&cbits)[NC], QSlot* slot)
{
    const bool lane0 = (threadIdx.x & 63) == 0;
    long long k[NC];
    bool any_bad[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint64_t b = (uint64_t)__double_as_longlong(s[c]);
        const bool bad = ((b ^ cbits[c]) >> 52) != 0ull;
        k[c] = bad ? 0ll : (long long)(b - cbits[c]);              // |k| < 2^51 per lane: 2^57 per wave
        any_bad[c] = __builtin_amdgcn_ballot_w64(bad) != 0ull;
        s[c] = __longlong_as_double((long long)cbits[c]);
    }
    // |k| < 2^51: k + 2^51 is an unsigned integer of 52 bits, two limbs of 26 whose totals over 64 lanes fit 32 bits
    uint32_t limb[2 * NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const unsigned long long u = (unsigned long long)(k[c] + (1ll << 51));
        limb[2 * c] = (uint32_t)(u & 0x3FFFFFFull);
        limb[2 * c + 1] = (uint32_t)(u >> 26);
    }
    wave_total_u32<2 * NC>(limb);          // the columns share the rounds
#pragma unroll
    for (int c = 0; c < NC; ++c) k[c] = (long long)limb[2 * c] + ((long long)limb[2 * c + 1] << 26) - (64ll << 51);
    if (lane0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            slot[c].lo += (unsigned long long)(k[c] & 0xFFFFFFFFll);
            slot[c].hi += (unsigned long long)(k[c] >> 32);
            if (any_bad[c]) slot[c].flags |= (unsigned int)xs::XS_F_NAN;
        }
    }
}
// the same for lanes that hold the integer itself (|k| < 2^62: low 32 bits and high part go through the wave separately); all
// 64 lanes take part
__device__ __forceinline__ void q_flush_int(unsigned long long& acc, QSlot* slot)
{
    const long long k = (long long)acc;
    long long h[2] = {k & 0xFFFFFFFFll, k >> 32};
    wave_total_i64<2>(h);
    if ((threadIdx.x & 63) == 0) {
        slot->lo += (unsigned long long)h[0];
        slot->hi += (unsigned long long)h[1];
    }
    acc = 0ull;
}
// One acceptance ratio accepted / total (callback_acceptance, metropolis.jl:319-321: Int / Int -> Float64) as a multiple of
// 2^XS_E_RATIO, added to the lane's integer; a chain that never picked the move has 0 / 0 = NaN (`nan` is set; the
// division is then by 1).
__device__ __forceinline__ void ratio_add(unsigned long long& acc, bool& nan, uint32_t accepted, uint32_t total)
{
    const uint32_t den = total > 1u ? total : 1u;
    nan = nan | (total == 0u);
    const double q = (double)accepted / (double)den;
    const double t = xs::xs_c(xs::XS_E_RATIO) + __longlong_as_double(__double_as_longlong(q) | 1ll);
    acc += (unsigned long long)__double_as_longlong(t) - xs::xs_c_bits(xs::XS_E_RATIO);
}
// ... for counts beyond 32 bits (a handle whose counters have been carried into their 64-bit bases, counter_rebase_kernel): Int / Int
// of the reference, both below 2^53
__device__ __forceinline__ void ratio_add(unsigned long long& acc, bool& nan, unsigned long long accepted, unsigned long long total)
{
    const unsigned long long den = total > 1ull ? total : 1ull;
    nan = nan | (total == 0ull);
    const double q = (double)accepted / (double)den;
    const double t = xs::xs_c(xs::XS_E_RATIO) + __longlong_as_double(__double_as_longlong(q) | 1ll);
    acc += (unsigned long long)__double_as_longlong(t) - xs::xs_c_bits(xs::XS_E_RATIO);
}
__device__ __forceinline__ xs::PartQ q_slot_value(const QSlot& s)
{
    // k = hi 2^32 + lo
    const xs::i128 h = xs::i128_of((long long)s.hi);
    const xs::i128 hs = xs::i128{h.lo << 32, (int64_t)(((uint64_t)h.hi << 32) | (h.lo >> 32))};
    return xs::PartQ{xs::i128_add(hs, xs::i128{(uint64_t)s.lo, 0}), s.flags};
}

// Kind R: the running-top accumulators of NC columns.  top[] is wave-uniform -- every assignment comes from a readfirstlane --;
// a1[] / a2[] are the lane's 64-bit sums of the BIT PATTERNS of t = c1 + lsb1(v) and t2 = c2 + lsb1(r) (amc_xsum.h):
// n bits(c) + the sum of the multiples, n[] = summands since the last flush.  The two levels' constants and the bound are
// formed from top where they are used (a handful of scalar-unit integer operations): kept in registers across the sampling
// loops they would be ten more SGPRs per column in kernels that have none to spare -- spilled to VGPR lanes and fetched back
// with v_readlane, a vector-unit instruction, at every use (measured: +50 VALU instructions per trip).
template <int NC>
struct RLanes {
    unsigned long long a1[NC], a2[NC];
    int top[NC];
    int n[NC];
};
struct RLevel {
    double c1, c2, cap;
};
__device__ __forceinline__ RLevel r_level(int top)
{
    asm volatile("" : "+s"(top));          // not hoisted out of the caller's loop (see above)
    RLevel v;
    v.c1 = __longlong_as_double((long long)xs::xs_level_c_bits(top));
    v.c2 = __longlong_as_double((long long)xs::xs_level_c_bits(top - 1));
    v.cap = xs::xs_level_cap(top);
    return v;
}

template <int NC>
__device__ __forceinline__ void r_init(RLanes<NC>& L, xs::PartR* slot)
{
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        L.n[c] = 0;
        L.top[c] = xs::XS_LMIN;
        L.a1[c] = L.a2[c] = 0ull;
        if ((threadIdx.x & 63) == 0) slot[c] = xs::part_r_empty();
    }
}

// The rare arm of r_deposit: some lane holds a value the current top cannot take, or one that is not finite (or as good as).
// Every wave of a launch comes through here once per column -- its first deposit finds the column's level --, so the common
// case is kept short (round 5; the general form below cost ~135 instructions per column, a wave-wide maximum through six LDS
// permutes among them): the level a wave needs is a function of its largest |v|, and for everything finite |v| is monotone in
// the high word of its bit pattern -- one v_and, six v_max_u32 through DPP, and the level is formed on the scalar unit.  Only a
// wave that holds a NaN, an infinity or a finite value of 2^999 or more (high words that sort above every level's) classifies
// its lanes one by one.
template <int NC>
__device__ __forceinline__ int r_slow_classify(int c, double& v, xs::PartR* slot)
{
    int need = xs::XS_LMIN;
    uint32_t fl = 0u;
    {
        const uint64_t bits = (uint64_t)__double_as_longlong(v);
        const bool nan = ((bits >> 52) & 0x7FFull) == 0x7FFull && (bits & 0xFFFFFFFFFFFFFull) != 0ull;
        const int l = xs::xs_level_of(v);                  // > LMAX for infinities, NaN and finite |v| >= 2^999
        if (l > xs::XS_LMAX) {
            fl = nan ? xs::XS_F_NAN : ((bits >> 63) ? xs::XS_F_NINF : xs::XS_F_PINF);
            v = 0.0;                                       // the flags carry it
        } else {
            need = l;
        }
    }
    const bool lane0 = (threadIdx.x & 63) == 0;
    const uint32_t f_nan = __builtin_amdgcn_ballot_w64((fl & xs::XS_F_NAN) != 0u) ? xs::XS_F_NAN : 0u;
    const uint32_t f_pinf = __builtin_amdgcn_ballot_w64((fl & xs::XS_F_PINF) != 0u) ? xs::XS_F_PINF : 0u;
    const uint32_t f_ninf = __builtin_amdgcn_ballot_w64((fl & xs::XS_F_NINF) != 0u) ? xs::XS_F_NINF : 0u;
    if (lane0) slot[c].flags |= f_nan | f_pinf | f_ninf;
    return wave_max_i32(need);
}
template <int NC>
__device__ __forceinline__ void r_slow(RLanes<NC>& L, int c, double& v, xs::PartR* slot)
{
    const bool lane0 = (threadIdx.x & 63) == 0;
    const int be_max = (int)(wave_max_u32((uint32_t)((uint64_t)__double_as_longlong(v) >> 32) & 0x7FFFFFFFu) >> 20);
    const int need = be_max >= (int)xs::XS_BE_BEYOND ? r_slow_classify<NC>(c, v, slot) : xs::xs_level_of_exponent(be_max);
    if (need > L.top[c]) {
        // one level up the level-1 multiples ARE the new level-2 multiples; further up nothing of what was taken so far is
        // as large as half a quantum of the new lower level.  (n summands are on the books: n times the new constants' bits.)
        const unsigned long long n = (unsigned long long)(unsigned)L.n[c];
        const unsigned long long k1 = L.a1[c] - n * xs::xs_level_c_bits(L.top[c]);
        const bool one_up = need - L.top[c] == 1;
        L.top[c] = need;
        L.a1[c] = n * xs::xs_level_c_bits(need);
        L.a2[c] = (one_up ? k1 : 0ull) + n * xs::xs_level_c_bits(need - 1);
        if (lane0) xs::part_r_raise(slot[c], need);
    }
}

// One summand per lane (every lane of the wave is in the call: pass 0.0 where there is nothing to add).
template <int NC>
__device__ __forceinline__ void r_deposit(RLanes<NC>& L, int c, double v, xs::PartR* slot)
{
    // |v| < 2^(50 top + 49) (NaN compares false: it takes the rare arm like infinities and finite values of 2^999 or more)
    if (__builtin_amdgcn_ballot_w64(!(__builtin_fabs(v) < r_level(L.top[c]).cap)) != 0ull) r_slow(L, c, v, slot);
    const RLevel lv = r_level(L.top[c]);
    const double v1 = __longlong_as_double(__double_as_longlong(v) | 1ll);
    const double t = lv.c1 + v1;
    const double r = v1 - (t - lv.c1);
    const double t2 = lv.c2 + __longlong_as_double(__double_as_longlong(r) | 1ll);
    L.a1[c] += (unsigned long long)__double_as_longlong(t);
    L.a2[c] += (unsigned long long)__double_as_longlong(t2);
    L.n[c] += 1;
}

template <int NC>
__device__ __forceinline__ void r_flush(RLanes<NC>& L, xs::PartR* slot)
{
    const bool lane0 = (threadIdx.x & 63) == 0;
    // per column: the two levels' multiples, |.| < n 2^49 < 2^63 (the 64-bit arithmetic modulo 2^64 holds them); a wave's 64 lanes
    // need up to 6 more bits.  Few summands (a launch over 1e7 chains gives a lane 10 to 16): |k| < 2^56, the 64 lanes' sum fits
    // 64 bits, and ALL columns' multiples go through the wave together (wave_total_i64: four values per folded register).
    // Otherwise low 32 bits and high parts travel separately, a column at a time.
    int n_max = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) n_max = L.n[c] > n_max ? L.n[c] : n_max;
    auto multiples = [&](int c, long long& k1, long long& k2) {
        const unsigned long long n = (unsigned long long)(unsigned)L.n[c];
        k1 = (long long)(L.a1[c] - n * xs::xs_level_c_bits(L.top[c]));
        k2 = (long long)(L.a2[c] - n * xs::xs_level_c_bits(L.top[c] - 1));
        L.a1[c] = L.a2[c] = 0ull;
        L.n[c] = 0;
    };
    if (n_max <= 128) {
#pragma unroll
        for (int c = 0; c + 2 <= NC; c += 2) {            // two columns per folded register
            long long v[4];
            multiples(c, v[0], v[1]);
            multiples(c + 1, v[2], v[3]);
            wave_total_i64<4>(v);
            if (lane0) {
                slot[c].k1 = xs::i128_add(slot[c].k1, xs::i128_of(v[0]));
                slot[c].k2 = xs::i128_add(slot[c].k2, xs::i128_of(v[1]));
                slot[c + 1].k1 = xs::i128_add(slot[c + 1].k1, xs::i128_of(v[2]));
                slot[c + 1].k2 = xs::i128_add(slot[c + 1].k2, xs::i128_of(v[3]));
            }
        }
        if (NC & 1) {
            long long v[2];
            multiples(NC - 1, v[0], v[1]);
            wave_total_i64<2>(v);
            if (lane0) {
                slot[NC - 1].k1 = xs::i128_add(slot[NC - 1].k1, xs::i128_of(v[0]));
                slot[NC - 1].k2 = xs::i128_add(slot[NC - 1].k2, xs::i128_of(v[1]));
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            long long k1, k2;
            multiples(c, k1, k2);
            long long v[4] = {k1 & 0xFFFFFFFFll, k1 >> 32, k2 & 0xFFFFFFFFll, k2 >> 32};
            wave_total_i64<4>(v);
            if (lane0) {
                slot[c].k1 = xs::i128_add(slot[c].k1, xs::i128_add(xs::i128_shl(xs::i128_of(v[1]), 32), xs::i128_of(v[0])));
                slot[c].k2 = xs::i128_add(slot[c].k2, xs::i128_add(xs::i128_shl(xs::i128_of(v[3]), 32), xs::i128_of(v[2])));
            }
        }
    }
}

// The block's row of a kind-R / kind-Q column from its wave slots slots[wave][NC] (thread 0, after a barrier).
template <int NC>
__device__ __forceinline__ xs::PartR r_block_total(const xs::PartR (*slots)[NC], int c)
{
    xs::PartR t = slots[0][c];
    for (int w = 1; w < AMC_BLOCK / 64; ++w) xs::part_r_merge(t, slots[w][c]);
    return t;
}
template <int NC>
__device__ __forceinline__ xs::PartQ q_block_total(const QSlot (*slots)[NC], int c)
{
    xs::PartQ t = q_slot_value(slots[0][c]);
    for (int w = 1; w < AMC_BLOCK / 64; ++w) {
        const xs::PartQ o = q_slot_value(slots[w][c]);
        t.k = xs::i128_add(t.k, o.k);
        t.flags |= o.flags;
    }
    return t;
}

// The callback sums a REDUCE launch forms of the state it stores (callback_energy particle_1d.jl:68-70, the moments of
// test/distribution_test.jl:36-37): kind-R columns sum e, sum x, sum x^2 -- with U = x^2 in Float64 sum x^2 IS sum e (the same
// products), and the row's third column is a copy of the first.  Float64 sums whatever the state type.
// A column's SUMMANDS are the chain PAIRS' sums (global chains 2p and 2p + 1, the pair a lane owns): fl(e_2p + e_2p+1),
// fl(x_2p + x_2p+1), fl(fl(x_2p^2) + fl(x_2p+1^2)) -- one of the orders in which the reference's `mean` may add, fixed by the global
// chain ids alone (shards begin at even ids), and half the work of taking the chains one by one (a lone last chain is
// its own summand).
enum { RED_COLS = 3, RED_ROW_COUNT = RED_COLS * XS_ROW_R, RED_ROW_SLOT = RED_ROW_COUNT + 1, RED_ROW_WORDS = RED_ROW_COUNT + 2 };
// The block's row in its COMPACT form (round 5): ONE 64-byte line per block on the link to the host instead of three.  A launch
// whose lanes see at most RED_COMPACT_TRIPS summands per column (the host knows: trips per lane) has block totals below 2^62 --
// 256 lanes x 32 x 2^49 -- so a column is two 64-bit words, and the three columns' tops and flags share a word:
//   word 0       16 bits per column: (top + 128) | flags << 8
//   word 1 + 2c  k1 of column c        word 2 + 2c  k2 of column c
//   word 7       the pool-wide accepted slot (as in the wide row), else 0
// The count needs no word: the rows of a launch cover the handle's chains, the host knows their number.
enum { RED_COMPACT_WORDS = 8, RED_COMPACT_SLOT = 7, RED_COMPACT_TRIPS = 32 };
// Which sums a launch forms (SweepArgs.red_cols; amc_set_reduce_columns): callback_energy needs sum e alone, the moments of
// test/distribution_test.jl sum x and sum x^2 -- a deposit costs nine vector instructions per trip and column, so what nobody
// asked for is not formed (its record stays empty).
enum { RED_WANT_E = 1, RED_WANT_X = 2, RED_WANT_XX = 4, RED_WANT_ALL = 7 };
template <int POT>
struct RedCols {
    static constexpr bool X2_IS_E = POT == POT_HARMONIC && sizeof(real_t) == 8;
    static constexpr int NC = X2_IS_E ? 2 : 3;
};
__host__ __device__ inline xs::PartR xs_load_compact_row(const xs_word* row, int c)
{
    xs::PartR p;
    const unsigned int f = (unsigned int)(row[0] >> (16 * c)) & 0xFFFFu;
    p.top = (int32_t)(f & 0xFFu) - 128;
    p.flags = f >> 8;
    p.k1 = xs::i128_of((long long)row[1 + 2 * c]);
    p.k2 = xs::i128_of((long long)row[2 + 2 * c]);
    return p;
}

// cols: RED_WANT_* bits (wave-uniform: a kernel argument)
template <int POT>
__device__ __forceinline__ void red_add_pair(RLanes<RedCols<POT>::NC>& L, real2 xv, bool v0, bool v1, const double* s_math,
                                             xs::PartR* slot, int cols)
{
    const double x0 = v0 ? (double)xv.x : 0.0, x1 = v1 ? (double)xv.y : 0.0;
    if (cols & (RedCols<POT>::X2_IS_E ? (RED_WANT_E | RED_WANT_XX) : RED_WANT_E)) {
        const double e0 = v0 ? (double)potential<POT>(xv.x, s_math) : 0.0, e1 = v1 ? (double)potential<POT>(xv.y, s_math) : 0.0;
        r_deposit(L, 0, e0 + e1, slot);
    }
    if (cols & RED_WANT_X) r_deposit(L, 1, x0 + x1, slot);
    if (!RedCols<POT>::X2_IS_E && (cols & RED_WANT_XX)) r_deposit(L, 2, x0 * x0 + x1 * x1, slot);
}

// End of the launch: the block's row.
// compact (block-uniform; see RED_COMPACT_WORDS): the lanes' integers go through the wave (wave_total_i64), lane 0 leaves the
// wave's totals and tops in LDS, one barrier, and threads 0 .. 6 of the block form one word of the row each -- the tops' maximum,
// the waves' totals brought to it (amc_xsum.h: one level up a k1 total is the k2 total, further up nothing is left) -- and store
// it: seven lanes of one instruction, one 64-byte write.  No wave slot is read-modified-written and nothing is 128 bits wide;
// the slots only carry the flags of the rare arm.  (Round 4's form, kept as the wide form below, cost the K = 2 launch 4.2 us
// at 1e7 chains, its three-line row 1.4 us of them.)
// wide: flush into the wave slots, thread c merges the block's slots of column c (any number of summands, mid-launch flushes).
template <int POT>
__device__ __forceinline__ void red_finish(RLanes<RedCols<POT>::NC>& L, xs::PartR (*slots)[RedCols<POT>::NC], xs_word* row, bool compact,
                                           int cols)
{
    constexpr int NC = RedCols<POT>::NC;
    if (compact) {
        __shared__ long long s_fin_k[AMC_BLOCK / 64][2 * NC];
        __shared__ int s_fin_top[AMC_BLOCK / 64][NC];
        long long k[2 * NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned long long n = (unsigned long long)(unsigned)L.n[c];
            k[2 * c] = (long long)(L.a1[c] - n * xs::xs_level_c_bits(L.top[c]));
            k[2 * c + 1] = (long long)(L.a2[c] - n * xs::xs_level_c_bits(L.top[c] - 1));
        }
        // (the common request is sum e alone -- callback_energy --: its two integers travel by themselves, the columns nobody
        // deposited into are zero without a sum)
        if ((cols & ~(RedCols<POT>::X2_IS_E ? (RED_WANT_E | RED_WANT_XX) : RED_WANT_E)) == 0) {
            long long k0[2] = {k[0], k[1]};
            wave_total_i64<2>(k0);
            k[0] = k0[0]; k[1] = k0[1];
        } else {
            wave_total_i64<2 * NC>(k);
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                s_fin_k[threadIdx.x >> 6][2 * c] = k[2 * c];
                s_fin_k[threadIdx.x >> 6][2 * c + 1] = k[2 * c + 1];
                s_fin_top[threadIdx.x >> 6][c] = L.top[c];
            }
        }
        __syncthreads();
        const int t = (int)threadIdx.x;
        if (t < 1 + 2 * RED_COLS) {
            xs_word w = 0ull;
            // the lane column behind row column rc (sum x^2 = sum e where they are the same sums)
            auto lane_col = [](int rc) { return (RedCols<POT>::X2_IS_E && rc == 2) ? 0 : rc; };
            auto top_of = [&](int c) {
                int T = s_fin_top[0][c];
                for (int wv = 1; wv < AMC_BLOCK / 64; ++wv) T = s_fin_top[wv][c] > T ? s_fin_top[wv][c] : T;
                return T;
            };
            if (t == 0) {
                for (int rc = 0; rc < RED_COLS; ++rc) {
                    const int c = lane_col(rc);
                    uint32_t fl = 0u;
                    for (int wv = 0; wv < AMC_BLOCK / 64; ++wv) fl |= slots[wv][c].flags;
                    w |= (xs_word)((uint32_t)(top_of(c) + 128) | (fl << 8)) << (16 * rc);
                }
            } else {
                const int c = lane_col((t - 1) >> 1), second = (t - 1) & 1;
                const int T = top_of(c);
                long long sum = 0;
                for (int wv = 0; wv < AMC_BLOCK / 64; ++wv) {
                    const int d = T - s_fin_top[wv][c];
                    if (d == 0) sum += s_fin_k[wv][2 * c + second];
                    else if (d == 1 && second) sum += s_fin_k[wv][2 * c];
                }
                w = (xs_word)sum;
            }
            row[t] = w;
        }
        return;
    }
    r_flush(L, slots[threadIdx.x >> 6]);
    __syncthreads();                                // the slots are visible to the threads that compose the row
    // the row goes to pinned host memory: composed in LDS, stored by ONE wave instruction (consecutive words: three 64-byte
    // writes on the link instead of twenty 8-byte ones)
    __shared__ xs_word s_row[RED_ROW_WORDS];
    if (threadIdx.x < RED_COLS) {          // thread c: column c (sum x^2 = sum e where they are the same sums)
        const int c = (RedCols<POT>::X2_IS_E && threadIdx.x == 2) ? 0 : (int)threadIdx.x;
        xs_store_r_row(s_row + threadIdx.x * XS_ROW_R, r_block_total<NC>(slots, c));
        if (threadIdx.x == 0) s_row[RED_ROW_COUNT] = 0ull;          // (unused: the host knows the chains a launch covers)
    }
    __syncthreads();
    if (threadIdx.x <= RED_ROW_COUNT) row[threadIdx.x] = s_row[threadIdx.x];
}

struct SweepArgs {
    real_t* x;
    const real_t* beta_arr;       // nullptr unless per-chain beta
    uint8_t* log;                 // [log_depth][m_stride] per-chain step log (LOG launches), else nullptr
    const double* ptab;           // [PT_ROWS][AMC_MAX_MOVES]
    const uint8_t* pick_tab;      // [AMC_PICK_CELLS] move pick by the 12 leading bits of the pick uniform (K > 1), see prepare_pick_kernel
    unsigned long long* acc_total;  // pool-wide accepted count (K == 1)
    int64_t n_chains;             // local chains
    int64_t m_stride;             // padded length of per-chain arrays
    uint64_t pair0;               // global pair id of local pair 0 (= chain_offset / 2)
    uint64_t t0;                  // step index of the first MH step of this launch
    int32_t n_steps;              // MH steps fused in this launch
    int32_t n_moves;
    uint32_t key0, key1;
    double beta;
    xs_word* red_partials;        // REDUCE launches: [grid][red_stride] block rows, pinned host memory: three kind-R columns
                                  // (sum e, sum x, sum x^2: XS_ROW_R words each), then as doubles the count and this block's
                                  // pool-wide accepted slot after the launch (RED_ROW_COUNT, RED_ROW_SLOT)
    int32_t red_stride;           // words per row: RED_ROW_WORDS, or RED_COMPACT_WORDS for the compact form (red_finish)
    int32_t red_cols;             // RED_WANT_* bits: the sums this launch forms
    int32_t log_pos;              // row of the step log the first step of this launch writes
    int32_t exact_accept;         // != 0: skip the accept filter, every decision by accept_exact (tests; AMC_EXACT_ACCEPT)
    int32_t n_slots;              // length of acc_total (launches of different grids share it)
};

// The Philox result every MH step of a pair needs -- its normal draw: a pure function of (seed, pair, step), so it can
// be formed before the pair's state has arrived from memory.  (Its spare bits lead the accept and pick uniforms; the
// accept draw itself is formed only where those 12-bit brackets leave something open.)
struct StepDraws {
    u32x4 normal;
};

__device__ __forceinline__ StepDraws step_draws(const SweepArgs& a, uint64_t pair, uint64_t t)
{
    StepDraws d;
    d.normal = philox4x32_10(draw_counter(pair, t, DRAW_NORMAL, STREAM_METROPOLIS), a.key0, a.key1);
    return d;
}

// rand(rng, Categorical(weights)) (metropolis.jl:206) from the 12 leading bits of the pick uniform.  The walk of
// Distributions.jl's sampler, #(cum[i] <= r), is monotone in r, so every r of the cell [c, c+1) 2^-12 picks the same
// move unless a cumulative weight lies inside the cell: AMC_PICK_CELLS bytes, entry = the move index, or
// AMC_PICK_OPEN for the <= K-1 cells that hold a boundary (then the accept draw supplies 24 more bits and the walk
// runs on the 36-bit uniform).  Built on the device from the same cum[] the walk uses (prepare_pick_kernel).
#define AMC_PICK_CELLS 4096
#define AMC_PICK_OPEN 0xFFu

// Copy the pick table into this block's LDS: 256 threads x 16 bytes (visible after the block's next barrier).
__device__ __forceinline__ void stage_pick_table(uint8_t* lds, const uint8_t* tab)
{
    for (int i = threadIdx.x; i < AMC_PICK_CELLS / 16; i += AMC_BLOCK)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(tab)[i];
}

__device__ __forceinline__ int categorical_walk(const double* s_tab, int K, double r)
{
    int k = 0;
    for (int i = 0; i < K - 1; ++i) k += (s_tab[3 * AMC_MAX_MOVES + i] <= r) ? 1 : 0;     // cp = w1; while cp <= r && i < K: cp += w[i+1]
    return k;
}

// Per-chain Move.accepted_calls / total_calls (metropolis.jl:208-209) are not read-modify-written by the sweep:
// every MH step appends (move index << 1) | accepted per chain to a step log, and fold_log_kernel adds a batch of log
// rows into the counters when somebody asks for them or the log is full.  The counters themselves cost 16 K bytes of HBM
// traffic per chain and pass (every line of every move's array is touched); the log costs 1 byte per chain and step --
// and half a byte where the move index fits three bits (K <= AMC_PACKED_LOG_MOVES): the two chains of a lane then share
// ONE byte, chain 0 in the low nibble (rows of m_stride / 2 bytes, 64 contiguous bytes per wave); otherwise one byte per
// chain (rows of m_stride bytes, 128 per wave).  What the callback's fold reads is halved with it.
#define AMC_PACKED_LOG_MOVES 8
#define AMC_LOG_NONE 0
#define AMC_LOG_PACKED 1      // K <= AMC_PACKED_LOG_MOVES
#define AMC_LOG_BYTES 2       // K > AMC_PACKED_LOG_MOVES
template <int LOG>
__device__ __forceinline__ void store_log_pair(const SweepArgs& a, int row, int64_t p, uint32_t word)
{
    // write-through (sc1) like the positions: plain stores would stay dirty in the XCDs' L2s until the
    // kernel boundary writes them back (K = 2 sweep 35.8 -> 35.2 us per launch incl. amortised folds, same-box A/B, round 3).
    // The address lives on the scalar unit: p - threadIdx.x is block-uniform.  word: chain 0 in bits 0..7, chain 1 in 8..15.
    // (The form is a template argument: chosen from a.n_moves at run time the two stores cost the K = 2 launches 0.7-3 %.)
    if (LOG == AMC_LOG_PACKED) {
        uint8_t* base = a.log + (int64_t)row * (a.m_stride >> 1) + (p - (int64_t)threadIdx.x);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(word | (word >> 4)), r, threadIdx.x, 0, 16);
    } else {
        uint8_t* base = a.log + (int64_t)row * a.m_stride + 2 * (p - (int64_t)threadIdx.x);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b16((uint16_t)word, r, threadIdx.x * 2, 0, 16);
    }
}

// `n_steps` fused MH steps of one chain pair held in registers (the body of mc_sweep!, metropolis.jl:205-210).
// PRE: the draws of the (single) step were formed ahead by the caller and come in `pre`.
// LOG: the step-log word of the pair; SINGLE launches hand it back in `log_word` (the caller stores it together
// with x), multi-step launches store one word per step right away.
template <int POT, bool MULTI, int LOG, bool SINGLE, bool PRE = false>
__device__ __forceinline__ void pair_steps(const SweepArgs& a, real2& xv, real_t b0, real_t b1, uint64_t pair,
                                           int64_t p, bool v0, bool v1, const double* s_tab, const uint8_t* s_pick,
                                           const double* s_math, double sigma1, double den1, double rden1, double logc1,
                                           unsigned long long& wave_acc, uint32_t& log_word,
                                           const StepDraws* pre = nullptr, const MathK& mk = math_k_literal(),
                                           const UserTheta& th1 = UserTheta{0.0, 0.0, 0.0})
{
    static_assert(!PRE || SINGLE, "pre-formed draws cover exactly one step");
    const int K = a.n_moves;
    const int n_steps = SINGLE ? 1 : a.n_steps;      // SINGLE: the sweepstep = 1 launch, straight-line code
    const unsigned long long force_mask = a.exact_accept ? ~0ull : 0ull;
    const MoveExact m1 = {den1, rden1, logc1};
    for (int s = 0; s < n_steps; ++s) {
        const uint64_t t = a.t0 + (uint64_t)s;
        const StepDraws dr = PRE ? *pre : step_draws(a, pair, t);
        const u32x4 accept_ctr = draw_counter(pair, t, DRAW_ACCEPT, STREAM_METROPOLIS);
        double sg0 = sigma1, sg1 = sigma1;
        int k0 = 0, k1 = 0;
        u32x4 pu = {0u, 0u, 0u, 0u};
        bool have_pu = false;                        // wave-uniform
        if (MULTI) {
            // rand(rng, Categorical(weights)) metropolis.jl:206
            const uint32_t q0 = spare_pick12(dr.normal, 0), q1 = spare_pick12(dr.normal, 1);
            k0 = s_pick[q0];
            k1 = s_pick[q1];
            const bool open = ((k0 | k1) & 0x80) != 0;
            if ((__builtin_amdgcn_ballot_w64(open) | force_mask) != 0ull) {
                // some chain's cell holds a cumulative weight: the accept draw supplies the pick's low 24 bits, and
                // every chain of the wave walks the full 36-bit uniform (equal to its table entry where that was closed)
                pu = philox4x32_10(accept_ctr, a.key0, a.key1);
                have_pu = true;
                k0 = categorical_walk(s_tab, K, uniform_pick(q0, pu.x));
                k1 = categorical_walk(s_tab, K, uniform_pick(q1, pu.z));
            }
            sg0 = s_tab[k0];
            sg1 = s_tab[k1];
        }
        double z0, z1;
        box_muller(dr.normal, z0, z1, s_math, mk);
        unsigned long long m0, m1m;
        uint32_t acc_bits;
        mh_pair<POT, MULTI>(xv, b0, b1, sg0, sg1, k0, k1, s_tab, m1, z0, z1, dr.normal, pu, have_pu, accept_ctr, a.key0,
                            a.key1, s_math, force_mask, acc_bits, m0, m1m, th1);
        // K == 1: wavefront-ballot accept mask -> one scalar popcount per chain slot (pool-wide total)
        if (!MULTI) wave_acc += __popcll(m0 & __builtin_amdgcn_ballot_w64(v0)) + __popcll(m1m & __builtin_amdgcn_ballot_w64(v1));
        if (LOG) {
            // Move.accepted_calls += accepted; Move.total_calls += 1 (metropolis.jl:208-209), deferred: see above
            log_word = acc_bits | ((uint32_t)k0 << 1) | ((uint32_t)k1 << 9);
            if (!SINGLE && v0) store_log_pair<LOG>(a, a.log_pos + s, p, log_word);
        }
    }
}

// K1: the sweep.  make_step!(::Metropolis) metropolis.jl:302-309 -> mc_sweep! :203-212.
// MULTI: K > 1 (categorical move pick, parameter table staged in LDS)
// LOG: per-chain counters are kept (always when K > 1): AMC_LOG_PACKED / AMC_LOG_BYTES, the step log's form (store_log_pair)
// BETA: per-chain beta array
// SINGLE: exactly one MH step per launch (the default sweepstep = 1 make_step!): no step loop
// REDUCE: also leave the callback sums of the state AFTER the sweep in red_partials (sum e, sum x, sum x^2, count;
//         and, pool-wide counter only, the accepted total), so a sweep that is followed by callback_energy /
//         callback_acceptance needs no second pass over x
template <int POT, bool MULTI, int LOG, bool BETA, bool SINGLE, bool REDUCE = false>
__global__ __launch_bounds__(AMC_BLOCK) void sweep_kernel(const SweepArgs a)
{
    // REDUCE with LOG (per-chain counters): rows carry the sums over x only; the acceptance ratios of the same
    // callback come from the fold of the step log that follows (fold_log_kernel<KS, true>)
    static_assert(!MULTI || LOG, "K > 1 always keeps per-chain counters");
    // the callback sums: reproducible (amc_xsum.h); the count of full trips lives on the scalar unit
    constexpr int RNC = RedCols<POT>::NC;
    RLanes<RNC> red;
    __shared__ xs::PartR s_red[REDUCE ? AMC_BLOCK / 64 : 1][RNC];
    if (REDUCE) r_init(red, s_red[threadIdx.x >> 6]);
    __shared__ double s_tab[MULTI ? 5 * AMC_MAX_MOVES : 1];
    __shared__ __attribute__((aligned(16))) uint8_t s_pick[MULTI ? AMC_PICK_CELLS : 16];
    __shared__ double s_math[TAB_DOUBLES];        // exp / log / sincospi tables, 4.4 KB
    const int K = a.n_moves;
    if (MULTI) {
        stage_pick_table(s_pick, a.pick_tab);
        for (int i = threadIdx.x; i < K; i += AMC_BLOCK) {
            s_tab[0 * AMC_MAX_MOVES + i] = a.ptab[PT_SIGMA * AMC_MAX_MOVES + i];
            s_tab[1 * AMC_MAX_MOVES + i] = a.ptab[PT_DEN * AMC_MAX_MOVES + i];
            s_tab[2 * AMC_MAX_MOVES + i] = a.ptab[PT_LOGC * AMC_MAX_MOVES + i];
            s_tab[3 * AMC_MAX_MOVES + i] = a.ptab[PT_CUM * AMC_MAX_MOVES + i];
            s_tab[4 * AMC_MAX_MOVES + i] = a.ptab[PT_RDEN * AMC_MAX_MOVES + i];
        }
        // visible to the block after the barrier that ends stage_math_tables below
    }
    // K == 1: wave-uniform scalars (s_load)
    const double sigma1 = a.ptab[PT_SIGMA * AMC_MAX_MOVES];
    const double den1 = a.ptab[PT_DEN * AMC_MAX_MOVES];
    const double logc1 = a.ptab[PT_LOGC * AMC_MAX_MOVES];
    const double rden1 = a.ptab[PT_RDEN * AMC_MAX_MOVES];
    UserTheta th1 = {0.0, 0.0, 0.0};
#ifdef AMC_USER_LOGQ
    if (!MULTI) th1 = user_theta_uniform(a.ptab, 0);
#endif

    const int64_t n_pairs = (a.n_chains + 1) >> 1;
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    const int64_t first = (int64_t)blockIdx.x * AMC_BLOCK;
    unsigned long long wave_acc = 0;   // wave-uniform
    // REDUCE with the pool-wide counter: the row's last column is the accepted total this block can see -- its own slot
    // (plus the slots beyond this launch's grid, filled by launches with a larger one).  Launches are ordered on the
    // stream and only block b touches slot b inside a launch, so the old values are read HERE, under the first load,
    // instead of by a returning atomic at the very end of the block.
    unsigned long long slots_before = 0;
    if (REDUCE && !LOG && !MULTI && threadIdx.x == 0)
        for (int sl = (int)blockIdx.x; sl < a.n_slots; sl += (int)gridDim.x) slots_before += a.acc_total[sl];

    // Memory schedule.  hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` at the top of a loop that carries a
    // prefetched load across its back edge while stores are pending: vmcnt counts loads and stores together
    // and the two kinds complete out of order with respect to each other, so no counted wait can name one
    // load ("mixed pending events" in LLVM's SIInsertWaitcnts).  A store issued at the END of an iteration
    // is therefore waited for immediately, at full write-through latency.  The schedule below issues ALL
    // memory operations at the START of an iteration -- the prefetch of iteration i+1 and the stores of
    // iteration i-1's results (x and the step-log word, kept one iteration in registers) -- so the vmcnt(0) at
    // the end of the iteration finds them a whole iteration (~2 us of other waves' arithmetic) old.
    // An iteration that has a successor covers 256 in-range pairs on every lane (stride >= 256), so the
    // loop body runs without per-lane predicates; only the LAST iteration of a block can be ragged and is
    // peeled.  Loads need no clamp either: the arrays carry AMC_PAD_DOUBLES of readable padding.
    auto load_x = [&](int64_t b) -> real2 { return load_pair_block(a.x + 2 * b); };
    auto load_b = [&](int64_t b) -> real2 { return load_pair_block(a.beta_arr + 2 * b); };
    real2 x_nxt = {(real_t)0.0, (real_t)0.0}, b_nxt = {(real_t)a.beta, (real_t)a.beta};
    if (first < n_pairs) {
        x_nxt = load_x(first);
        if (BETA) b_nxt = load_b(first);
    }
    // SINGLE: the Philox draws of an iteration are formed one iteration ahead -- those of the first iteration
    // right here, while the first load and the table loads are in flight (the arithmetic of ~80 VALU
    // instructions per wave would otherwise start only after both have landed).
    constexpr bool AHEAD = SINGLE;
    StepDraws dr_nxt = {};
    if (AHEAD && first < n_pairs) dr_nxt = step_draws(a, a.pair0 + (uint64_t)(first + threadIdx.x), a.t0);
#ifdef AMC_USER_LOGQ
    stage_user_theta(a.ptab);
#endif
    stage_math_tables(s_math, threadIdx.x, AMC_BLOCK);     // overlaps the latency of the first load; ends in a barrier
    // Drain the first load HERE, once.  Otherwise the compiler must assume it is still pending inside the loop
    // and puts a counted wait before the first use of x in every iteration -- which in steady state waits for
    // the prefetch issued a few dozen instructions earlier instead of leaving it a whole iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0), other counters untouched
    real2 x_done = {(real_t)0.0, (real_t)0.0};
    uint32_t lw_done = 0;
    int64_t base_done = -1;                                  // block-uniform
    int64_t base = first;
    for (; base + stride < n_pairs; base += stride) {        // full iterations
        const int64_t p = base + threadIdx.x;
        real2 xv = x_nxt;
        const real_t b0 = b_nxt.x, b1 = b_nxt.y;
        x_nxt = load_x(base + stride);
        if (BETA) b_nxt = load_b(base + stride);
        if (base_done >= 0) {
            store_pair_block_writethrough(a.x + 2 * base_done, x_done);
            if (LOG && SINGLE) store_log_pair<LOG>(a, a.log_pos, base_done + threadIdx.x, lw_done);
        }
        const StepDraws dr = dr_nxt;
        uint32_t lw = 0;
        pair_steps<POT, MULTI, LOG, SINGLE, AHEAD>(a, xv, b0, b1, a.pair0 + (uint64_t)p, p, true, true, s_tab, s_pick, s_math,
                                                   sigma1, den1, rden1, logc1, wave_acc, lw, &dr, math_k_literal(), th1);
        // a successor exists (loop condition); lanes past the end of a ragged one form draws nobody uses
        if (AHEAD) dr_nxt = step_draws(a, a.pair0 + (uint64_t)(p + stride), a.t0);
        if (REDUCE) {
            red_add_pair<POT>(red, xv, true, true, s_math, s_red[threadIdx.x >> 6], a.red_cols);
        }
        x_done = xv;
        lw_done = lw;
        base_done = base;
    }
    if (base < n_pairs) {                                    // last, possibly ragged, iteration
        const int64_t p = base + threadIdx.x;
        const bool v0 = p < n_pairs;
        const bool v1 = v0 && (2 * p + 1 < a.n_chains);
        real2 xv = x_nxt;
        if (base_done >= 0) {
            store_pair_block_writethrough(a.x + 2 * base_done, x_done);
            if (LOG && SINGLE) store_log_pair<LOG>(a, a.log_pos, base_done + threadIdx.x, lw_done);
        }
        uint32_t lw = 0;
        pair_steps<POT, MULTI, LOG, SINGLE, AHEAD>(a, xv, b_nxt.x, b_nxt.y, a.pair0 + (uint64_t)(v0 ? p : 0), p, v0, v1,
                                                   s_tab, s_pick, s_math, sigma1, den1, rden1, logc1, wave_acc, lw, &dr_nxt, math_k_literal(), th1);
        // a lone last chain (odd n_chains) writes its whole pair (x and log): the odd slot is padding
        if (v0) {
            store_pair_block_writethrough(a.x + 2 * base, xv);
            if (LOG && SINGLE) store_log_pair<LOG>(a, a.log_pos, p, lw);
        }
        if (REDUCE) red_add_pair<POT>(red, xv, v0, v1, s_math, s_red[threadIdx.x >> 6], a.red_cols);
    }
    if (REDUCE)
        red_finish<POT>(red, s_red, a.red_partials + (int64_t)blockIdx.x * a.red_stride, a.red_stride == RED_COMPACT_WORDS, a.red_cols);
    if (!MULTI) {
        // Pool-wide accepted count: each block owns ONE u64 slot (thousands of atomics on a single
        // address at kernel end serialise at ~13 ns each; one address per block does not contend).
        __shared__ unsigned long long s_acc[AMC_BLOCK / 64];
        if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = wave_acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < AMC_BLOCK / 64; ++w) t += s_acc[w];
            if (REDUCE && !LOG) {
                // the callback wants the pool-wide accepted total: column 4 of this block's row carries the slot's
                // value after this launch (exact in a double below 2^53); the rows are summed by the host
                if (t != 0) __hip_atomic_fetch_add(a.acc_total + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.red_partials[(int64_t)blockIdx.x * a.red_stride + (a.red_stride == RED_COMPACT_WORDS ? (int)RED_COMPACT_SLOT : (int)RED_ROW_SLOT)] =
                    (xs_word)__double_as_longlong((double)(slots_before + t));
            } else if (t != 0) {
                // no-return atomic: fire and forget (a read-modify-write would hold the block for a memory round trip)
                __hip_atomic_fetch_add(a.acc_total + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The same pool-wide bookkeeping for kernels that run an MH step without being the sweep kernel (K == 1 only).
// Returns the block's count (valid in thread 0).
__device__ __forceinline__ unsigned long long add_block_accepts(unsigned long long* acc_total, unsigned long long wave_acc)
{
    __shared__ unsigned long long s_acc2[AMC_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) s_acc2[threadIdx.x >> 6] = wave_acc;
    __syncthreads();
    unsigned long long t = 0;
    if (threadIdx.x == 0) {
        for (int w = 0; w < AMC_BLOCK / 64; ++w) t += s_acc2[w];
        if (t != 0) __hip_atomic_fetch_add(acc_total + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return t;
}

// Adds `n_rows` rows of the step log into the per-chain u32 counters acc[K][m_stride] / tot[K - 1][m_stride].
// total_calls of the LAST move has no array: every chain takes the same number of MH steps (mc_sweep!, metropolis.jl:205-210),
// so sum_k total_calls_ck == t_counted on every chain and the last move's count is t_counted minus the others -- one
// read-modify-write array of four less at K = 2 (K == 1: none at all, total_calls is the step count).  Entries of the
// padding behind n_chains are never read back as counts.
// KS moves per launch (K == KS <= 4 in one pass; more moves: GROUP below).  A block works on tiles of 4096 adjacent chains.
// Log side: a thread owns SIXTEEN adjacent chains -- one 8-byte load per row of the nibble log (the block reads 2 KiB of every
// row), one 16-byte load of the byte log -- and accumulates the rows bytewise in packed registers: the accept bit and the move
// bits of four chains (nibble log: the four even, then the four odd chains of a word) are masked out at once and added as
// four 8-bit counters (n_rows <= 255), ~2 VALU operations per chain and row instead of 6 K; one byte permute per word pair
// puts the nibble log's counts back into chain order.  Counter side: the
// packed words go through LDS so that lane t updates the quad of chains 4 (i 256 + t), i = 0..3 -- 16-byte
// read-modify-writes that are contiguous across the wave (a thread updating its own sixteen chains would touch 16 bytes
// in every 64).
// RATIO: the counters are in registers right after the update, so the launch also forms
// callback_acceptance's sums  sum_c accepted_ck / total_ck  (metropolis.jl:319-321; Int/Int -> Float64 division,
// 0/0 = NaN) -- block partials [grid][rp_stride] -- instead of a reduction pass re-reading 8 K bytes per chain.
// t_counted: MH steps counted per chain INCLUDING the rows of this launch (< 2^32: the host refuses to count further).
// CT / HIGH: the counters' storage.  The callback's fold is a read-modify-write of every counter from HBM, so handles with
// K <= 4 keep them as two u16 planes: `acc` / `tot` hold the low halves, `acc_hi` / `tot_hi` the high halves.  No counter can
// exceed the number of steps counted, so while that is below 2^16 the high planes are all zero and the launch leaves them
// alone (HIGH = false: 4 bytes per counter and fold); afterwards it READS the high half and writes it only where a low half
// has just carried (HIGH = true: 6 bytes, against 8 for a u32 counter).  At K = 2 and ten packed rows that is 17 / 23 / 29
// bytes per chain.  CT = uint32_t (K > 4, or AMC_WIDE_COUNTERS): plain u32 arrays, no planes.
#define AMC_FOLD_TILE (16 * AMC_BLOCK)
// four adjacent counters as one aligned access: 16 bytes of u32, 8 bytes of u16
__device__ __forceinline__ uint4 load_counter_quad(const uint32_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 load_counter_quad(const uint16_t* p)
{
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return uint4{v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16};
}
__device__ __forceinline__ void store_counter_quad(uint32_t* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ void store_counter_quad(uint16_t* p, uint4 v)
{
    *reinterpret_cast<uint2*>(p) = uint2{v.x | (v.y << 16), v.z | (v.w << 16)};       // every value < 2^16 (see above)
}

// adds the 8-bit increments of `w` to a quad of counters; returns the quad's full values
template <bool HIGH, typename CT>
__device__ __forceinline__ uint4 bump_counter_quad(CT* lo, uint16_t* hi, uint32_t w)
{
    uint4 v = load_counter_quad(lo);
    v.x += w & 0xFFu; v.y += (w >> 8) & 0xFFu; v.z += (w >> 16) & 0xFFu; v.w += w >> 24;
    if (!HIGH) {
        store_counter_quad(lo, v);                         // u32, or u16 that cannot carry yet
        return v;
    }
    uint4 h = load_counter_quad(hi);
    if (((v.x | v.y | v.z | v.w) >> 16) != 0u) {           // a low half has carried: rare (n_rows in 65 536 folds per counter)
        h.x += v.x >> 16; h.y += v.y >> 16; h.z += v.z >> 16; h.w += v.w >> 16;
        store_counter_quad(hi, h);
    }
    v.x &= 0xFFFFu; v.y &= 0xFFFFu; v.z &= 0xFFFFu; v.w &= 0xFFFFu;
    store_counter_quad(lo, v);
    return uint4{v.x | (h.x << 16), v.y | (h.y << 16), v.z | (h.z << 16), v.w | (h.w << 16)};
}

// GROUP passes (pools of more than four moves): the register-resident form counts four moves per launch.  GROUP = 0: the whole
// pool in one pass (K = KS <= 4).  GROUP = 1: moves 4 g .. 4 g + 3 of a larger pool (KS = 4; every one of them has a total
// array), GROUP = 2: the pool's last moves 4 g .. K - 1 (KS = K - 4 g; the very last has no total array); `group` = g, and
// acc / tot point at move 4 g's rows.  ceil(K / 4) passes over the log instead of one read-modify-write per chain and
// logged step (that form took 12 ms per 128 rows at 1e7 chains: 94 us per sweep at K = 5, 159 at K = 8).
// BYTES: the log holds one byte per chain (pools of more than eight moves) instead of a nibble.
template <int KS, bool RATIO = false, typename CT = uint32_t, bool HIGH = false, int GROUP = 0, bool BYTES = false>
__global__ __launch_bounds__(AMC_BLOCK) void fold_log_kernel(const uint8_t* log, int n_rows, CT* acc,
                                                              CT* tot, uint16_t* acc_hi, uint16_t* tot_hi,
                                                              int64_t n_chains, int64_t m_stride,
                                                              int group, uint64_t t_counted, xs_word* ratio_partials,
                                                              int rp_stride)
{
    static_assert(KS >= 1 && KS <= 4, "four moves per pass");
    static_assert(!HIGH || sizeof(CT) == 2, "high planes belong to 16-bit low planes");
    static_assert(!RATIO || GROUP == 0, "ratio sums ride on the single pass of pools of up to four moves");
    static_assert(GROUP != 1 || KS == 4, "inner groups are full");
    static_assert(!BYTES || GROUP != 0, "pools of up to eight moves log nibbles");
    constexpr int KK = KS;
    constexpr uint32_t ONES = 0x01010101u;
    constexpr bool ALL_TOT = GROUP == 1;                                         // every move of this pass has a total array
    __shared__ __attribute__((aligned(16))) uint32_t s_pk[2 * KK][4 * AMC_BLOCK];   // [k: accepted, total][quad of the tile]
    // callback_acceptance's sums: kind-Q columns of quantum 2^XS_E_RATIO (amc_xsum.h), 16 ratios per lane, move and tile
    unsigned long long ratio[KK];
    bool ratio_nan[KK];
    int ratio_tiles = 0;
    __shared__ QSlot s_ratio[RATIO ? AMC_BLOCK / 64 : 1][KK];
    if (RATIO) {
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int k = 0; k < KK; ++k) q_slot_clear(s_ratio[threadIdx.x >> 6][k]);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < KK; ++k) { ratio[k] = 0ull; ratio_nan[k] = false; }
    // the group a step belongs to: bit 3 of a nibble (pools of 5..8), bits 3..6 of a byte (up to 64 moves)
    const uint32_t group_field = BYTES ? 0x0F0F0F0Fu : ONES;
    const uint32_t group_word = (uint32_t)group * ONES;
    const int64_t n_tiles = (n_chains + AMC_FOLD_TILE - 1) / AMC_FOLD_TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t c_tile = tile * AMC_FOLD_TILE;
        const int64_t c_mine = c_tile + 16 * (int64_t)threadIdx.x;   // first of this thread's 16 chains (log side)
        // every per-chain array is m_stride long (a multiple of 256, >= n_chains + 520): indices below m_stride are
        // readable and writable, what lies behind n_chains is padding
        const bool log_ok = c_mine < m_stride;
        uint32_t pa[KK][4], pt[KK][4];                     // packed 8-bit counters of four chains each (see below for which)
#pragma unroll
        for (int k = 0; k < KK; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) pa[k][j] = pt[k][j] = 0u;
        if (log_ok) {
            // w[j]: four chains' steps in the low bits of its four bytes -- nibble log: word j / 2 of the load, its even (j even)
            // or odd chains; byte log: chains 4 j .. 4 j + 3
            auto add_words = [&](const uint32_t (&w)[4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = w[j] & ONES;                       // accepted
                    const uint32_t b0 = (w[j] >> 1) & ONES, b1 = (w[j] >> 2) & ONES;      // move index within its group
                    uint32_t eq[4];
                    if (KS == 1) { eq[0] = ONES; }
                    else if (KS == 2) { eq[1] = b0; eq[0] = b0 ^ ONES; }
                    else { eq[0] = (b0 | b1) ^ ONES; eq[1] = b0 & ~b1; eq[2] = b1 & ~b0; eq[3] = b0 & b1; }
                    uint32_t mine = ONES;
                    if (GROUP != 0) {
                        // bytes whose group field equals `group`: x = field ^ group is zero there and below 0x80 everywhere,
                        // so bit 7 of x + 0x7F marks the others
                        const uint32_t x = ((w[j] >> 3) & group_field) ^ group_word;
                        mine = (((x + 0x7F7F7F7Fu) >> 7) & ONES) ^ ONES;
                    }
#pragma unroll
                    for (int k = 0; k < KK; ++k) {
                        const uint32_t hit = GROUP == 0 ? eq[k] : (eq[k] & mine);
                        if (ALL_TOT || k < KK - 1) pt[k][j] += hit;       // the pool's last move has no total array
                        pa[k][j] += hit & a;
                    }
                }
            };
            // Rows in flight per lane: with one, a wave has 512 bytes outstanding and the launch waits for latency (the full
            // 128-row fold moved 0.76 GB in 202 us; four in flight: 154 us).  The callback's form needs its registers for
            // the counter side (83 VGPRs with four): two in flight there (ten-row launch 35.0-35.7 -> 33.7 us before the
            // 16-bit mark, 41.5 -> 39-40 after; same box).
            constexpr int U = RATIO ? 2 : 4;
            if (!BYTES) {
                const int64_t row_bytes = m_stride >> 1;
                const uint8_t* mine_rows = log + (c_mine >> 1);
                auto add_row = [&](const uint2 w2) {
                    const uint32_t w[4] = {w2.x, w2.x >> 4, w2.y, w2.y >> 4};
                    add_words(w);
                };
                int r = 0;
                for (; r + U <= n_rows; r += U) {
                    uint2 w[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const uint2*>(mine_rows + (int64_t)(r + u) * row_bytes);
#pragma unroll
                    for (int u = 0; u < U; ++u) add_row(w[u]);
                }
                for (; r < n_rows; ++r) add_row(*reinterpret_cast<const uint2*>(mine_rows + (int64_t)r * row_bytes));
            } else {
                const uint8_t* mine_rows = log + c_mine;
                auto add_row = [&](const uint4 w4) {
                    const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
                    add_words(w);
                };
                int r = 0;
                for (; r + 2 <= n_rows; r += 2) {
                    const uint4 w0 = *reinterpret_cast<const uint4*>(mine_rows + (int64_t)r * m_stride);
                    const uint4 w1 = *reinterpret_cast<const uint4*>(mine_rows + (int64_t)(r + 1) * m_stride);
                    add_row(w0); add_row(w1);
                }
                for (; r < n_rows; ++r) add_row(*reinterpret_cast<const uint4*>(mine_rows + (int64_t)r * m_stride));
            }
        }
        // nibble log: (even chains 0 2 4 6, odd chains 1 3 5 7) of a word -> chains 0..3 and 4..7
        auto in_chain_order = [](const uint32_t (&v)[4]) {
            if (BYTES) return uint4{v[0], v[1], v[2], v[3]};
            return uint4{__builtin_amdgcn_perm(v[1], v[0], 0x05010400u), __builtin_amdgcn_perm(v[1], v[0], 0x07030602u),
                         __builtin_amdgcn_perm(v[3], v[2], 0x05010400u), __builtin_amdgcn_perm(v[3], v[2], 0x07030602u)};
        };
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            reinterpret_cast<uint4*>(s_pk[2 * k])[threadIdx.x] = in_chain_order(pa[k]);
            if (ALL_TOT || k < KK - 1) reinterpret_cast<uint4*>(s_pk[2 * k + 1])[threadIdx.x] = in_chain_order(pt[k]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int quad = i * AMC_BLOCK + (int)threadIdx.x;
            const int64_t c0 = c_tile + 4 * (int64_t)quad;
            if (c0 >= m_stride) continue;
            uint32_t tsum[4] = {0u, 0u, 0u, 0u};           // total_calls of the moves before k, per chain of the quad
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                const int64_t at = (int64_t)k * m_stride + c0;
                const uint4 va = bump_counter_quad<HIGH>(acc + at, HIGH ? acc_hi + at : nullptr, s_pk[2 * k][quad]);
                uint4 vt;
                if (ALL_TOT || k < KK - 1) {
                    vt = bump_counter_quad<HIGH>(tot + at, HIGH ? tot_hi + at : nullptr, s_pk[2 * k + 1][quad]);
                    tsum[0] += vt.x; tsum[1] += vt.y; tsum[2] += vt.z; tsum[3] += vt.w;
                } else {
                    const uint32_t tc = (uint32_t)t_counted;
                    vt = uint4{tc - tsum[0], tc - tsum[1], tc - tsum[2], tc - tsum[3]};
                }
                if (RATIO) {
                    const uint32_t a4[4] = {va.x, va.y, va.z, va.w}, t4[4] = {vt.x, vt.y, vt.z, vt.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c0 + e < n_chains)                        // the padding behind the last chain has no ratio
                            ratio_add(ratio[k], ratio_nan[k], a4[e], t4[e]);
                }
            }
        }
        __syncthreads();                                   // the next tile overwrites s_pk
        if (RATIO && ++ratio_tiles == xs::XS_RATIO_LANE_CAP / 16) {       // (ensembles beyond 2e9 chains)
#pragma unroll
            for (int k = 0; k < KK; ++k) q_flush_int(ratio[k], &s_ratio[threadIdx.x >> 6][k]);
            ratio_tiles = 0;
        }
    }
    if (RATIO) {
        // one row of two words per move and block; a 0/0 = NaN among the ratios (a chain that never picked the move,
        // metropolis.jl:320) makes every column NaN that met one
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            q_flush_int(ratio[k], &s_ratio[threadIdx.x >> 6][k]);
            if (__builtin_amdgcn_ballot_w64(ratio_nan[k]) != 0ull && (threadIdx.x & 63) == 0)
                s_ratio[threadIdx.x >> 6][k].flags |= (unsigned int)xs::XS_F_NAN;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < KK; ++k)
                xs_store_q_row(ratio_partials + ((int64_t)blockIdx.x * rp_stride + k) * XS_ROW_Q, q_block_total<KK>(s_ratio, k));
        }
    }
}


// K0: synthetic initial ensemble, x_c = lo + (hi-lo)*u (MC_harmonic_oscillator.jl:13).
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void init_uniform_kernel(double* x, int64_t n_chains, uint64_t pair0,
                                                                  uint32_t key0, uint32_t key1, double lo,
                                                                  double hi)
{
    const int64_t n_pairs = (n_chains + 1) >> 1;
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t p = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; p < n_pairs; p += stride) {
        const u32x4 v = philox4x32_10(draw_counter(pair0 + (uint64_t)p, 0, 0, STREAM_INIT), key0, key1);
        const double x0 = lo + (hi - lo) * uniform_co(v.x, v.y);
        const double x1 = lo + (hi - lo) * uniform_co(v.z, v.w);
        x[2 * p] = x0;
        if (2 * p + 1 < n_chains) x[2 * p + 1] = x1;
    }
}

// Derived per-move parameters, computed ON DEVICE so the arithmetic is the kernel's.
// den = 2*(s*s); logc = log(2pi*(s*s))/2 (particle_1d.jl:53); cum = running sum of
// weights in the order Distributions.jl accumulates them; dden, dlhalf: d/dsigma
// pieces of gradients.jl:28-33 (ForwardDiff's dual rules written out).
// (the entries of move k that depend on its sigma alone)
// What a move's sigma determines (the loop invariants of particle_1d.jl:53 and of its sigma-derivative), in registers.
enum { DEF_SIGMA = 0, DEF_DEN, DEF_RDEN, DEF_LOGC, DEF_C3HI, DEF_C3LO, DEF_DLHALF, DEF_DDEN, DEF_N };
__device__ __forceinline__ void derive_move_params(double sigma, double (&d)[DEF_N])
{
    const double TWO_PI = 0x1.921fb54442d18p+2;
    const double s2 = sigma * sigma;
    const double ds2 = sigma + sigma;
    d[DEF_SIGMA] = sigma;
    d[DEF_DEN] = 2.0 * s2;
    d[DEF_RDEN] = 1.0 / (2.0 * s2);                                  // RN(1/den) for div_by_const
    d[DEF_DDEN] = 2.0 * ds2;
    const double av = TWO_PI * s2;
    d[DEF_LOGC] = log_f64(av) / 2.0;
    d[DEF_DLHALF] = ((TWO_PI * ds2) / av) / 2.0;
    // dden / den^2 (= 1/sigma^3) as an unevaluated sum hi + lo: the coefficient of delta^2 in the estimator's
    // d logq / d sigma (pg_sample), good to ~2^-100 so that no rounding of a CONSTANT biases a sum over 1e7+ samples
    const double den = 2.0 * s2, dden = 2.0 * ds2;
    const double d_hi = den * den, d_lo = __builtin_fma(den, den, -d_hi);
    const double c_hi = dden / d_hi;
    const double res = __builtin_fma(-c_hi, d_hi, dden) - c_hi * d_lo;
    d[DEF_C3HI] = c_hi;
    d[DEF_C3LO] = res / d_hi;
}
__device__ __forceinline__ void prepare_move_params(double* ptab, int k, double sigma)
{
    double d[DEF_N];
    derive_move_params(sigma, d);
    ptab[PT_DEN * AMC_MAX_MOVES + k] = d[DEF_DEN];
    ptab[PT_RDEN * AMC_MAX_MOVES + k] = d[DEF_RDEN];
    ptab[PT_DDEN * AMC_MAX_MOVES + k] = d[DEF_DDEN];
    ptab[PT_LOGC * AMC_MAX_MOVES + k] = d[DEF_LOGC];
    ptab[PT_DLHALF * AMC_MAX_MOVES + k] = d[DEF_DLHALF];
    ptab[PT_C3HI * AMC_MAX_MOVES + k] = d[DEF_C3HI];
    ptab[PT_C3LO * AMC_MAX_MOVES + k] = d[DEF_C3LO];
}

__device__ __forceinline__ void prepare_params(double* ptab, int n_moves)
{
    double cp = 0.0;
    for (int k = 0; k < n_moves; ++k) {
        prepare_move_params(ptab, k, ptab[PT_SIGMA * AMC_MAX_MOVES + k]);
        const double w = ptab[PT_WEIGHT * AMC_MAX_MOVES + k];
        cp = (k == 0) ? w : cp + w;
        ptab[PT_CUM * AMC_MAX_MOVES + k] = cp;
    }
}

AMC_KERNEL_LINKAGE __global__ void prepare_params_kernel(double* ptab, int n_moves)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    prepare_params(ptab, n_moves);
}

// The move-pick table (see AMC_PICK_CELLS): cell c covers the pick uniforms r in [c, c+1) 2^-12 (both ends exact).
// The walk's count #(cum[i] <= r), i < K-1, is monotone in r, so it is the same for every r of the cell iff it is the
// same at the two ends: #(cum[i] <= c 2^-12) == #(cum[i] < (c+1) 2^-12).  Launched after prepare_params (same stream)
// whenever the weights change.
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void prepare_pick_kernel(const double* ptab, int n_moves, uint8_t* pick_tab)
{
    const int c = (int)(blockIdx.x * AMC_BLOCK + threadIdx.x);
    if (c >= AMC_PICK_CELLS) return;
    const double lo = (double)c * 0x1.0p-12, hi = (double)(c + 1) * 0x1.0p-12;
    int n_lo = 0, n_hi = 0;
    for (int i = 0; i < n_moves - 1; ++i) {
        const double cum = ptab[PT_CUM * AMC_MAX_MOVES + i];
        n_lo += (cum <= lo) ? 1 : 0;
        n_hi += (cum < hi) ? 1 : 0;
    }
    pick_tab[c] = (n_lo == n_hi) ? (uint8_t)n_lo : (uint8_t)AMC_PICK_OPEN;
}

// ---- device-resident policy-gradient bookkeeping (src/PolicyGuided/estimator.jl:130-131, update.jl:50-57) ----
struct PgIds { int32_t v[AMC_MAX_LEARN]; };
struct PgOpts { int32_t kind[AMC_MAX_LEARN]; double h0[AMC_MAX_LEARN]; double h1[AMC_MAX_LEARN]; };
enum { OPT_STATIC = 0, OPT_VPG = 1, OPT_BLPG = 2, OPT_BLAPG = 3, OPT_NPG = 4, OPT_ANPG = 5, OPT_BLANPG = 6 };

// gradients_data[k] = gradients_data[k] + gd (estimator.jl:130): red[l*4 + i] holds the (all-reduced) sums of
// (j, grad j, grad logq, g) over chains x q_batch samples of learnable move l; acc is [AMC_MAX_MOVES][5].
__device__ __forceinline__ void pg_accumulate_one(const double* red, int l, int lid, double n_samples, double* acc)
{
    double* a = acc + lid * 5;
    for (int i = 0; i < 4; ++i) a[i] += red[l * 4 + i];
    a[4] += n_samples;
}

// make_step!(::PolicyGradientUpdate) (update.jl:50-57) for P = 1: average (gradients.jl:83-85), learning_step!
// (learning.jl:32-34, 50-52, 77-79, 103-105, 130-134, 160-164; inv(g + eps I) is a scalar reciprocal), reset
// the accumulators, refresh the derived parameter table.  A step that leaves sigma outside [1e-100, 1e100]
// (or NaN) is not applied; status[0] is set instead.
// learning_step! of one move from its averaged GradientData (see below); `theta` is its sigma
__device__ __forceinline__ double pg_learning_step(int kind, double h0, double h1, double theta, double j, double dj, double dlogq,
                                                   double g)
{
    switch (kind) {
    case OPT_VPG: return theta + h0 * dj;
    case OPT_BLPG: return theta + h0 * (dj - j * dlogq);
    case OPT_BLAPG: {
        const double eta = __builtin_sqrt(2.0 * h0 / (dj * dj + h1));
        return theta + eta * (dj - j * dlogq);
    }
    case OPT_NPG: {
        const double finv = 1.0 / (g + h1 * 1.0);
        return theta + h0 * finv * dj;
    }
    case OPT_ANPG: {
        const double finv = 1.0 / (g + h1 * 1.0);
        const double eta = __builtin_sqrt(2.0 * h0 / (dj * (finv * dj)));
        return theta + eta * finv * dj;
    }
    case OPT_BLANPG: {
        const double finv = 1.0 / (g + h1 * 1.0);
        const double bj = dj - j * dlogq;
        const double eta = __builtin_sqrt(2.0 * h0 / (bj * (finv * bj)));
        return theta + eta * finv * bj;
    }
    default: return theta;
    }
}

// One thread.  `red` != nullptr: first gradients_data[k] += the sums in red (pg_accumulate_one), in registers -- the
// accumulators are read once, the new sigma and what derives from it are written from registers, and only the moves that
// learned get their derived parameters refreshed (the cumulative weights do not depend on sigma): the few dependent round
// trips to memory this thread makes are the tail of every PGMC time step (62.4 -> 61.4 us per fused step, same box).
// theta_used (optional): the sigma the launch proposed with, per learnable move -- the table's, except in a launch that took a
// pending step in its prologue (pg_apply_pending: the table is then one step behind, sigma' lives in the ring)
__device__ __forceinline__ void pg_update_all(double* ptab, double* acc, int n_learn, const int32_t* ids,
                                              const PgOpts& opt, int n_moves, int* status, const double* red = nullptr,
                                              double n_samples = 0.0, const double* theta_used = nullptr)
{
    (void)n_moves;
    for (int l = 0; l < n_learn; ++l) {
        const int k = ids[l];
        double* a = acc + k * 5;
        double v[5] = {a[0], a[1], a[2], a[3], a[4]};
        const double theta = theta_used ? theta_used[l] : ptab[PT_SIGMA * AMC_MAX_MOVES + k];
        if (red) {
            for (int i = 0; i < 4; ++i) v[i] += red[l * 4 + i];
            v[4] += n_samples;
        }
        const double n = v[4];
        const double j = v[0] / n, dj = v[1] / n, dlogq = v[2] / n, g = v[3] / n;
        const double next = pg_learning_step(opt.kind[l], opt.h0[l], opt.h1[l], theta, j, dj, dlogq, g);
        for (int i = 0; i < 5; ++i) a[i] = 0.0;
        if (next >= 1e-100 && next <= 1e100) {
            ptab[PT_SIGMA * AMC_MAX_MOVES + k] = next;
            prepare_move_params(ptab, k, next);
        } else
            status[0] = 1;
    }
}

AMC_KERNEL_LINKAGE __global__ void pg_update_kernel(double* ptab, double* acc, int n_learn, PgIds ids, PgOpts opt, int n_moves, int* status)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    pg_update_all(ptab, acc, n_learn, ids.v, opt, n_moves, status);
}

// K2a: the callback reductions as a pass of their own (when no sweep launch could carry them).  Row b of `rows` (pinned
// host memory, p_stride words apart): the kind-R columns sum e (callback_energy particle_1d.jl:68-70), sum x, sum x^2
// (distribution_test.jl:36-37) as in red_finish, the count, and the pool-wide accepted slots b, b + grid, ... (exact).
// ratio_mode: 0 = no per-chain ratios here (K == 1 without per-chain counters: the host uses the pool-wide total; K <= 4: the
//                 fold of the step log forms them);
//             1 = K == 1 with per-chain acc (total = t_steps for every chain), 2 = K > 1: sum_c accepted/total per move
//                 (callback_acceptance metropolis.jl:319-321) as kind-Q integers of quantum 2^XS_E_RATIO, added into
//                 ratio_acc[k][3] = (sum of low 32-bit halves, sum of high halves, blocks that met a NaN) with one atomic per block
//                 and word -- integer additions, so the order the blocks arrive in does not matter.
template <int POT>
__global__ __launch_bounds__(AMC_BLOCK) void reduce_kernel(const real_t* x, const uint32_t* acc,
                                                            const uint32_t* tot, int64_t n_chains,
                                                            int64_t m_stride, int n_moves, int ratio_mode,
                                                            uint64_t t_steps, xs_word* rows, int p_stride,
                                                            const unsigned long long* slots, int n_slots,
                                                            unsigned long long* ratio_acc, int red_cols,
                                                            const unsigned long long* acc_base, const unsigned long long* tot_base,
                                                            uint64_t t_base)
{
    // acc_base / tot_base (nullptr on most handles): what the 32-bit counters have been carried into so far, t_base the steps
    // counted with it (counter_rebase_kernel); a counter's value is base + array, the steps counted t_base + t_steps
    // p_stride names the row's form (red_finish): the compact one for passes of at most RED_COMPACT_TRIPS trips per lane
    const bool compact = p_stride == RED_COMPACT_WORDS;
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    __shared__ double s_math[POT == POT_CUSTOM ? TAB_DOUBLES : 1];      // a custom potential may call amc_exp
    if (POT == POT_CUSTOM) stage_math_tables(s_math, threadIdx.x, AMC_BLOCK);
    constexpr int RNC = RedCols<POT>::NC;
    RLanes<RNC> red;
    __shared__ xs::PartR s_red[AMC_BLOCK / 64][RNC];
    r_init(red, s_red[threadIdx.x >> 6]);
    int n_trips = 0;                                          // summands per column since the last flush (scalar unit)
    const int64_t n_pairs = (n_chains + 1) >> 1;              // 16-byte loads; x is padded, the odd slot of a lone last chain is masked
    // every lane of a wave takes the same number of trips (the flushes inside are wave-wide): lanes past the end add zeros
    const int64_t wave_first = (int64_t)blockIdx.x * AMC_BLOCK + (threadIdx.x & ~63);
    for (int64_t pw = wave_first; pw < n_pairs; pw += stride) {
        const int64_t p = pw + (threadIdx.x & 63);
        const bool v0 = p < n_pairs, v1 = v0 && (2 * p + 1 < n_chains);
        real2 xp = {(real_t)0.0, (real_t)0.0};
        if (v0) xp = *reinterpret_cast<const real2*>(x + 2 * p);
        red_add_pair<POT>(red, xp, v0, v1, s_math, s_red[threadIdx.x >> 6], red_cols);
        if (++n_trips > xs::XS_LANE_CAP - 2) { r_flush(red, s_red[threadIdx.x >> 6]); n_trips = 0; }      // (never in a compact pass)
    }
    xs_word* out = rows + (int64_t)blockIdx.x * p_stride;
    red_finish<POT>(red, s_red, out, compact, red_cols);
    if (threadIdx.x == 0) {
        unsigned long long a = 0;
        if (slots)
            for (int s = blockIdx.x; s < n_slots; s += gridDim.x) a += slots[s];
        out[compact ? (int)RED_COMPACT_SLOT : (int)RED_ROW_SLOT] = (xs_word)__double_as_longlong((double)a);
    }
    if (ratio_mode != 0) {
        __shared__ QSlot s_ratio[AMC_BLOCK / 64][1];
        for (int k = 0; k < n_moves; ++k) {
            __syncthreads();
            if ((threadIdx.x & 63) == 0) q_slot_clear(s_ratio[threadIdx.x >> 6][0]);
            __syncthreads();
            unsigned long long r = 0ull;
            bool nan = false;
            int n_r = 0;
            for (int64_t cw = (int64_t)blockIdx.x * AMC_BLOCK + (threadIdx.x & ~63); cw < n_chains; cw += stride) {
                const int64_t c = cw + (threadIdx.x & 63);        // every lane of a wave takes the same trips (the flush is wave-wide)
                if (++n_r == xs::XS_RATIO_LANE_CAP) { q_flush_int(r, &s_ratio[threadIdx.x >> 6][0]); n_r = 0; }
                if (c >= n_chains) continue;
                unsigned long long a = acc[(int64_t)k * m_stride + c];
                if (acc_base) a += acc_base[(int64_t)k * m_stride + c];
                unsigned long long n = t_base + t_steps;
                if (ratio_mode == 2) {
                    // total_calls of the last move has no array: the step count minus the other moves' (fold_log_kernel)
                    if (k + 1 < n_moves) {
                        n = tot[(int64_t)k * m_stride + c];
                        if (tot_base) n += tot_base[(int64_t)k * m_stride + c];
                    } else {
                        unsigned long long others = 0;
                        for (int j = 0; j + 1 < n_moves; ++j) {
                            others += tot[(int64_t)j * m_stride + c];
                            if (tot_base) others += tot_base[(int64_t)j * m_stride + c];
                        }
                        n = t_base + t_steps - others;
                    }
                }
                if (acc_base) ratio_add(r, nan, a, n);      // Int/Int -> Float64 division; 0/0 = NaN like the reference
                else ratio_add(r, nan, (uint32_t)a, (uint32_t)n);
            }
            q_flush_int(r, &s_ratio[threadIdx.x >> 6][0]);
            if (__builtin_amdgcn_ballot_w64(nan) != 0ull && (threadIdx.x & 63) == 0)
                s_ratio[threadIdx.x >> 6][0].flags |= (unsigned int)xs::XS_F_NAN;
            __syncthreads();
            if (threadIdx.x == 0) {
                // the block's total as low 32 bits + high part: two 64-bit atomics on the move's words (three with the NaN count)
                const xs::PartQ t = q_block_total<1>(s_ratio, 0);
                if (t.flags) atomicAdd(ratio_acc + 3 * k + 2, 1ull);
                else {
                    atomicAdd(ratio_acc + 3 * k, (unsigned long long)(t.k.lo & 0xFFFFFFFFull));
                    atomicAdd(ratio_acc + 3 * k + 1, (unsigned long long)(((uint64_t)t.k.hi << 32) | (t.k.lo >> 32)));
                }
            }
        }
    }
}

// Move.accepted_calls / total_calls are Int in the reference (src/metropolis.jl:145-146); the device counts in 32 bits (two u16
// planes or a u32 array) and CARRIES: before the call that would count step 2^32 the host adds every counter into a 64-bit base
// of its own and restarts the arrays at zero -- once per 2^32 counted steps, days into a run.  n: counters of the array.
template <typename CT>
__global__ __launch_bounds__(AMC_BLOCK) void counter_rebase_kernel(CT* lo, uint16_t* hi, int64_t n, unsigned long long* base)
{
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; i < n; i += stride) {
        unsigned long long v = lo[i];
        lo[i] = 0;
        if (hi) { v |= (unsigned long long)hi[i] << 16; hi[i] = 0; }
        base[i] += v;
    }
}

// Exact integer totals of the per-chain counters (K > 1): out[k] += sum_c a[k][c].  16-byte loads, one atomic per
// block and value (same-address atomics serialise at ~13 ns each: per-wave atomics from a full grid cost 0.2 ms here).
template <typename CT>
__global__ __launch_bounds__(AMC_BLOCK) void counter_totals_kernel(const CT* acc, const CT* tot, const uint16_t* acc_hi,
                                                                    const uint16_t* tot_hi, int64_t n_chains, int64_t m_stride,
                                                                    int n_moves, unsigned long long* out_acc,
                                                                    unsigned long long* out_tot)
{
    __shared__ unsigned long long s_a[AMC_BLOCK / 64], s_t[AMC_BLOCK / 64];
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    const int64_t n_quads = (n_chains + 3) >> 2;              // rows are padded: the last quad is readable
    for (int k = 0; k < n_moves; ++k) {
        unsigned long long sa = 0, st = 0;
        for (int64_t q = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; q < n_quads; q += stride) {
            const int64_t at = (int64_t)k * m_stride + 4 * q;
            const uint4 z4 = uint4{0u, 0u, 0u, 0u};
            uint4 va = load_counter_quad(acc + at);
            // tot has n_moves - 1 rows (the last move's totals are the step count minus the others: the host completes them)
            uint4 vt = (tot && k + 1 < n_moves) ? load_counter_quad(tot + at) : z4;
            if (acc_hi) {                                      // u16 planes in use: full value = low | high << 16
                const uint4 ha = load_counter_quad(acc_hi + at), ht = (tot && k + 1 < n_moves) ? load_counter_quad(tot_hi + at) : z4;
                va = uint4{va.x | (ha.x << 16), va.y | (ha.y << 16), va.z | (ha.z << 16), va.w | (ha.w << 16)};
                vt = uint4{vt.x | (ht.x << 16), vt.y | (ht.y << 16), vt.z | (ht.z << 16), vt.w | (ht.w << 16)};
            }
            const uint32_t a4[4] = {va.x, va.y, va.z, va.w}, t4[4] = {vt.x, vt.y, vt.z, vt.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < n_chains) {                   // the padding behind the last chain holds no counts
                    sa += a4[j];
                    st += t4[j];
                }
        }
        for (int off = 32; off > 0; off >>= 1) {
            sa += __shfl_down(sa, off, 64);
            st += __shfl_down(st, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            s_a[threadIdx.x >> 6] = sa;
            s_t[threadIdx.x >> 6] = st;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long ta = 0, tt = 0;
            for (int w = 0; w < AMC_BLOCK / 64; ++w) { ta += s_a[w]; tt += s_t[w]; }
            if (ta) atomicAdd(out_acc + k, ta);
            if (tt) atomicAdd(out_tot + k, tt);
        }
        __syncthreads();
    }
}

// What the estimator launch needs besides the chains.  Only what varies from launch to launch travels as kernel
// arguments; everything that is fixed for a handle and an estimator configuration (the tail's buffers, optimiser
// settings, learnable-move ids) sits in a PgTail record in device memory that the host rewrites when it changes --
// kernel arguments are loaded into SGPRs at entry and stay live across the sampling loop, and the 350 bytes this
// struct used to have cost the fused sweep + estimator kernel ~70 v_readlane / v_writelane spill instructions per loop trip.
struct PgTail {
    uint32_t* tickets;            // [1 + n_groups], zero between launches
    xs_word* group_sums;          // [NL*4][PG_GROUP][words per column]
    double* out;                  // tail_mode 1: records [n_ranks][NL*4][XS_WORDS], this shard's slot filled, the others zeroed
    double* gd_acc;               // [AMC_MAX_MOVES][5]
    double* ptab_rw;              // == ptab (written by the update)
    int* status;
    double n_samples;
    double n_samples_global;      // ... of all shards (the learning step a launch leaves pending divides by it: pg_apply_pending)
    double* theta_ring;           // [2][AMC_MAX_LEARN]: sigma of the learnable moves as the launches of even / odd estimator steps used it
    int32_t n_moves;
    int32_t rank, n_ranks;        // slot of this shard in `out` (0 of 1 without a communicator)
    int32_t pad_;
    int32_t learn_ids[AMC_MAX_LEARN];
    PgOpts opt;
};

struct PgArgs {
    real_t* x;
    const real_t* beta_arr;
    const double* ptab;
    xs_word* partials;            // [groups of PG_GROUP blocks][NL*4][PG_GROUP][words per column]: block rows
    const PgTail* tail;           // device memory
    int64_t n_chains;
    uint64_t pair0;
    uint64_t t_est;               // estimator call index
    int32_t q_batch;
    int32_t n_learn;
    int32_t learn_ids[AMC_MAX_LEARN];
    uint32_t key0, key1;
    double beta;
    // Tail of the launch (no further launches for the fold's bookkeeping; each tiny launch costs ~5 us plus a ~6 us
    // dependent-launch gap on this part).  tail_mode 0: block rows only; 1: + their total as records in `out`;
    // 2: + gradients_data[k] += gd (estimator.jl:130); 3: + make_step!(::PolicyGradientUpdate).
    int32_t tail_mode;
    // AMC_NP > 1, AMC_NCLASS > 1: a launch takes ONE learnable move (several parameters: its columns fill a row; classes: hipcc
    // 7.2 fails on the unrolled loop over moves with a class switch in it, "illegal VGPR to SGPR copy"), the l_base-th of the
    // estimator call -- the index that, with q, names the sample's draw
    int32_t l_base;
};
// Rewrites the record in stream order: the value travels as a kernel argument (copied at launch), so no host buffer has
// to outlive the call and launches already queued keep reading the old record until they are done.
AMC_KERNEL_LINKAGE __global__ void pg_tail_store_kernel(PgTail value, PgTail* dst)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = value;
}

enum { PG_GROUP = 64 };           // blocks per first-level group of the in-kernel final reduction

// The GradientData fold (gradients.jl:68-76 over estimator.jl:113-129) is a reproducible sum (amc_xsum.h).  Its kind:
// Q, quanta from sigma, for the Gaussian displacement policy on a built-in potential with the model's reward delta^2 -- every
// summand is then bounded by a function of sigma (xs_gd_exponents) --; R, running top, as soon as a script-defined expression
// takes part (potential and reward come together as POT_CUSTOM; a state-dependent width; a whole proposal).
template <int POT>
struct PgKind {
#if defined(AMC_USER_SCALE) || defined(AMC_USER_LOGQ)
    static constexpr bool Q = false;
#else
    static constexpr bool Q = POT != POT_CUSTOM;
#endif
    static constexpr int ROW = Q ? XS_ROW_Q : XS_ROW_R;       // words per column of a block row
};

// One pgmc_estimate sample (gradients.jl:93-109 via sample_gradient_data :117-121), P = 1.
// Leaves x at (x+delta)+(-delta) like the reference (perform_action_cached! :103).
// log_proposal_density (particle_1d.jl:52-54) and its derivative with respect to sigma as ForwardDiff forms it
// (withgrad_log_proposal_density!, gradients.jl:28-33), from the per-move table entries of prepare_params.
struct LogQ { double logq, dlogq; };
__device__ __forceinline__ LogQ log_proposal_density_withgrad(real_t delta, double den, double rden, double logc,
                                                              double dden, double dlhalf)
{
    const double q1 = div_by_const((double)(-(delta * delta)), den, rden);
    LogQ r;
    r.logq = q1 - logc;
    r.dlogq = -div_by_const(q1, den, rden) * dden - dlhalf;
    return r;
}

#ifdef AMC_USER_SCALE
// log_proposal_density at width w = sigma * scale(x) and its sigma-derivative by ForwardDiff's dual rules in the
// function's own order: w = sigma*s -> (w, s); w^2 = w*w -> (w2, s*w + w*s); 2*w2; c/Dual -> -(v/den)*dden; log -> da/a.
__device__ __forceinline__ LogQ log_proposal_density_withgrad_w(real_t delta, double w, double dw)
{
    const double TWO_PI = 0x1.921fb54442d18p+2;
    const double w2 = w * w, dw2 = dw * w + w * dw;
    const double den = 2.0 * w2, dden = 2.0 * dw2;
    const double q1 = ((double)(-(delta * delta))) / den;
    const double a = TWO_PI * w2, da = TWO_PI * dw2;
    LogQ r;
    r.logq = q1 - log_f64(a) / 2.0;
    r.dlogq = -(q1 / den) * dden - (da / a) / 2.0;
    return r;
}

// pgmc_estimate (gradients.jl:93-109) with the state-dependent width: the forward density and gradient at the old state,
// the backward ones at the new state; grad_j takes the forward gradient when alpha == 1, else the backward one (:106).
// g: the sample's four summands (j, grad j, grad logq, g) in the reference's operations.
template <int POT>
__device__ __forceinline__ void pg_sample_scaled(real_t& x, real_t beta, double sigma, double z, double (&g)[4], const double* T)
{
    const double s_f = user_scale(x, T);
    const double w_f = sigma * s_f;
    const real_t delta = (real_t)__builtin_fma(w_f, z, 0.0);         // 0.0 + w_f*z, bit for bit (see propose)
    const LogQ f = log_proposal_density_withgrad_w(delta, w_f, s_f);
    const real_t e1 = potential<POT>(x, T);
    const real_t xn = x + delta;
    const real_t e2 = potential<POT>(xn, T);
    const real_t dlogp = ((-e2) * beta) - ((-e1) * beta);
    const double r = (POT == POT_CUSTOM) ? user_reward(delta, xn, T) : (double)(delta * delta);
    const real_t nd = -delta;
    const double s_b = user_scale(xn, T);
    const LogQ b = log_proposal_density_withgrad_w(nd, sigma * s_b, s_b);
    x = xn + nd;
    const double arg = ((double)dlogp + b.logq) - f.logq;
    double ex = exp_core_f64(arg, T);
    asm volatile("" : "+v"(ex));
    double alpha = (arg >= -708.0) ? ex : ((arg != arg) ? arg : 0.0);
    alpha = (arg >= 0.0) ? 1.0 : alpha;
    const double j = r * alpha;
    g[0] = j;
    g[1] = j * ((alpha == 1.0) ? f.dlogq : b.dlogq);
    g[2] = f.dlogq;
    g[3] = f.dlogq * f.dlogq;
}
#endif

#ifdef AMC_USER_LOGQ
// pgmc_estimate (gradients.jl:93-109) with a script-defined proposal: value and sigma-derivative of the forward density
// at the old state (:97), of the backward density at the new state (:102); grad_j takes the forward gradient when
// alpha == 1, else the backward one (:106).  g: the sample's four summands in the reference's operations.
template <int POT>
__device__ __forceinline__ void pg_sample_script(real_t& x, real_t beta, double sigma, double z, double (&g)[AMC_PG_NC], const double* T, int k,
                                                 const UserTheta& th)
{
    const real_t delta = user_sample(z, x, sigma, T, k, th);
    double d_f[AMC_NP], d_b[AMC_NP];
    const double logq_f = user_logq(delta, x, sigma, T, k, th);
    user_dlogq(delta, x, sigma, T, k, th, d_f);
    const real_t e1 = potential<POT>(x, T);
    const real_t xn = user_perform(x, delta, T, k);
    const real_t e2 = potential<POT>(xn, T);
    const real_t dlogp = ((-e2) * beta) - ((-e1) * beta);
    const double r = (POT == POT_CUSTOM) ? user_reward(delta, xn, T) : (double)(delta * delta);
    const real_t nd = user_invert(delta, xn, T, k);
    const double logq_b = user_logq(nd, xn, sigma, T, k, th);
    user_dlogq(nd, xn, sigma, T, k, th, d_b);
    x = user_perform(xn, nd, T, k);
    const double arg = ((double)dlogp + logq_b) - logq_f;
    double ex = exp_core_f64(arg, T);
    asm volatile("" : "+v"(ex));
    double alpha = (arg >= -708.0) ? ex : ((arg != arg) ? arg : 0.0);
    alpha = (arg >= 0.0) ? 1.0 : alpha;
    const double j = r * alpha;
    // GradientData(j, grad j, grad logq_forward, g = grad logq_forward * grad logq_forward', 1): gradients.jl:104-108.  g is
    // symmetric (a product commutes): its upper triangle, row by row
    g[0] = j;
    int at = 1 + 2 * AMC_NP;
#pragma unroll
    for (int p = 0; p < AMC_NP; ++p) {
        g[1 + p] = j * ((alpha == 1.0) ? d_f[p] : d_b[p]);
        g[1 + AMC_NP + p] = d_f[p];
#pragma unroll
        for (int q = p; q < AMC_NP; ++q) g[at++] = d_f[p] * d_f[q];
    }
}
#endif

// One pgmc_estimate sample of the StandardGaussian policy.  What leaves this function per chain is (a) the position,
// x = (x + delta) + (-delta) in the reference's operations, and (b) four SUMMANDS of GradientData (j, grad j, grad logq, g:
// gradients.jl:104-108), which the reference folds with `+` over all chains in whatever order its reducer takes (foldxl /
// foldxt, estimator.jl:94,113).  The summands follow the ARITHMETIC SPEC of DESIGN.md section 3.6b, which the oracle restates
// operation for operation (its spec-form sample) next to the reference-ordered form (a few ulp apart,
// test_pg_sample_summands_within_ulps): two parts of the reference's rounding sequence that cost 19 of its 46 f64 operations
// per sample are replaced --
//   * alpha = min(1, exp((dlogp + logq_b) - logq_f)) with logq_b == logq_f bit for bit: the detour through logq moves
//     the argument by at most 2^-53 (2|dlogp| + |logq|) -- alpha = exp(min(dlogp, 0)), and log_proposal_density itself (a
//     division by 2 sigma^2 and log(2 pi sigma^2)/2) is not formed at all;
//   * d logq / d sigma, which ForwardDiff forms as -((-(d^2)/den)/den) dden - dlhalf (two IEEE divisions), is the
//     polynomial d^2 (dden/den^2) - dlhalf: one fma with the coefficient split hi + lo (prepare_params) so that no
//     constant's rounding biases the sum, and a second fma for the lo part.
// ACC (kind Q): g[] are the lane's four accumulators; the summands enter them with the last bit of j and of d logq / d sigma
// set (lsb1, amc_xsum.h) -- grad j and g as EXACT products rounded once by the accumulator's fma.  !ACC (kind R: a
// script-defined potential or reward): g[] receives the four summands, products rounded to Float64.
// valid (ACC only; the ragged last trip): a lane without a chain goes through the motions and adds exact zeros.
template <int POT, bool ACC>
__device__ __forceinline__ void pg_sample(real_t& x, real_t beta, double sigma, double c3hi, double c3lo, double c1,
                                          double z, double (&g)[4], const double* T, bool valid = true)
{
    const real_t delta = (real_t)__builtin_fma(sigma, z, 0.0);       // 0.0 + sigma*z, bit for bit (see propose)
    const real_t e1 = potential<POT>(x, T);
    const real_t xn = x + delta;
    const real_t e2 = potential<POT>(xn, T);
    const real_t dlogp = ((-e2) * beta) - ((-e1) * beta);
    const real_t d2t = delta * delta;                              // (delta)^2 in T (particle_1d.jl:43,53)
    const double d2 = (double)d2t;
    const double r = (POT == POT_CUSTOM) ? user_reward(delta, xn, T) : d2;   // reward, particle_1d.jl:42-44 (in T)
    x = xn + (-delta);
    const double dlogq = __builtin_fma(d2, c3hi, __builtin_fma(d2, c3lo, -c1));
    // alpha = min(1, exp(arg)) with Julia's NaN-propagating min: exp(arg >= 0) >= 1 and exp(arg <= 0) <= 1 hold exactly
    // for the spec's exp, so  arg >= 0 -> 1;  -708 <= arg < 0 -> exp(arg);  arg < -708 -> 0;  NaN -> NaN
    // Formed as exp(min(arg, 0)): exp_core(0) == 1.0 exactly (k = 0, r = 0, table entry 2^0), and min maps +inf and every
    // arg > 0 there (one v_min_f64 instead of a compare and two selects).  The two remaining cases -- NaN, which min turns
    // into 0, and arg < -708 -- are looked for with one compare and repaired inside a wave-uniform branch almost no wave takes.
    const double arg = (double)dlogp;
    double alpha = exp_core_f64(__builtin_fmin(arg, 0.0), T);
    asm volatile("" : "+v"(alpha));       // keep the exp unconditional: no divergent branch around it
    const bool rare = !(arg >= -708.0);   // arg < -708 or NaN
    if (__builtin_amdgcn_ballot_w64(rare) != 0ull) {
        asm volatile("" : "+v"(alpha));   // not speculatable: the repair stays inside the branch (it was flattened into selects otherwise)
        alpha = rare ? ((arg != arg) ? arg : 0.0) : alpha;
    }
    const double j = r * alpha;
    if (ACC) {
        // forward and backward gradients coincide for this policy (gradients.jl:106)
        double j1 = __longlong_as_double(__double_as_longlong(j) | 1ll);
        double d1 = __longlong_as_double(__double_as_longlong(dlogq) | 1ll);
        j1 = valid ? j1 : 0.0;                 // (folds away where valid is the literal true)
        d1 = valid ? d1 : 0.0;
        g[0] += j1;
        g[1] = __builtin_fma(j1, d1, g[1]);
        g[2] += d1;
        g[3] = __builtin_fma(d1, d1, g[3]);
    } else {
        g[0] = j;
        g[1] = j * dlogq;
        g[2] = dlogq;
        g[3] = dlogq * dlogq;
    }
}

// gradients_data[k] = gradients_data[k] + gd (estimator.jl:130) from the records the shards have exchanged: recs is
// [n_ranks][nv][XS_WORDS], slot r filled by shard r (the in-place all-reduce(sum) over disjoint slots is a gather).  One
// thread merges the shards' integer totals per column -- in any order: integers -- and rounds once.
__device__ __forceinline__ void pg_merge_slots(const double* recs, int n_ranks, int nv, double* vals)
{
    for (int c = 0; c < nv; ++c) {
        double rec[xs::XS_WORDS];
        xs::rec_clear(rec);
        for (int r = 0; r < n_ranks; ++r) xs::rec_merge(rec, recs + ((size_t)r * nv + c) * xs::XS_WORDS);
        vals[c] = xs::rec_round(rec);
    }
}

AMC_KERNEL_LINKAGE __global__ void pg_accumulate_kernel(const double* recs, int n_ranks, int n_learn, PgIds ids, double n_samples, double* acc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double vals[AMC_MAX_LEARN * 4];
    pg_merge_slots(recs, n_ranks, n_learn * 4, vals);
    for (int l = 0; l < n_learn; ++l) pg_accumulate_one(vals, l, ids.v[l], n_samples, acc);
}

// Both in one launch, for shards connected by a communicator: what follows the in-place all-reduce of the estimator's sums
// when the time step also updates (estimator.jl:130, then update.jl:50-57) -- one tiny launch on the critical path instead of two.
// theta_used: see pg_update_all (the ring slot of the launch before, or nullptr)
AMC_KERNEL_LINKAGE __global__ void pg_accumulate_update_kernel(const double* recs, int n_ranks, double* ptab, double* acc, int n_learn, PgIds ids,
                                                              double n_samples, PgOpts opt, int n_moves, int* status, const double* theta_used)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double vals[AMC_MAX_LEARN * 4];
    pg_merge_slots(recs, n_ranks, n_learn * 4, vals);
    pg_update_all(ptab, acc, n_learn, ids.v, opt, n_moves, status, vals, n_samples, theta_used);
}

// ---- policies with several parameters (handles of amc_create_vector_policy_model with n_params > 1) ----
// gradients_data of a move: [j, grad j [P], grad logq_forward [P], g [P][P] row by row, n] -- GradientData, gradients.jl:41-61 --
// AMC_GD_STRIDE_MAX doubles apart.  The estimator's launch takes one learnable move and leaves 1 + 2P + P(P+1)/2 records (g's
// upper triangle: the outer product of a vector with itself is symmetric bit for bit).  Compiled offline, P at run time.
#define AMC_GD_STRIDE_MAX (2 + 2 * AMC_MAX_NP + AMC_MAX_NP * AMC_MAX_NP)
__host__ __device__ inline int pg_gd_stride(int np) { return 2 + 2 * np + np * np; }
__host__ __device__ inline int pg_n_columns(int np) { return 1 + 2 * np + np * (np + 1) / 2; }

// the 1 + 2P + P(P+1)/2 column totals `vals` of one move, spread out as GradientData's fields (g: both triangles)
__host__ __device__ inline void pg_np_unpack(const double* vals, int np, double* gd)
{
    for (int i = 0; i < 1 + 2 * np; ++i) gd[i] = vals[i];
    int at = 1 + 2 * np;
    for (int p = 0; p < np; ++p)
        for (int q = p; q < np; ++q) {
            gd[1 + 2 * np + p * np + q] = vals[at];
            gd[1 + 2 * np + q * np + p] = vals[at];
            ++at;
        }
}

// gradients_data[k] = gradients_data[k] + gd (estimator.jl:130) for the move lid: recs[n_ranks][columns][XS_WORDS]
AMC_KERNEL_LINKAGE __global__ void pg_accumulate_np_kernel(const double* recs, int n_ranks, int np, int lid, double n_samples, double* acc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double vals[1 + 2 * AMC_MAX_NP + AMC_MAX_NP * (AMC_MAX_NP + 1) / 2], gd[AMC_GD_STRIDE_MAX];
    pg_merge_slots(recs, n_ranks, pg_n_columns(np), vals);
    pg_np_unpack(vals, np, gd);
    double* a = acc + (size_t)lid * AMC_GD_STRIDE_MAX;
    for (int i = 0; i < 1 + 2 * np + np * np; ++i) a[i] += gd[i];
    a[1 + 2 * np + np * np] += n_samples;
}

// inv(A) of a P x P matrix, P <= 4, by Gauss-Jordan elimination with partial pivoting (rows swapped for the largest |pivot| of
// the column, the first of equals), in this exact order of operations -- the tests' CPU restatement is the same sequence.
// (Julia's inv(::Matrix) is LAPACK's getrf + getri: the same pivoting rule, another order of the same eliminations, so the
// two differ by rounding, a few ulp times the condition number; for P = 1 both are 1 / a.)  false: a pivot was 0 or not finite.
__host__ __device__ inline bool pg_inv_small(const double* A, int np, double* inv)
{
    double m[AMC_MAX_NP][2 * AMC_MAX_NP];
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < np; ++j) { m[i][j] = A[i * np + j]; m[i][np + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < np; ++c) {
        int piv = c;
        double best = m[c][c] < 0.0 ? -m[c][c] : m[c][c];
        for (int r = c + 1; r < np; ++r) {
            const double v = m[r][c] < 0.0 ? -m[r][c] : m[r][c];
            if (v > best) { best = v; piv = r; }
        }
        if (!(best > 0.0) || !(best <= 1.7976931348623157e308)) return false;
        if (piv != c)
            for (int j = 0; j < 2 * np; ++j) { const double t = m[c][j]; m[c][j] = m[piv][j]; m[piv][j] = t; }
        const double d = m[c][c];
        for (int j = 0; j < 2 * np; ++j) m[c][j] = m[c][j] / d;
        for (int r = 0; r < np; ++r) {
            if (r == c) continue;
            const double f = m[r][c];
            for (int j = 0; j < 2 * np; ++j) m[r][j] = m[r][j] - f * m[c][j];
        }
    }
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < np; ++j) inv[i * np + j] = m[i][np + j];
    return true;
}

// learning_step! (learning.jl:32-34, 50-52, 77-79, 103-105, 130-134, 160-164) on the averaged GradientData gd of a move with
// np parameters theta: the array expressions of the reference written out left to right -- `eta * inv(F) * v` is
// (eta * inv(F)) * v, a matrix-vector product adds its terms in index order, dot(a, b) likewise.  false: F is singular.
__host__ __device__ inline bool pg_learning_step_np(int kind, double h0, double h1, int np, const double* gd, double* theta)
{
    const double j = gd[0];
    const double* dj = gd + 1;
    const double* dl = gd + 1 + np;
    const double* g = gd + 1 + 2 * np;
    double v[AMC_MAX_NP], step[AMC_MAX_NP];
    double eta = h0;
    const bool baseline = kind == OPT_BLPG || kind == OPT_BLAPG || kind == OPT_BLANPG;
    for (int p = 0; p < np; ++p) v[p] = baseline ? dj[p] - j * dl[p] : dj[p];
    if (kind == OPT_VPG || kind == OPT_BLPG || kind == OPT_BLAPG) {
        if (kind == OPT_BLAPG) {
            double dot = 0.0;
            for (int p = 0; p < np; ++p) dot = p == 0 ? dj[0] * dj[0] : dot + dj[p] * dj[p];
            eta = __builtin_sqrt(2.0 * h0 / (dot + h1));
        }
        for (int p = 0; p < np; ++p) step[p] = eta * v[p];
    } else if (kind == OPT_NPG || kind == OPT_ANPG || kind == OPT_BLANPG) {
        double F[AMC_MAX_NP * AMC_MAX_NP], Fi[AMC_MAX_NP * AMC_MAX_NP];
        for (int a = 0; a < np; ++a)
            for (int b = 0; b < np; ++b) F[a * np + b] = a == b ? g[a * np + b] + h1 * 1.0 : g[a * np + b];     // g + eps I
        if (!pg_inv_small(F, np, Fi)) return false;
        if (kind != OPT_NPG) {
            double w[AMC_MAX_NP];
            for (int a = 0; a < np; ++a) {
                double t = Fi[a * np] * v[0];
                for (int b = 1; b < np; ++b) t = t + Fi[a * np + b] * v[b];
                w[a] = t;
            }
            double dot = v[0] * w[0];
            for (int p = 1; p < np; ++p) dot = dot + v[p] * w[p];
            eta = __builtin_sqrt(2.0 * h0 / dot);
        }
        for (int a = 0; a < np; ++a) {
            double t = (eta * Fi[a * np]) * v[0];
            for (int b = 1; b < np; ++b) t = t + (eta * Fi[a * np + b]) * v[b];
            step[a] = t;
        }
    } else {
        return true;                       // Static
    }
    for (int p = 0; p < np; ++p) theta[p] = theta[p] + step[p];
    return true;
}

// make_step!(::PolicyGradientUpdate) (update.jl:50-57) for the move lid of a pool whose policy has np parameters: average
// (gradients.jl:83-85), learning_step!, initialise_gradient_data.  A step that leaves a parameter non-finite (or meets a
// singular metric) is not applied; status[0] is set instead.
AMC_KERNEL_LINKAGE __global__ void pg_update_np_kernel(double* ptab, double* acc, int np, int lid, int kind, double h0, double h1, int* status)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double* a = acc + (size_t)lid * AMC_GD_STRIDE_MAX;
    const int nf = 1 + 2 * np + np * np;
    const double n = a[nf];
    double gd[AMC_GD_STRIDE_MAX];
    for (int i = 0; i < nf; ++i) gd[i] = a[i] / n;
    double theta[AMC_MAX_NP];
    for (int p = 0; p < np; ++p) theta[p] = ptab[(p == 0 ? PT_SIGMA : PT_THETA1 + p - 1) * AMC_MAX_MOVES + lid];
    bool ok = pg_learning_step_np(kind, h0, h1, np, gd, theta);
    for (int p = 0; p < np; ++p) ok = ok && theta[p] - theta[p] == 0.0;          // finite
    for (int i = 0; i <= nf; ++i) a[i] = 0.0;
    if (ok)
        for (int p = 0; p < np; ++p) ptab[(p == 0 ? PT_SIGMA : PT_THETA1 + p - 1) * AMC_MAX_MOVES + lid] = theta[p];
    else
        status[0] = 1;
}

// The same two steps at the end of the estimator launch of such a policy (one learnable move per launch: pg_estimate_kernel's
// tail, one thread): gradients_data[lid] += gd from the launch's column totals `vals` (estimator.jl:130) and, with `update`,
// make_step!(::PolicyGradientUpdate) right behind it -- the operations of pg_accumulate_np_kernel and pg_update_np_kernel in their
// order, so a fused time step and the three launches it replaces leave the same bits.
__device__ __forceinline__ void pg_tail_np(const double* vals, int np, int lid, double n_samples, bool update, int kind, double h0, double h1,
                                           double* ptab, double* acc, int* status)
{
    double gd[AMC_GD_STRIDE_MAX];
    pg_np_unpack(vals, np, gd);
    double* a = acc + (size_t)lid * AMC_GD_STRIDE_MAX;
    const int nf = 1 + 2 * np + np * np;
    if (!update) {
        for (int i = 0; i < nf; ++i) a[i] += gd[i];
        a[nf] += n_samples;
        return;
    }
    const double n = a[nf] + n_samples;
    for (int i = 0; i < nf; ++i) gd[i] = (a[i] + gd[i]) / n;
    double theta[AMC_MAX_NP];
    for (int p = 0; p < np; ++p) theta[p] = ptab[(p == 0 ? PT_SIGMA : PT_THETA1 + p - 1) * AMC_MAX_MOVES + lid];
    bool ok = pg_learning_step_np(kind, h0, h1, np, gd, theta);
    for (int p = 0; p < np; ++p) ok = ok && theta[p] - theta[p] == 0.0;          // finite
    for (int i = 0; i <= nf; ++i) a[i] = 0.0;
    if (ok)
        for (int p = 0; p < np; ++p) ptab[(p == 0 ? PT_SIGMA : PT_THETA1 + p - 1) * AMC_MAX_MOVES + lid] = theta[p];
    else
        status[0] = 1;
}

// Sum, over rows[n_rows][NV][ROW words] that OTHER blocks wrote (agent-scope loads), of column c: integers, so the order is
// immaterial; thread c of the calling block owns column c.
template <bool Q>
struct PgCol {
    xs::PartQ q;
    xs::PartR r;
    __device__ __forceinline__ void clear() { q = xs::PartQ{xs::i128{0, 0}, 0u}; r = xs::part_r_empty(); }
    __device__ __forceinline__ void add_row(const xs_word* w)        // a row in LDS (pg_sum_rows)
    {
        if (Q) {
            const xs::PartQ b = xs_load_q_row(w);
            q.k = xs::i128_add(q.k, b.k);
            q.flags |= b.flags;
        } else {
            xs::part_r_merge(r, xs_load_r_row(w));
        }
    }
    __device__ __forceinline__ void store_row(xs_word* row) const
    {
        xs_word w[Q ? XS_ROW_Q : XS_ROW_R];
        if (Q) xs_store_q_row(w, q); else xs_store_r_row(w, r);
        // agent scope (sc1: aux 16), 16 bytes per instruction -- rows are 16-byte aligned
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < (Q ? XS_ROW_Q : XS_ROW_R); i += 2) {
            const u32v4_t v = {(uint32_t)w[i], (uint32_t)(w[i] >> 32), (uint32_t)w[i + 1], (uint32_t)(w[i + 1] >> 32)};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, i * 8, 0, 16);
        }
    }
};

// Wave-wide total of the lanes' 128-bit integers (valid in every lane).  The low word travels as two 32-bit limbs, the high
// word whole (|v| < 2^120 here: 64 high words add without overflow); the three sums share the rounds of wave_total_i64.
__device__ __forceinline__ xs::i128 wave_sum_i128(xs::i128 v)
{
    long long l[3] = {(long long)(v.lo & 0xFFFFFFFFull), (long long)(v.lo >> 32), (long long)v.hi};
    wave_total_i64<3>(l);
    xs::i128 r = xs::i128_add(xs::i128_of(l[0]), xs::i128_shl(xs::i128_of(l[1]), 32));
    r.hi = (int64_t)((uint64_t)r.hi + (uint64_t)l[2]);
    return r;
}

// Total of ONE column over n_rows <= 64 rows that OTHER blocks wrote (row r at rows + r stride_words): lane r of the calling
// wave loads row r (agent-scope loads, all in flight together: one far-memory round trip) and the wave adds up -- integers: the
// order is immaterial.  Valid in lane 0.  (A single thread walking 64 rows paid 64 dependent steps: 10 us per launch.)
template <bool Q>
__device__ __forceinline__ PgCol<Q> pg_col_total(const xs_word* rows, int n_rows, int64_t stride_words)
{
    constexpr int ROW = Q ? XS_ROW_Q : XS_ROW_R;
    const int lane = threadIdx.x & 63;
    const bool have = lane < n_rows;
    xs_word w[ROW];
    {
        // agent scope (sc1: aux 16), 16 bytes per instruction, no branch around the loads: the lanes past the last row read row 0
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rows, 0, 0x7fffffff, 0x00020000);
        const int off = (have ? lane : 0) * (int)stride_words * 8;
#pragma unroll
        for (int i = 0; i < ROW; i += 2) {
            const u32v4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + i * 8, 0, 16);
            w[i] = have ? (((xs_word)v.y << 32) | v.x) : 0ull;
            w[i + 1] = have ? (((xs_word)v.w << 32) | v.z) : 0ull;
        }
    }
    PgCol<Q> col;
    col.clear();
    if (Q) {
        const xs::PartQ p = xs_load_q_row(w);
        col.q.k = wave_sum_i128(p.k);
        col.q.flags = __builtin_amdgcn_ballot_w64(p.flags != 0u) != 0ull ? (uint32_t)xs::XS_F_NAN : 0u;
    } else {
        xs::PartR p = have ? xs_load_r_row(w) : xs::part_r_empty();
        const int top = wave_max_i32(p.top);
        xs::part_r_raise(p, top);                      // exact (amc_xsum.h)
        col.r.top = top;
        col.r.k1 = wave_sum_i128(p.k1);
        col.r.k2 = wave_sum_i128(p.k2);
        col.r.flags = (__builtin_amdgcn_ballot_w64((p.flags & xs::XS_F_NAN) != 0u) != 0ull ? (uint32_t)xs::XS_F_NAN : 0u) |
                      (__builtin_amdgcn_ballot_w64((p.flags & xs::XS_F_PINF) != 0u) != 0ull ? (uint32_t)xs::XS_F_PINF : 0u) |
                      (__builtin_amdgcn_ballot_w64((p.flags & xs::XS_F_NINF) != 0u) != 0ull ? (uint32_t)xs::XS_F_NINF : 0u);
    }
    return col;
}

// ---- the learning step a launch leaves PENDING (round 5) -------------------------------------------------------------------
// make_step!(::PolicyGradientUpdate) (update.jl:50-57) needs the sums over ALL chains, so in a launch that also takes the step it
// sits at the very end, behind the second level of the in-kernel reduction, a ticket and a few dependent trips to memory -- on
// the critical path of the next time step, which proposes with the new sigma.  A fused time step that updates every step
// (amc_pgmc_steps) may instead STOP at the group sums (tail_mode PG_TAIL_GROUPS) -- or, between shards, at this shard's records and
// the all-reduce behind them -- and leave the rest to the NEXT launch's prologue: every block adds up the (at most 64) group rows,
// or the shards' records, rounds once and takes learning_step! itself -- the same integers and the same operations in every
// block, so every block proposes with the same sigma' -- while its first load of positions is in flight.  Nothing a block reads
// here is written during the launch: the sigma the previous launch used lives in a ring of two slots (by estimator step parity;
// block 0 leaves sigma' in the other slot), the group rows likewise, gradients_data is zero throughout (the host defers only
// behind an update).  The parameter table itself catches up when something else wants it (pg_resolve_kernel).
// tail_mode: low byte = what the tail does; bits 8-9 = a pending step to take first (PG_PENDING_*); bits 16-23 = the groups the
// launch that left it wrote.
enum { PG_TAIL_GROUPS = 4 };
enum { PG_PENDING_NONE = 0, PG_PENDING_GROUPS = 1, PG_PENDING_RECORDS = 2 };
enum { PG_PARITY_WORDS = PG_GROUP * 32 * XS_ROW_R };      // words of group rows per parity (NV <= 32 columns)
__host__ __device__ inline int pg_tail_of(int tail_mode) { return tail_mode & 0xFF; }
__host__ __device__ inline int pg_pending_of(int tail_mode) { return (tail_mode >> 8) & 3; }
__host__ __device__ inline int pg_pending_groups_of(int tail_mode) { return (tail_mode >> 16) & 0xFF; }

// All threads of the block call (one barrier inside).  s_val[n_learn * 4], s_def[n_learn][DEF_N]: LDS; valid after the caller's
// next barrier.  prev: the parity of the launch that left the step pending; writer: this block records sigma' and the status.
__device__ __forceinline__ void pg_apply_pending(const PgTail* tl, int pending, int prev, int prev_groups, int n_learn, double* s_val,
                                                 double (*s_def)[DEF_N], bool writer)
{
    const int nv = n_learn * 4, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double* ring = tl->theta_ring + prev * AMC_MAX_LEARN;
    for (int c = wave; c < nv; c += AMC_BLOCK / 64) {
        const int e = xs::xs_gd_exponent_of(ring[c >> 2], c & 3);      // the quanta the sums were formed with: from the sigma that launch used
        double val = 0.0;
        if (pending == PG_PENDING_GROUPS) {
            const PgCol<true> col = pg_col_total<true>(tl->group_sums + (int64_t)prev * PG_PARITY_WORDS + (int64_t)c * PG_GROUP * XS_ROW_Q, prev_groups, XS_ROW_Q);
            val = xs::part_q_round(col.q, e);
        } else {
            // the shards' records behind the all-reduce (slot r: shard r's, the all-reduce was a gather): limbs and flags add word by
            // word -- integers below 2^53, exact in any order --, kind and exponent are the same on every shard
            __shared__ double s_rec[AMC_BLOCK / 64][xs::XS_WORDS];
            if (lane < xs::XS_WORDS) {
                double w = 0.0;
                for (int r = 0; r < tl->n_ranks; ++r) w += tl->out[((size_t)r * nv + c) * xs::XS_WORDS + lane];
                s_rec[wave][lane] = lane == 0 ? (double)xs::XS_Q : lane == 1 ? (double)e : lane == 2 ? (w != 0.0 ? (double)xs::XS_F_NAN : 0.0) : w;
            }
            if (lane == 0) val = xs::rec_round(s_rec[wave]);           // (the wave's own LDS writes: in order)
        }
        if (lane == 0) s_val[c] = val;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = pending == PG_PENDING_GROUPS ? tl->n_samples : tl->n_samples_global;
        for (int l = 0; l < n_learn; ++l) {
            const double theta = ring[l];
            // average (gradients.jl:83-85) of gradients_data = 0 + the sums, then learning_step! -- the operations of pg_update_all
            const double j = s_val[4 * l] / n, dj = s_val[4 * l + 1] / n, dlogq = s_val[4 * l + 2] / n, g = s_val[4 * l + 3] / n;
            double next = pg_learning_step(tl->opt.kind[l], tl->opt.h0[l], tl->opt.h1[l], theta, j, dj, dlogq, g);
            if (!(next >= 1e-100 && next <= 1e100)) {       // a step that leaves sigma outside its range (or NaN) is not applied
                next = theta;
                if (writer) tl->status[0] = 1;
            }
            derive_move_params(next, s_def[l]);
            if (writer) tl->theta_ring[(prev ^ 1) * AMC_MAX_LEARN + l] = next;
        }
    }
}

// Brings the parameter table up to date with a pending step (one block; the host launches it before anything but the next fused
// time step reads sigma).
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void pg_resolve_kernel(const PgTail* tl, int pending, int prev, int prev_groups, int n_learn)
{
    __shared__ double s_val[AMC_MAX_LEARN * 4];
    __shared__ double s_def[AMC_MAX_LEARN][DEF_N];
    pg_apply_pending(tl, pending, prev, prev_groups, n_learn, s_val, s_def, true);
    __syncthreads();
    if (threadIdx.x == 0)
        for (int l = 0; l < n_learn; ++l) {
            const int k = tl->learn_ids[l];
            tl->ptab_rw[PT_SIGMA * AMC_MAX_MOVES + k] = s_def[l][DEF_SIGMA];
            prepare_move_params(tl->ptab_rw, k, s_def[l][DEF_SIGMA]);
        }
}

// K3: make_step!(::PolicyGradientEstimator) estimator.jl:111-134, all learnable moves fused.
// SWEEP != 0: the launch first performs ONE make_step!(::Metropolis) of sweepstep = 1 on the pair it has just loaded
// (1: K == 1, 2: K > 1, per-chain counters through the step log in both; 3: K == 1 with the pool-wide counter only) -- run! calls the two algorithms back to
// back at the same t (src/simulation.jl:185-190), and x then makes one HBM round trip for both instead of two.
// Per chain the operations and their order are those of the two separate launches.
// (96-104 VGPRs: 4-5 waves per SIMD.  Capping the registers for 6-8 waves spills and is slower: 104 -> 111 / 155 /
// 194 us per config-5 step, measured.)
// REDUCE (with SWEEP): the launch also leaves the callback sums (sum e, sum x, sum x^2, count; SWEEP == 3: and the pool-wide
// accepted total) of the state it stores -- AFTER the estimator's samples, which is what a callback scheduled at the same t
// observes (run! calls Metropolis, estimator, update, then the callbacks: src/simulation.jl:185-190) -- as one row per block in
// sw.red_partials, like sweep_kernel<.., REDUCE>: a callback after a fused time step needs no pass over x.
template <int POT, int NL, bool BETA, int SWEEP = 0, bool REDUCE = false, bool MIDFLUSH = false>
__global__ __launch_bounds__(AMC_BLOCK) void pg_estimate_kernel(const PgArgs a, const SweepArgs sw)
{
    static_assert(!REDUCE || SWEEP != 0, "the callback sums ride on the fused time step");
    constexpr bool QK = PgKind<POT>::Q;
    constexpr int ROW = PgKind<POT>::ROW;
    constexpr int NC = AMC_PG_NC;               // GradientData columns per learnable move (4 for one parameter)
    static_assert(!QK || NC == 4, "the quanta of xs_gd_exponents are those of the one-parameter Gaussian policy");
    constexpr int NV = NL * NC;
    static_assert(NV <= 32, "the tail's wave 0 owns a row's columns");
    // the callback sums: reproducible (amc_xsum.h); the count of full trips lives on the scalar unit
    constexpr int RNC = RedCols<POT>::NC;
    RLanes<RNC> red;
    __shared__ xs::PartR s_red[REDUCE ? AMC_BLOCK / 64 : 1][RNC];
    if (REDUCE) r_init(red, s_red[threadIdx.x >> 6]);
    // the pool-wide accepted total this block can see before the launch (see sweep_kernel)
    unsigned long long slots_before = 0;
    if (REDUCE && SWEEP == 3 && threadIdx.x == 0)
        for (int sl = (int)blockIdx.x; sl < sw.n_slots; sl += (int)gridDim.x) slots_before += sw.acc_total[sl];
    __shared__ double s_math[TAB_DOUBLES];
    __shared__ double s_tab[SWEEP == 2 ? 5 * AMC_MAX_MOVES : 1];
    __shared__ __attribute__((aligned(16))) uint8_t s_pick[SWEEP == 2 ? AMC_PICK_CELLS : 16];
    // the GradientData fold: wave slots of the columns' integer totals
    __shared__ QSlot s_gq[AMC_BLOCK / 64][QK ? NV : 1];
    __shared__ xs::PartR s_gr[AMC_BLOCK / 64][QK ? 1 : NV];
    if (SWEEP == 2) {
        stage_pick_table(s_pick, sw.pick_tab);
        for (int i = threadIdx.x; i < sw.n_moves; i += AMC_BLOCK) {
            s_tab[0 * AMC_MAX_MOVES + i] = sw.ptab[PT_SIGMA * AMC_MAX_MOVES + i];
            s_tab[1 * AMC_MAX_MOVES + i] = sw.ptab[PT_DEN * AMC_MAX_MOVES + i];
            s_tab[2 * AMC_MAX_MOVES + i] = sw.ptab[PT_LOGC * AMC_MAX_MOVES + i];
            s_tab[3 * AMC_MAX_MOVES + i] = sw.ptab[PT_CUM * AMC_MAX_MOVES + i];
            s_tab[4 * AMC_MAX_MOVES + i] = sw.ptab[PT_RDEN * AMC_MAX_MOVES + i];
        }
    }
    double sw_sigma1 = SWEEP ? sw.ptab[PT_SIGMA * AMC_MAX_MOVES] : 0.0;
    double sw_den1 = SWEEP ? sw.ptab[PT_DEN * AMC_MAX_MOVES] : 0.0;
    double sw_logc1 = SWEEP ? sw.ptab[PT_LOGC * AMC_MAX_MOVES] : 0.0;
    double sw_rden1 = SWEEP ? sw.ptab[PT_RDEN * AMC_MAX_MOVES] : 0.0;
    // script-defined policies: the further parameters of the sweep's only move (K == 1) and of the learnable moves, wave-uniform
    UserTheta sw_th1 = {0.0, 0.0, 0.0};
    UserTheta c_th[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) c_th[l] = UserTheta{0.0, 0.0, 0.0};
#ifdef AMC_USER_LOGQ
    if (SWEEP == 1 || SWEEP == 3) sw_th1 = user_theta_uniform(sw.ptab, 0);
#pragma unroll
    for (int l = 0; l < NL; ++l)
        if (l < a.n_learn) c_th[l] = user_theta_uniform(a.ptab, a.learn_ids[l]);
#endif
    // a learning step the previous launch left pending (pg_apply_pending): wave-uniform
    constexpr bool CAN_DEFER = QK && NL <= 2 && AMC_NP == 1;
    const int pending = CAN_DEFER ? pg_pending_of(a.tail_mode) : 0;
    __shared__ double s_pend_val[CAN_DEFER ? NL * 4 : 1];
    __shared__ double s_def[CAN_DEFER ? NL : 1][DEF_N];
    // ... taken HERE, before the loop's state is set up (few registers are live), with this block's first loads already under way
    real2 x_early = {(real_t)0.0, (real_t)0.0}, b_early = {(real_t)a.beta, (real_t)a.beta};
    if (CAN_DEFER && pending) {
        const int64_t first_pair = (int64_t)blockIdx.x * AMC_BLOCK;
        if (first_pair < ((a.n_chains + 1) >> 1)) {
            x_early = load_pair_block(a.x + 2 * first_pair);
            if (BETA) b_early = load_pair_block(a.beta_arr + 2 * first_pair);
        }
        // sums, rounding, learning_step!: sigma' and what derives from it land in s_def (visible after the barrier that ends
        // stage_math_tables below)
        pg_apply_pending(a.tail, pending, (int)((a.t_est ^ 1ull) & 1ull), pg_pending_groups_of(a.tail_mode), a.n_learn, s_pend_val, s_def, blockIdx.x == 0);
        if (SWEEP == 2 && threadIdx.x == 0) {
#pragma unroll
            for (int l = 0; l < NL; ++l)                  // the pool's table in LDS: the learnable moves' rows from sigma'
                if (l < a.n_learn) {                      // (constant indices into the kernel argument: no private copy of it)
                    const int k = a.learn_ids[l];
                    s_tab[0 * AMC_MAX_MOVES + k] = s_def[CAN_DEFER ? l : 0][DEF_SIGMA];
                    s_tab[1 * AMC_MAX_MOVES + k] = s_def[CAN_DEFER ? l : 0][DEF_DEN];
                    s_tab[2 * AMC_MAX_MOVES + k] = s_def[CAN_DEFER ? l : 0][DEF_LOGC];
                    s_tab[4 * AMC_MAX_MOVES + k] = s_def[CAN_DEFER ? l : 0][DEF_RDEN];
                }
        }
    }
    unsigned long long wave_acc = 0;
    // the Box-Muller polynomials' addend coefficients as live 64-bit VGPR values (amc_math.h, MathK): this kernel has no scalar
    // registers to spare, and a literal addend costs a v_mov_b64 per fma here (18 per pair-iteration before)
    const MathK mk = math_k_pinned();
    // one mc_step! of the pair (mc_sweep! with mc_steps = 1), its step-log byte pair stored right away
    auto mh = [&](real2& xv, real_t b0, real_t b1, uint64_t pair, int64_t p, bool v0, bool v1) {
        uint32_t lw = 0;
        // SWEEP == 3: K == 1 with the pool-wide counter only -- no step log
        pair_steps<POT, SWEEP == 2, (SWEEP != 3 ? AMC_LOG_PACKED : AMC_LOG_NONE), true>(sw, xv, b0, b1, pair, p, v0, v1, s_tab, s_pick, s_math, sw_sigma1, sw_den1,
                                                      sw_rden1, sw_logc1, wave_acc, lw, nullptr, mk, sw_th1);
        if (SWEEP != 3 && v0) store_log_pair<AMC_LOG_PACKED>(sw, sw.log_pos, p, lw);
    };
    const int64_t n_pairs = (a.n_chains + 1) >> 1;
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    // The lane's GradientData accumulators.  Kind Q: g[l][i] starts at 1.5 * 2^(E + 52), E the column's quantum exponent from
    // the move's sigma (xs_gd_exponents: integer arithmetic on the scalar unit; the constants are formed again where a flush
    // needs them instead of staying live across the sampling loop).  Kind R: two accumulators per column and a running top.
    double g[QK ? NL : 1][4];
    RLanes<QK ? 1 : NV> gr;
    // (of sigma only its binade is kept across the sampling loop, one scalar per move: a flush that went back to the table for sigma
    // paid a scalar load and its wait on every block's way out)
    int gd_es[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l)
        gd_es[l] = (QK && l < a.n_learn) ? __builtin_amdgcn_readfirstlane(xs::xs_gd_es(a.ptab[PT_SIGMA * AMC_MAX_MOVES + a.learn_ids[l]])) : 0;
    auto q_constants = [&](int l, uint64_t (&cb)[4]) {
        const xs::GdExponents ge = xs::xs_gd_exponents_es(gd_es[l]);
#pragma unroll
        for (int i = 0; i < 4; ++i) cb[i] = xs::xs_c_bits(ge.e[i]);
    };
    // MIDFLUSH, kind Q: a full f64 accumulator is emptied into an INTEGER of the lane (bits(S) - bits(C), S back to C: eight
    // vector instructions per column, nothing crosses lanes), and the lanes' integers go through the wave once, at the end --
    // a wave-wide flush every 16 samples cost launches with q_batch 4 a sixth of their time.  |k| < 2^51 per emptying: the
    // 64-bit integer takes LANE_FLUSHES of them before it is itself flushed (wave-wide, in halves of 32 bits).
    constexpr bool LANE_INT = QK && MIDFLUSH;
    constexpr int LANE_FLUSHES = 1024;
    // (the integers live in LDS, one word per thread and column -- conflict-free --: eight more registers per move cost the
    // kernel a wave per SIMD)
    __shared__ long long s_kl[LANE_INT ? NL * 4 : 1][LANE_INT ? AMC_BLOCK : 1];
    uint32_t kl_bad = 0u;                 // bit l * 4 + i: column i of move l met a value that is no multiple of its quantum (NaN, Inf)
    int kl_n[LANE_INT ? NL : 1];
#pragma unroll
    for (int l = 0; l < (LANE_INT ? NL : 1); ++l) {
        kl_n[l] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) s_kl[LANE_INT ? l * 4 + i : 0][LANE_INT ? threadIdx.x : 0] = 0ll;
    }
    auto flush_move = [&](int l) {       // kind Q: the four accumulators of learnable move l, and what the lane's integers hold, into the wave's slots
        uint64_t cb[4];
        q_constants(l, cb);
        q_flush<4>(g[QK ? l : 0], cb, s_gq[threadIdx.x >> 6] + (QK ? l * 4 : 0));
        if (LANE_INT) {
            // one column after the other: this sits inside the sampling loop (rarely run), where registers are dear
            for (int i = 0; i < 4; ++i) {
                const long long k = s_kl[LANE_INT ? l * 4 + i : 0][LANE_INT ? threadIdx.x : 0];
                s_kl[LANE_INT ? l * 4 + i : 0][LANE_INT ? threadIdx.x : 0] = 0ll;
                long long h[2] = {k & 0xFFFFFFFFll, k >> 32};
                wave_total_i64<2>(h);
                const bool any = __builtin_amdgcn_ballot_w64(((kl_bad >> (l * 4 + i)) & 1u) != 0u) != 0ull;
                if ((threadIdx.x & 63) == 0) {
                    QSlot* slot = s_gq[threadIdx.x >> 6] + (QK ? l * 4 + i : 0);
                    slot->lo += (unsigned long long)h[0];
                    slot->hi += (unsigned long long)h[1];
                    if (any) slot->flags |= (unsigned int)xs::XS_F_NAN;
                }
            }
            kl_bad &= ~(0xFu << (l * 4));
            kl_n[LANE_INT ? l : 0] = 0;
        }
    };
    auto lane_flush = [&](int l) {       // MIDFLUSH, kind Q: the four accumulators of move l into the lane's integers
        uint64_t cb[4];
        q_constants(l, cb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t b = (uint64_t)__double_as_longlong(g[QK ? l : 0][i]);
            const bool bad = ((b ^ cb[i]) >> 52) != 0ull;
            s_kl[LANE_INT ? l * 4 + i : 0][LANE_INT ? threadIdx.x : 0] += bad ? 0ll : (long long)(b - cb[i]);
            kl_bad |= (bad ? 1u : 0u) << (l * 4 + i);
            g[QK ? l : 0][i] = __longlong_as_double((long long)cb[i]);
        }
        if ((kl_n[LANE_INT ? l : 0] += 1) >= LANE_FLUSHES) flush_move(l);
    };
    auto flush_gd = [&]() {
        if (QK) {
#pragma unroll
            for (int l = 0; l < NL; ++l)
                if (l < a.n_learn) flush_move(l);
        } else {
            r_flush(gr, s_gr[threadIdx.x >> 6]);
        }
    };
    // summands the lane has put into each GradientData accumulator since the last flush (two per sample: both chains)
    constexpr int GD_CAP = QK ? xs::XS_GD_LANE_CAP : xs::XS_LANE_CAP;
    int dep[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) dep[l] = 0;
    // A lane's accumulators take GD_CAP summands between two flushes.  Whether a launch needs a flush before its end at all is
    // known to the host (trips per lane x samples per trip: pg_fits_without_flush, amc_api.hip), and the launches that do are
    // a different instantiation (MIDFLUSH): flush code inside the sampling loop costs the common launch -- ~20 summands per lane
    // at 1e7 chains -- 1 to 2.6 us where it never runs (same-box A/B of both forms, profiles/r04_NOTES.md).
    // Same memory schedule as the sweep kernel: prefetch of the next iteration and the write-through store of
    // the previous one at the START of an iteration; full iterations without per-lane predicates (arrays are
    // padded), the ragged last iteration peeled.
    // per-move constants of the learnable moves: wave-uniform, read once (s_load) before the loop when they fit in
    // SGPRs (NL <= 2: 8 doubles), inside the trip otherwise
    constexpr bool HOIST = NL <= 2;
    double c_sg[NL], c_hi[NL], c_lo[NL], c_c1[NL];
    auto move_consts = [&](int l) {
        const int lid = a.learn_ids[l];
        c_sg[l] = a.ptab[PT_SIGMA * AMC_MAX_MOVES + lid];
        c_hi[l] = a.ptab[PT_C3HI * AMC_MAX_MOVES + lid];
        c_lo[l] = a.ptab[PT_C3LO * AMC_MAX_MOVES + lid];
        c_c1[l] = a.ptab[PT_DLHALF * AMC_MAX_MOVES + lid];
    };
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        c_sg[l] = c_hi[l] = c_lo[l] = c_c1[l] = 0.0;
        if (HOIST && l < a.n_learn) move_consts(l);
    }
    // Every lane of the wave is in every call (the flushes inside use wave-wide operations).  whole_trip: all 256 pairs of the
    // trip exist; otherwise (the ragged last trip) v0 / v1 say which of the lane's two chains do, and the others add zeros.
    auto samples = [&](real2& xv, real_t b0, real_t b1, uint64_t pair, bool v0, bool v1, bool whole_trip) {
        (void)v0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (l < a.n_learn) {
                if (!HOIST) move_consts(l);
                // MIDFLUSH: the samples go in blocks of at most GD_CAP / 2 (two summands each), the check for room in the lane's
                // accumulators BETWEEN the blocks: inside the loop over samples the branch alone cost 4-10 % per sample
                constexpr int QBLOCK = GD_CAP / 2;
                for (int q0 = 0; q0 < a.q_batch; q0 += MIDFLUSH ? QBLOCK : (1 << 30)) {
                const int q1 = MIDFLUSH ? (a.q_batch - q0 < QBLOCK ? a.q_batch : q0 + QBLOCK) : a.q_batch;
                if (MIDFLUSH && (dep[l] += 2 * (q1 - q0)) > GD_CAP) {
                    if (QK) lane_flush(l);
                    else if (l == 0) r_flush(gr, s_gr[threadIdx.x >> 6]);       // all columns at once
                    dep[l] = 2 * (q1 - q0);
                }
                for (int q = q0; q < q1; ++q) {
                    double z0, z1;
#if AMC_NP > 1 || AMC_NCLASS > 1
                    const uint32_t sample_id = (uint32_t)((a.l_base + l) * a.q_batch + q);
#else
                    const uint32_t sample_id = (uint32_t)(l * a.q_batch + q);
#endif
                    box_muller(philox4x32_10(draw_counter(pair, a.t_est, sample_id, STREAM_ESTIMATOR),
                                             a.key0, a.key1),
                               z0, z1, s_math, mk);
#if defined(AMC_USER_LOGQ) || defined(AMC_USER_SCALE)
                    double s0[NC], s1[NC];
#pragma unroll
                    for (int i = 0; i < NC; ++i) s1[i] = 0.0;
#if defined(AMC_USER_LOGQ)
                    const int move_key = user_move_key_uniform(a.learn_ids[l], a.ptab);
                    pg_sample_script<POT>(xv.x, b0, c_sg[l], z0, s0, s_math, move_key, c_th[l]);
                    if (v1) pg_sample_script<POT>(xv.y, b1, c_sg[l], z1, s1, s_math, move_key, c_th[l]);
#else
                    pg_sample_scaled<POT>(xv.x, b0, c_sg[l], z0, s0, s_math);
                    if (v1) pg_sample_scaled<POT>(xv.y, b1, c_sg[l], z1, s1, s_math);
#endif
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        r_deposit(gr, QK ? 0 : l * NC + i, v0 ? s0[i] : 0.0, s_gr[threadIdx.x >> 6]);
                        r_deposit(gr, QK ? 0 : l * NC + i, s1[i], s_gr[threadIdx.x >> 6]);
                    }
#else
                    if (QK) {
                        if (whole_trip) {
                            pg_sample<POT, true>(xv.x, b0, c_sg[l], c_hi[l], c_lo[l], c_c1[l], z0, g[QK ? l : 0], s_math);
                            pg_sample<POT, true>(xv.y, b1, c_sg[l], c_hi[l], c_lo[l], c_c1[l], z1, g[QK ? l : 0], s_math);
                        } else {
                            pg_sample<POT, true>(xv.x, b0, c_sg[l], c_hi[l], c_lo[l], c_c1[l], z0, g[QK ? l : 0], s_math, v0);
                            pg_sample<POT, true>(xv.y, b1, c_sg[l], c_hi[l], c_lo[l], c_c1[l], z1, g[QK ? l : 0], s_math, v1);
                        }
                    } else {
                        double s0[4], s1[4] = {0.0, 0.0, 0.0, 0.0};
                        pg_sample<POT, false>(xv.x, b0, c_sg[l], c_hi[l], c_lo[l], c_c1[l], z0, s0, s_math);
                        if (v1) pg_sample<POT, false>(xv.y, b1, c_sg[l], c_hi[l], c_lo[l], c_c1[l], z1, s1, s_math);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                        r_deposit(gr, QK ? 0 : l * 4 + i, v0 ? s0[i] : 0.0, s_gr[threadIdx.x >> 6]);
                        r_deposit(gr, QK ? 0 : l * 4 + i, s1[i], s_gr[threadIdx.x >> 6]);
                    }
                    }
#endif
                }
                }       // blocks of samples
            }
        }
    };
    auto load_x = [&](int64_t b) -> real2 { return load_pair_block(a.x + 2 * b); };
    auto load_b = [&](int64_t b) -> real2 { return load_pair_block(a.beta_arr + 2 * b); };
    const int64_t first = (int64_t)blockIdx.x * AMC_BLOCK;
    real2 x_nxt = x_early, b_nxt = b_early;
    if (!pending && first < n_pairs) {
        x_nxt = load_x(first);
        if (BETA) b_nxt = load_b(first);
    }
#ifdef AMC_USER_LOGQ
    stage_user_theta(a.ptab);
#endif
    // the accumulators' constants, formed with the first load in flight: they need sigma's value at once -- a scalar load of a table the
    // previous launch's learning step has just rewritten, and a wait
    auto start_accumulators = [&]() {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            uint64_t cb[4] = {0, 0, 0, 0};
            if (l < a.n_learn) q_constants(l, cb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                g[QK ? l : 0][i] = __longlong_as_double((long long)cb[i]);
                if ((threadIdx.x & 63) == 0) q_slot_clear(s_gq[threadIdx.x >> 6][QK ? l * 4 + i : 0]);
            }
        }
    };
    if (QK) {
        if (!pending) start_accumulators();
    } else {
        r_init(gr, s_gr[threadIdx.x >> 6]);
    }
    stage_math_tables(s_math, threadIdx.x, AMC_BLOCK);      // overlaps the latency of the first load
    if (CAN_DEFER && pending) {
        auto uni = [](double v) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(v);
            const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)b), hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(b >> 32));
            return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        };
        if (SWEEP == 1 || SWEEP == 3) {                      // K == 1: the pool's only move is the one that learns
            sw_sigma1 = uni(s_def[0][DEF_SIGMA]); sw_den1 = uni(s_def[0][DEF_DEN]); sw_logc1 = uni(s_def[0][DEF_LOGC]); sw_rden1 = uni(s_def[0][DEF_RDEN]);
        }
#pragma unroll
        for (int l = 0; l < NL; ++l)
            if (l < a.n_learn) {
                c_sg[l] = uni(s_def[CAN_DEFER ? l : 0][DEF_SIGMA]); c_hi[l] = uni(s_def[CAN_DEFER ? l : 0][DEF_C3HI]);
                c_lo[l] = uni(s_def[CAN_DEFER ? l : 0][DEF_C3LO]); c_c1[l] = uni(s_def[CAN_DEFER ? l : 0][DEF_DLHALF]);
                gd_es[l] = __builtin_amdgcn_readfirstlane(xs::xs_gd_es(c_sg[l]));
            }
        start_accumulators();
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): first load drained once (see sweep_kernel)
    real2 x_done = {(real_t)0.0, (real_t)0.0};
    int64_t base_done = -1;
    int64_t base = first;
    for (; base + stride < n_pairs; base += stride) {        // full iterations
        real2 xv = x_nxt;
        const real_t b0 = b_nxt.x, b1 = b_nxt.y;
        x_nxt = load_x(base + stride);
        if (BETA) b_nxt = load_b(base + stride);
        if (base_done >= 0) store_pair_block_writethrough(a.x + 2 * base_done, x_done);
        if (SWEEP) mh(xv, b0, b1, a.pair0 + (uint64_t)(base + threadIdx.x), base + threadIdx.x, true, true);
        samples(xv, b0, b1, a.pair0 + (uint64_t)(base + threadIdx.x), true, true, true);
        if (REDUCE) red_add_pair<POT>(red, xv, true, true, s_math, s_red[threadIdx.x >> 6], sw.red_cols);
        x_done = xv;
        base_done = base;
    }
    if (base < n_pairs) {                                    // last, possibly ragged, iteration
        const int64_t p = base + threadIdx.x;
        const bool v0 = p < n_pairs;
        const bool v1 = v0 && (2 * p + 1 < a.n_chains);
        real2 xv = x_nxt;
        if (base_done >= 0) store_pair_block_writethrough(a.x + 2 * base_done, x_done);
        if (SWEEP) mh(xv, b_nxt.x, b_nxt.y, a.pair0 + (uint64_t)(v0 ? p : 0), p, v0, v1);
        // flushes and kind-R deposits are wave-wide (a raise of the running top is a wave-uniform decision): the lanes past the
        // end go through the motions on a pair nobody stores and add zeros
        samples(xv, b_nxt.x, b_nxt.y, a.pair0 + (uint64_t)(v0 ? p : 0), v0, v1, false);
        if (v0) store_pair_block_writethrough(a.x + 2 * base, xv);   // a lone last chain writes its whole pair: padding
        if (REDUCE) red_add_pair<POT>(red, xv, v0, v1, s_math, s_red[threadIdx.x >> 6], sw.red_cols);
    }
    if (REDUCE)
        red_finish<POT>(red, s_red, sw.red_partials + (int64_t)blockIdx.x * sw.red_stride, sw.red_stride == RED_COMPACT_WORDS, sw.red_cols);
    if (SWEEP == 1 || SWEEP == 3) {      // K == 1: the pool-wide accepted total (counter_totals)
        const unsigned long long t = add_block_accepts(sw.acc_total, wave_acc);
        // this block's slot after this launch (exact in a double below 2^53)
        if (REDUCE && SWEEP == 3 && threadIdx.x == 0)
            sw.red_partials[(int64_t)blockIdx.x * sw.red_stride + (sw.red_stride == RED_COMPACT_WORDS ? (int)RED_COMPACT_SLOT : (int)RED_ROW_SLOT)] =
                (xs_word)__double_as_longlong((double)(slots_before + t));
    }
    // Block totals -> this block's row of partials.  All cross-block traffic of the tail below goes
    // through AGENT-scope relaxed atomic stores / loads (sc1: written through to, and read from, the memory side --
    // the 8 XCDs have private L2s) instead of release/acquire fences: an agent-scope fence is an L2 write-back /
    // invalidate per block, which cost ~70 us per launch over 2048 blocks when it was tried.
    // (Integer atomic ADDS of the block totals straight into the group rows -- no block rows, one level of reading less -- were
    // tried for the kind-Q columns: 12 far atomics per block onto 32-64 addresses each cost the launch 15 us, 67 -> 82 us.)
    flush_gd();
    __syncthreads();
    if ((int)threadIdx.x < a.n_learn * NC) {
        PgCol<QK> col;
        if (QK) col.q = q_block_total<QK ? NV : 1>(s_gq, QK ? (int)threadIdx.x : 0);
        else col.r = r_block_total<QK ? 1 : NV>(s_gr, QK ? 0 : (int)threadIdx.x);
        // layout [group][column][block of the group][ROW words]: the wave that adds a column up reads consecutive rows
        col.store_row(a.partials + (((int64_t)(blockIdx.x / PG_GROUP) * NV + threadIdx.x) * PG_GROUP + blockIdx.x % PG_GROUP) * ROW);
    }
    const int tail_mode = pg_tail_of(a.tail_mode);
    if (tail_mode == 0) return;
    const PgTail* const tl = a.tail;       // loaded here, not at kernel entry (see PgArgs)
    // the pointers the tail works through, fetched together NOW (one far-memory round trip, under the wait for the row's stores)
    // instead of one dependent load at each first use between the tickets
    uint32_t* const tickets = tl->tickets;
    xs_word* const group_sums = tl->group_sums + (int64_t)(a.t_est & 1ull) * PG_PARITY_WORDS;      // group rows by the parity of the estimator step
    double* const tail_out = tl->out;
    const int tail_rank = tl->rank, tail_ranks = tl->n_ranks;
    asm volatile("" ::"v"(tickets), "v"(group_sums), "v"(tail_out), "v"(tail_rank), "v"(tail_ranks));

    // In-kernel final reduction, two levels of "the last one to arrive sums": the last block of each group of
    // PG_GROUP consecutive blocks adds the group's rows, the last group to finish adds the group sums -- integer additions
    // (amc_xsum.h), so neither the grid nor which block happens to be last enters the result.
    // Ordering: the threads that wrote the row wait for those stores (vmcnt(0)), the block meets at a barrier, and only then
    // thread 0 takes its ticket; the block that draws the last ticket reads the rows after the ticket's return value has arrived.
    __shared__ int s_role;
    __shared__ double s_tot[NV];
    const int grp = blockIdx.x / PG_GROUP;
    const int n_groups = (gridDim.x + PG_GROUP - 1) / PG_GROUP;
    const int r0 = grp * PG_GROUP;
    const int n_rows = ((int)gridDim.x - r0 < PG_GROUP) ? (int)gridDim.x - r0 : PG_GROUP;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the row's stores belong to this thread's wave (NV <= 32)
        const uint32_t prev = __hip_atomic_fetch_add(tickets + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_role = (prev == (uint32_t)n_rows - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (s_role != 1) return;
    const int nv = a.n_learn * NC;
    const int wave = threadIdx.x >> 6;
    const bool lane0 = (threadIdx.x & 63) == 0;
    // wave w adds up columns w, w + 4, ...: lanes = the group's rows (PG_GROUP = 64 of them at most)
    for (int c = wave; c < nv; c += AMC_BLOCK / 64) {
        const PgCol<QK> col = pg_col_total<QK>(a.partials + ((int64_t)grp * NV + c) * PG_GROUP * ROW, n_rows, ROW);
        if (lane0) col.store_row(group_sums + ((int64_t)c * PG_GROUP + grp) * ROW);          // [column][group][ROW words]
    }
    if (tail_mode == PG_TAIL_GROUPS) {
        // the rest -- adding up the group rows, rounding, the learning step -- is the next launch's (pg_apply_pending)
        if (threadIdx.x == 0) __hip_atomic_store(tickets + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // the quantum exponent of this wave's first column: the last block rounds with it -- formed here, under the ticket's latency,
    // not between the ticket and the group rows' loads (two dependent loads: 0.4 us on the launch's critical path)
    // (from the sigma this launch USED: the table's, or -- a launch that took a pending step in its prologue -- sigma')
    auto sigma_used = [&](int l) -> double {
        if (!HOIST) return a.ptab[PT_SIGMA * AMC_MAX_MOVES + a.learn_ids[l]];
        double v = c_sg[0];
#pragma unroll
        for (int k = 1; k < NL; ++k) v = l == k ? c_sg[k] : v;
        return v;
    };
    int e_first = 0;
    if (QK && wave < nv) e_first = xs::xs_gd_exponent_of(sigma_used(wave >> 2), wave & 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(tickets + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        const uint32_t prev = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_role = (prev == (uint32_t)n_groups - 1u) ? 2 : 0;
    }
    __syncthreads();
    if (s_role != 2) return;
    for (int c = wave; c < nv; c += AMC_BLOCK / 64) {          // lanes = the groups (at most 64: the host caps the grid)
        const int l = c >> 2, i = c & 3;
        const int e = !QK ? 0 : c == wave ? e_first : xs::xs_gd_exponent_of(sigma_used(l), i);
        const PgCol<QK> col = pg_col_total<QK>(group_sums + (int64_t)c * PG_GROUP * ROW, n_groups, ROW);
        if (lane0 && tail_mode == 1) {
            // records: this shard's slot filled, the other shards' slots zeroed (the all-reduce that follows is a gather)
            for (int r = 0; r < tail_ranks; ++r) {
                double* rec = tail_out + ((size_t)r * nv + c) * xs::XS_WORDS;
                if (r != tail_rank) xs::rec_clear(rec);
                else if (QK) xs::rec_from_q(rec, col.q, e);
                else xs::rec_from_r(rec, col.r);
            }
        } else if (lane0) {
            s_tot[c] = QK ? xs::part_q_round(col.q, e) : xs::part_r_round(col.r);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (loading what the update reads BEFORE the last ticket, in every block that might draw it, was measured: slower,
        // 62.4 against 61.4 us per fused time step -- the loads sit on the path from the group sums to the ticket)
#if AMC_NP > 1
        // a policy with several parameters: the launch's one learnable move (the host sees to it), its 1 + 2P + P(P+1)/2 totals
        if (tail_mode >= 2)
            pg_tail_np(s_tot, AMC_NP, tl->learn_ids[0], tl->n_samples, tail_mode >= 3, tl->opt.kind[0], tl->opt.h0[0], tl->opt.h1[0],
                       tl->ptab_rw, tl->gd_acc, tl->status);
#else
        if (tail_mode >= 3) {
            // (theta: the sigma this launch proposed with -- in registers; the table's, or sigma' of a pending step)
            double th[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) th[l] = HOIST ? c_sg[l] : 0.0;
            pg_update_all(tl->ptab_rw, tl->gd_acc, a.n_learn, tl->learn_ids, tl->opt, tl->n_moves, tl->status, s_tot, tl->n_samples,
                          (HOIST && pending) ? th : nullptr);
        }
        else if (tail_mode >= 2)
            for (int l = 0; l < a.n_learn; ++l) pg_accumulate_one(s_tot, l, tl->learn_ids[l], tl->n_samples, tl->gd_acc);
#endif
    }
}

// Device-side replacement for the per-chain text trajectories (StoreTrajectories, src/algorithms.jl:154-210):
// histogram of the chain positions over half-open bins [lo + i w, lo + (i+1) w), i < n_bins, with
// bin = floor((x - lo) * inv_w) in this exact f64 form; counts[n_bins..n_bins+2] = below lo, >= hi, NaN.
// Per-block LDS histogram (u32 LDS atomics), flushed with one u64 global atomic per non-empty bin.
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void histogram_kernel(const double* x, int64_t n_chains, double lo, double hi,
                                                               double inv_w, int n_bins, unsigned long long* counts)
{
    extern __shared__ unsigned int s_hist[];
    for (int i = threadIdx.x; i < n_bins + 3; i += AMC_BLOCK) s_hist[i] = 0u;
    __syncthreads();
    auto count = [&](double v) {
        int b;
        if (v != v) b = n_bins + 2;
        else if (v < lo) b = n_bins;
        else if (v >= hi) b = n_bins + 1;
        else {
            b = (int)((v - lo) * inv_w);
            b = b < n_bins ? b : n_bins - 1;       // (hi - ulp - lo) * inv_w can round up to n_bins
        }
        atomicAdd(&s_hist[b], 1u);
    };
    // four positions per lane and trip, both 16-byte loads issued before the first is used (one 8-byte load per trip left the
    // pass waiting for latency: ~50 us for 80 MB)
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    const int64_t n_quads = n_chains >> 2;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    for (int64_t q = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; q < n_quads; q += stride) {
        const double2 a = x2[2 * q], b = x2[2 * q + 1];
        count(a.x); count(a.y); count(b.x); count(b.y);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_chains & 3)) count(x[4 * n_quads + threadIdx.x]);
    __syncthreads();
    for (int i = threadIdx.x; i < n_bins + 3; i += AMC_BLOCK)
        if (s_hist[i]) atomicAdd(&counts[i], (unsigned long long)s_hist[i]);
}

// e[c] = potential(x[c]) (Particle.e, particle_1d.jl:13-15,33) for amc_download_state when the host cannot
// evaluate the potential itself (POT_CUSTOM).
template <int POT>
__global__ __launch_bounds__(AMC_BLOCK) void energy_kernel(const real_t* x, int64_t n_chains, double* e)
{
    __shared__ double s_math[TAB_DOUBLES];
    stage_math_tables(s_math, threadIdx.x, AMC_BLOCK);
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t c = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; c < n_chains; c += stride)
        e[c] = (double)potential<POT>(x[c], s_math);
}

// Float32 state (AMC_STATE_F32 builds only): the C ABI moves positions as doubles whatever the state type, so uploads
// are narrowed (T(x), round to nearest even -- what Particle(Float32(x), ...) does) and downloads widened (exact).
// The kernels that only READ positions for host-side consumers (histogram, strided snapshots) run on the widened copy.
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void narrow_state_kernel(const double* in, int64_t n, real_t* out)
{
    const int64_t gs = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; i < n; i += gs) out[i] = (real_t)in[i];
}

AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void widen_state_kernel(const real_t* in, int64_t n, double* out)
{
    const int64_t gs = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; i < n; i += gs) out[i] = (double)in[i];
}

// Strided snapshot: out[i] = x[first + i*stride] (binary stand-in for a subset of trajectory files).
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void gather_strided_kernel(const double* x, int64_t first, int64_t stride,
                                                                    int64_t count, double* out)
{
    const int64_t gs = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; i < count; i += gs) out[i] = x[first + i * stride];
}

// Parity-test hooks (amc_selftest_*): the arithmetic-spec primitives, one value per thread.
AMC_KERNEL_LINKAGE __global__ void selftest_math_kernel(int fn, const double* a, const double* b, double* out, int64_t n)
{
    __shared__ double s_math[TAB_DOUBLES];
    stage_math_tables(s_math, threadIdx.x, blockDim.x);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = a[i];
    double s, c, r = 0.0;
    switch (fn) {
    case 0: r = exp_f64(v, s_math); break;
    case 1: r = log_f64(v); break;
    case 2: sincospi_f64(v, s, c, s_math); r = s; break;
    case 3: sincospi_f64(v, s, c, s_math); r = c; break;
    case 4: r = __builtin_sqrt(v); break;
    case 5: r = v / b[i]; break;
    case 6: r = div_by_const(v, b[i], 1.0 / b[i]); break;
    case 7: r = logbm_f64(v, s_math); break;
    case 8: r = sqrt_radius_f64(v); break;
    case 9:
    case 10:
    case 11: {
        // log_proposal_density(delta = a, sigma = b) / its sigma-derivative, through the code the estimator runs:
        // prepare_params for a one-move pool, then log_proposal_density_withgrad
        double tab[PT_ROWS * AMC_MAX_MOVES];
        tab[PT_SIGMA * AMC_MAX_MOVES] = b[i];
        tab[PT_WEIGHT * AMC_MAX_MOVES] = 1.0;
        prepare_params(tab, 1);
        const LogQ lq = log_proposal_density_withgrad((real_t)v, tab[PT_DEN * AMC_MAX_MOVES], tab[PT_RDEN * AMC_MAX_MOVES],
                                                      tab[PT_LOGC * AMC_MAX_MOVES], tab[PT_DDEN * AMC_MAX_MOVES],
                                                      tab[PT_DLHALF * AMC_MAX_MOVES]);
        // 9: the reference-ordered log density; 10: d logq / d sigma in the reference's order (ForwardDiff's dual rules:
        // what withgrad_log_proposal_density! returns); 11: d logq / d sigma as the estimator kernel forms it (pg_sample)
        const double d2 = (double)((real_t)v * (real_t)v);
        const double dq = __builtin_fma(d2, tab[PT_C3HI * AMC_MAX_MOVES], __builtin_fma(d2, tab[PT_C3LO * AMC_MAX_MOVES], -tab[PT_DLHALF * AMC_MAX_MOVES]));
        r = fn == 9 ? lq.logq : (fn == 10 ? lq.dlogq : dq);
        break;
    }
    default: break;
    }
    out[i] = r;
}

// Exhaustive check of the accept filter's float estimate (accept_filter): for EVERY float t with bit pattern in
// [bits_lo, bits_hi] the relative deviation of v_exp_f32(max(t, -17) * log2e) from the spec's f64 exp(t); the maximum
// over the range lands in out_max_bits (bits of a non-negative double compare like integers).
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(256) void selftest_filter_kernel(uint32_t bits_lo, uint64_t count, unsigned long long* out_max_bits)
{
    __shared__ double s_math[TAB_DOUBLES];
    stage_math_tables(s_math, threadIdx.x, 256);
    double worst = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const float t = __uint_as_float(bits_lo + (uint32_t)i);
        const float ex = __builtin_amdgcn_exp2f(__builtin_fmaxf(t, -17.0f) * 0x1.715476p+0f);
        const double ref = exp_f64((double)__builtin_fmaxf(t, -17.0f), s_math);
        const double rel = __builtin_fabs((double)ex - ref) / ref;
        worst = (rel > worst) ? rel : worst;                  // NaN never enters (ref is finite and positive here)
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(worst, off, 64);
        worst = (o > worst) ? o : worst;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_max_bits, (unsigned long long)__double_as_longlong(worst));
}

// The wave-total primitives (wave_total_i64 by folding, wave_max_u32) on one wave's worth of arbitrary lane values: in is
// [6][64] 64-bit integers; out[0..5] the six totals through wave_total_i64<6>, out[6..7] two of them through <2>, out[8..10]
// three through <3>, out[11] one through <1>, out[12] wave_max_u32 of the low words of row 0; ref[0..5] the totals by the plain
// DPP form of round 4.  The host compares both with its own sums.
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(64) void selftest_wave_totals_kernel(const long long* in, long long* out, long long* ref)
{
    const int lane = threadIdx.x & 63;
    long long v[6], r[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = r[i] = in[i * 64 + lane];
    long long two[2] = {v[4], v[1]}, three[3] = {v[5], v[0], v[2]}, one[1] = {v[3]};
    const uint32_t m = wave_max_u32((uint32_t)(unsigned long long)v[0]);
    wave_total_i64<6>(v);
    wave_total_i64<2>(two);
    wave_total_i64<3>(three);
    wave_total_i64<1>(one);
    wave_total_i64_dpp<6>(r);
    if (lane == 17) {        // any lane: the totals are wave-uniform
#pragma unroll
        for (int i = 0; i < 6; ++i) { out[i] = v[i]; ref[i] = r[i]; }
        out[6] = two[0]; out[7] = two[1];
        out[8] = three[0]; out[9] = three[1]; out[10] = three[2];
        out[11] = one[0];
        out[12] = (long long)m;
    }
}

AMC_KERNEL_LINKAGE __global__ void selftest_philox_kernel(uint32_t key0, uint32_t key1, const uint64_t* pair, const uint64_t* t,
                                       uint32_t draw, uint32_t stream, uint32_t* out4, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 v = philox4x32_10(draw_counter(pair[i], t[i], draw, stream), key0, key1);
    out4[4 * i + 0] = v.x; out4[4 * i + 1] = v.y; out4[4 * i + 2] = v.z; out4[4 * i + 3] = v.w;
}

}  // namespace amc
