/*
 * amc.h -- C ABI of libamc.so: the MI355X (gfx950) many-chain Metropolis engine.
 *
 * This is the drop-in boundary for ONE path of Arianna.jl
 * (TheDisorderedOrganization/MonteCarlo @ 2025-03-02): the sweep
 *     make_step!(::Simulation, ::Metropolis)        src/metropolis.jl:302-309
 *       -> mc_sweep!                                src/metropolis.jl:203-212
 *         -> mc_step!                               src/metropolis.jl:176-190
 * over M independent particle_1d chains (example/particle_1d/particle_1d.jl:9-70),
 * the callback reductions that read its state (callback_energy particle_1d.jl:68-70,
 * callback_acceptance metropolis.jl:319-321) and the policy-gradient estimator that
 * sits on it (src/PolicyGuided/estimator.jl:111-134, gradients.jl:93-121).
 *
 * The reference is pure Julia and has no FFI of its own; these entry points are
 * what an `AriannaAlgorithm` subtype (src/algorithms.jl:6-37) binds with `ccall`
 * -- see INTEGRATION.md for the Julia stub -- and what montecarlo_amd/ (the
 * Python mirror of that plugin interface) binds with ctypes.
 *
 * Conventions
 *  - plain C types only; the caller owns every host buffer for the duration of
 *    the call; the library owns all device memory.
 *  - every function returns 0 on success or a negative amc_status; the message
 *    is available from amc_last_error() (thread-local).  Nothing throws.
 *  - one handle <-> one device <-> one HIP stream.  Calls on one handle must be
 *    serialised by the caller.  amc_sweep / amc_init_uniform are asynchronous on
 *    the handle's stream; every call that returns data to the host synchronises.
 *  - there is NO CPU fallback: without a usable gfx950 device amc_create fails.
 */
#ifndef AMC_H
#define AMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The entry points declared here are the library's WHOLE dynamic symbol table: libamc.so is built with hidden visibility
 * (montecarlo_amd/csrc/Makefile) and these declarations carry the default one. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define AMC_VERSION_MAJOR 0
#define AMC_VERSION_MINOR 1
#define AMC_MAX_MOVES 64        /* K: moves in a pool (Move, metropolis.jl:140-147) */
#define AMC_MAX_LEARN 8         /* learnable moves per estimator call */
#define AMC_MAX_QBATCH 256      /* q_batch_size * n_learn must stay < 4096 (12-bit draw id) */

typedef enum amc_status {
    AMC_OK = 0,
    AMC_ERR_BAD_ARG = -1,
    AMC_ERR_HIP = -2,
    AMC_ERR_OOM = -3,
    AMC_ERR_NO_DEVICE = -4,
    AMC_ERR_STATE = -5,
    AMC_ERR_COMM = -6,
    /* a script-defined model could not be BUILT for a reason that is not the script's: the run-time compiler died (a fatal
     * error inside LLVM), did not come back in time, or could not be started.  It runs in a child process that never touches
     * the GPU (amc_rtc_worker, beside libamc.so), so the caller's process is alive and amc_last_error() holds the compiler's
     * last words.  An expression that does not compile is the caller's error: AMC_ERR_BAD_ARG with the first diagnostic.
     * Reference convention: raise, do not die -- error("No ... is defined"), src/metropolis.jl:35. */
    AMC_ERR_COMPILE = -7
} amc_status;

/* potential(x): free function the driver script defines
 * (harmonic_oscillator/MC_harmonic_oscillator.jl:4).  A user Julia closure cannot
 * cross a C ABI, so the engine offers the particle_1d family by id. */
typedef enum amc_potential {
    AMC_POTENTIAL_HARMONIC = 0,     /* x*x            (reference) */
    AMC_POTENTIAL_DOUBLE_WELL = 1,  /* (x*x - 1)^2    (BASELINE config 3) */
    AMC_POTENTIAL_CUSTOM = 2        /* a C expression in x, compiled for gfx950 at run time: amc_create_custom */
} amc_potential;

/* Particle{T} / Displacement{T} are generic in T <: AbstractFloat (particle_1d.jl:9,26).  Float64 is what the reference's
 * scripts use and the type every parity figure is quoted on.  AMC_DTYPE_F32 keeps x, beta, e, delta and the log-target
 * difference in Float32 exactly where Julia's promotion rules would: the policy parameters, the
 * normal variate, log_proposal_density, the acceptance probability and the uniforms stay Float64.  The state then crosses
 * HBM as 4 + 4 bytes per update; the kernels are the same sources, compiled at run time (hiprtc) on first use.  Host
 * buffers of this ABI stay `double` whatever the state type: uploads are rounded to Float32 (round to nearest even),
 * downloads are exact.  Script-defined expressions then see x (and delta) as `float`: arithmetic against double literals
 * promotes as in Julia (2.0*x is Float64), overloaded calls take their float form (sqrt(x), fabs(x), fma(x, x, c) are
 * Float32 operations, as Julia's would be), and the value is converted to Float32 where the model stores it. */
typedef enum amc_state_dtype {
    AMC_DTYPE_F64 = 0,
    AMC_DTYPE_F32 = 1
} amc_state_dtype;

typedef struct amc_handle amc_handle;

/* Mirrors Metropolis(chains; pool, sweepstep=1, seed=1, ...) metropolis.jl:288-291
 * plus what the chains/pool objects carry on the Julia side. */
typedef struct amc_config {
    uint32_t struct_size;        /* = sizeof(amc_config); ABI guard */
    int32_t  device;             /* HIP device ordinal */
    int64_t  n_chains;           /* chains held by THIS handle (local shard) */
    int64_t  chain_offset;       /* global id of local chain 0; must be even */
    int64_t  n_chains_global;    /* total chains over all shards (>= offset + n_chains) */
    int32_t  potential;          /* amc_potential */
    int32_t  n_moves;            /* K = length(pool) */
    double   beta;               /* Particle.beta (particle_1d.jl:11), shared by all chains
                                    unless amc_upload_state passes a per-chain array */
    const double *sigma;         /* [K] StandardGaussian parameters sigma_k (particle_1d.jl:50),
                                    each in [1e-100, 1e100] */
    const double *weight;        /* [K] Move.weight (metropolis.jl:144); must sum to ~1 */
    uint64_t seed;               /* Metropolis.seed (metropolis.jl:235) -> Philox key */
    int32_t  sweepstep;          /* Metropolis.sweepstep (metropolis.jl:234), >= 1 */
    int32_t  per_chain_counters; /* 1: keep Move.accepted_calls/total_calls per chain (needed
                                    for amc_download_counters; forced on when K > 1).
                                    0 (K = 1 only): keep only the pool-wide accepted total */
    void    *stream;             /* optional hipStream_t to run on; NULL -> library-owned */
    int32_t  state_dtype;        /* amc_state_dtype.  Callers built against the 0.1 layout (struct_size 88, without
                                    this field) get AMC_DTYPE_F64 */
    int32_t  reserved;           /* must be 0 */
} amc_config;
#define AMC_CONFIG_SIZE_V0_1 88u

/* Layout of amc_reduce's output (doubles). */
enum {
    AMC_RED_SUM_E = 0,       /* sum_c e_c                  -> callback_energy = /M      */
    AMC_RED_SUM_X = 1,       /* sum_c x_c                                              */
    AMC_RED_SUM_XX = 2,      /* sum_c x_c^2   (statistic of test/distribution_test.jl:36-37) */
    AMC_RED_COUNT = 3,       /* local chain count as double                             */
    AMC_RED_SUM_RATIO0 = 4,  /* + k: sum_c accepted_ck / total_ck -> callback_acceptance = /M */
    AMC_RED_HEADER = 4
};
/* amc_reduce writes AMC_RED_HEADER + K doubles. */

/* amc_pg_estimate writes 5 doubles per learnable move: GradientData for P = 1
 * (gradients.jl:41-47): j, grad_j, grad_logq_forward, g, n -- SUMS over the local
 * chains x q_batch samples (the `+` fold of gradients.jl:68-76), n as double. */
enum { AMC_GD_J = 0, AMC_GD_GRAD_J = 1, AMC_GD_GRAD_LOGQ = 2, AMC_GD_G = 3, AMC_GD_N = 4,
       AMC_GD_STRIDE = 5 };

const char *amc_last_error(void);
int  amc_version(void);                          /* major*1000 + minor */
int  amc_device_count(int *count);

/* Metropolis(chains; ...) constructor, metropolis.jl:240-267 / :288-291.  Asserts
 * what the reference asserts (weights/parameters well-formed); allocates SoA
 * state x[M] (+ per-chain counters) in HBM.  State starts at x = 0. */
int  amc_create(const amc_config *cfg, amc_handle **out);
int  amc_destroy(amc_handle *h);

/* The reference lets the driver script define `potential(x)` freely (a global Julia function,
 * harmonic_oscillator/MC_harmonic_oscillator.jl:4; docs/src/man/system.md).  amc_create_custom is that hook on
 * the GPU path: cfg->potential = AMC_POTENTIAL_CUSTOM and potential_expr is the body as ONE C expression in the
 * double `x`, e.g. "x*x*x*x - 2.0*x*x + 0.25*x".  The sweep / reduction / estimator kernels are instantiated for
 * it with hiprtc on first use (about 1 s per kernel form, cached per process).  Vocabulary with bit-reproducible
 * results on any IEEE-754 host: + - * / (never contracted into fma), sqrt, fabs, fma, and the engine's own
 * amc_exp / amc_log.  Other device math functions (exp, sin, pow ...) compile too but
 * are only accurate to the device library's ulps.  Allowed characters: printable ASCII except # \ ; { } " ' ` $ @.
 * A malformed expression fails here with the compiler's first diagnostics in amc_last_error().
 * Environment: AMC_RTC_CACHE_DIR=<dir> keeps the compiled code objects on disk (one file per expression and kernel
 * form, keyed by a hash that includes the kernel sources), so later processes skip the compile. */
int  amc_create_custom(const amc_config *cfg, const char *potential_expr, amc_handle **out);
/* The same with the model's `reward(action, system)` (src/PolicyGuided/gradients.jl:20, evaluated at :100 right after
 * perform_action!; particle_1d.jl:42-44 defines delta^2) as a second expression, in `delta` and the NEW position `x`
 * -- what the policy-gradient estimator maximises (E[reward * alpha]).  potential_expr NULL: the built-in potential named
 * by cfg->potential (its own expression, compiled at run time); reward_expr NULL: delta^2. */
int  amc_create_model(const amc_config *cfg, const char *potential_expr, const char *reward_expr, amc_handle **out);
/* The same with a script-defined POLICY of the Gaussian-displacement family.  The reference hands `system` to
 * sample_action! and log_proposal_density (src/metropolis.jl:177-182; particle_1d.jl:52-59), so a policy's width may depend
 * on the state: scale_expr is ONE C expression in the CURRENT position `x`, and the proposal is
 *     delta = rand(rng, Normal(0, sigma_k * scale(x)))
 *     log_proposal_density = -(delta)^2 / (2 (sigma_k scale(x))^2) - log(2 pi (sigma_k scale(x))^2) / 2
 * with the forward density evaluated at the old state and the backward density at the new one, as mc_step! does
 * (metropolis.jl:178,182): the proposal ratio no longer cancels and enters the acceptance.  scale_expr NULL: scale = 1,
 * the reference's StandardGaussian (amc_create_model).  The policy-gradient estimator then takes the forward density
 * and its sigma-derivative at the old state and the backward ones at the new state (gradients.jl:97,102,106). */
int  amc_create_policy_model(const amc_config *cfg, const char *potential_expr, const char *reward_expr,
                             const char *scale_expr, amc_handle **out);
/* A script-defined PROPOSAL in full: the model's own sample_action! and log_proposal_density (the generic functions of
 * src/metropolis.jl:35-62; example/particle_1d/particle_1d.jl:52-59 are the particle_1d model's methods), each as ONE C
 * expression, compiled at run time like the potential:
 *     sample_expr   delta = f(z, x, sigma): the displacement from one standard normal variate `z` (the engine's
 *                   Box-Muller draw of the step), the current position `x` and the move's parameter `sigma`
 *                   -- e.g. a drifted (Langevin) proposal "-sigma*sigma*x + sigma*z"
 *     logq_expr     log q(delta | x, sigma), the log-density of what sample_expr returns (normalisation included where it
 *                   depends on x or sigma); mc_step! evaluates it for the forward action at the old state and for the
 *                   inverted action (-delta) at the new state (metropolis.jl:178,182)
 *     dlogq_expr    d logq / d sigma -- what the reference obtains from ForwardDiff / Enzyme / Zygote
 *                   (src/PolicyGuided/gradients.jl:28-33) and never asks its user for.  OPTIONAL: NULL and the engine
 *                   differentiates logq_expr itself -- the expression evaluated over dual numbers (forward mode,
 *                   ForwardDiff's rules: montecarlo_amd/csrc/amc_dual.h), the parameters carrying the partials, delta and x
 *                   constants, exactly ForwardDiff.gradient(p -> log_proposal_density(action, policy, p, system), parameters).
 *                   Given, the expression is used as given (a hand-derived form may order its operations differently:
 *                   equal to a few ulp, tests/test_autodiff.py).
 * Variables: z, x, sigma, delta; vocabulary as for amc_create_custom.  potential_expr NULL: cfg->potential's built-in.
 * Float32 state (cfg->state_dtype): x and delta are floats in the expressions -- C's usual arithmetic conversions restate Julia's
 * promotion rules for the same text; parameters, z and the densities stay Float64. */
int  amc_create_proposal_model(const amc_config *cfg, const char *potential_expr, const char *reward_expr,
                               const char *sample_expr, const char *logq_expr, const char *dlogq_expr, amc_handle **out);
/* The same for a policy with SEVERAL parameters: Move.parameters is an array in the reference (src/metropolis.jl:140-147),
 * GradientData keeps grad j and grad logq_forward as arrays of its shape and g as their P x P outer product
 * (src/PolicyGuided/gradients.jl:41-61,104-108), and the natural-gradient optimisers invert g + eps I
 * (learning.jl:103-104,130-133,159-163).  n_params = P in [1, AMC_MAX_PARAMS]; the expressions see the move's parameters as
 * theta0 .. theta{P-1} (`sigma` stays a name of theta0), dlogq_exprs[p] = d logq / d theta_p (all P of them, or NULL: logq_expr is
 * differentiated by the engine, see amc_create_proposal_model), perform_expr / invert_expr as in amc_create_action_model (both NULL: the displacement).  E.g. a Gaussian
 * with a learnable drift, delta = theta0 + theta1 z:
 *     sample "theta0 + theta1*z"     logq "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)"
 *     dlogq  { "(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1" }
 * Parameter 0 of every move starts at cfg->sigma[k] (which must pass its range check), the others at 0: set the vector
 * with amc_set_parameters(h, k, theta, P) -- any finite values.  With P > 1:
 *   - every output that holds GradientData has AMC_GD_STRIDE_P(P) = 2 + 2P + P^2 doubles (records) per learnable move:
 *     j, grad j [P], grad logq_forward [P], g [P][P] row by row, n  (P = 1: the five of AMC_GD_*);
 *   - the estimator takes one launch per learnable move (1 + 2P + P(P+1)/2 reproducible column sums: g is symmetric), the
 *     accumulate and update steps ride in that launch's own tail on a single shard (between shards: records, all-reduce, one tiny
 *     launch each); with ONE learnable move amc_pgmc_steps takes the whole time step -- sweep, estimator, learning step -- in one
 *     launch, with several the sweep rides in the first move's launch (amc_pg_route says which route a call gets);
 *   - a learning step that leaves a parameter non-finite, or meets a singular g + eps I, is not applied (as for P = 1:
 *     reported by amc_pg_get_accumulated);
 *   - inv(g + eps I) is Gauss-Jordan elimination with partial pivoting (amc::pg_inv_small) where Julia calls LAPACK's
 *     getrf / getri: equal up to rounding (a few ulp times the condition number), exactly 1 / a for P = 1. */
#define AMC_MAX_PARAMS 4
#define AMC_GD_STRIDE_P(P) (2 + 2 * (P) + (P) * (P))
int  amc_create_vector_policy_model(const amc_config *cfg, int n_params, const char *potential_expr, const char *reward_expr,
                                    const char *sample_expr, const char *logq_expr, const char *const *dlogq_exprs,
                                    const char *perform_expr, const char *invert_expr, amc_handle **out);
/* P and AMC_GD_STRIDE_P(P) of a handle (either pointer may be NULL). */
int  amc_n_params(amc_handle *h, int *n_params, int *gd_stride);
/* Pools that MIX policy and action types.  In the reference every Move carries its own `action` and `policy`
 * (src/metropolis.jl:140-162; the pool's moves need only agree across CHAINS, :249-260), and sample_action! /
 * log_proposal_density / perform_action! / invert_action! dispatch on their types.  Here: up to AMC_MAX_CLASSES expression sets
 * ("classes"), class_of_move[k] in [0, n_classes) names the one move k uses; sample_exprs / logq_exprs have one entry per class,
 * dlogq_exprs one per class with NULL entries, or a NULL array, for the classes whose logq the engine differentiates itself
 * (amc_create_proposal_model), perform_exprs / invert_exprs one per class with NULL entries (or NULL arrays) for the displacement.  One parameter (sigma) per move.  E.g. a plain Gaussian displacement beside a Langevin
 * (drifted) proposal and a scaling action in one pool.  Everything else -- counters, step log, callbacks, estimator, learning
 * steps, sharding -- is that of amc_create_action_model. */
#define AMC_MAX_CLASSES 4
int  amc_create_mixed_model(const amc_config *cfg, int n_classes, const int *class_of_move, const char *potential_expr,
                            const char *reward_expr, const char *const *sample_exprs, const char *const *logq_exprs,
                            const char *const *dlogq_exprs, const char *const *perform_exprs, const char *const *invert_exprs,
                            amc_handle **out);
/* Compile-only check of a script-defined model, amc_potential_check's sibling (no GPU needed): builds the estimator kernel -- the
 * one that uses every expression: sample, logq, its derivative (given, or by forward-mode differentiation where dlogq is NULL),
 * perform / invert, reward -- for the library's target ISA.  n_classes = 1: sample_exprs[0] ... describe the one policy and, with
 * n_params = P > 1, dlogq_exprs holds its P partials (or is NULL); n_classes > 1: one entry per class as for amc_create_mixed_model
 * (n_params = 1).  potential_expr NULL: the harmonic x*x.  Returns AMC_OK, AMC_ERR_BAD_ARG with the first diagnostic when the
 * text does not compile (an operator the dual numbers lack shows up here), AMC_ERR_COMPILE when the compiler itself failed. */
int  amc_model_check(int n_params, int n_classes, const char *potential_expr, const char *reward_expr,
                     const char *const *sample_exprs, const char *const *logq_exprs, const char *const *dlogq_exprs,
                     const char *const *perform_exprs, const char *const *invert_exprs, char *log, int log_capacity);
/* How an estimator call over n_learn learnable moves of this handle would run.  Returns
 *     2   (asked with `fused` != 0 only) the whole time step -- sweep + estimator [+ update] -- is ONE launch (amc_pgmc_steps)
 *     1   one estimator launch takes every learnable move, as the reference's make_step!(::PolicyGradientEstimator) loops over all
 *         of them in one step whatever their policy types (estimator.jl:111-134) -- (more than four learnable moves of the built-in
 *         policy with a q_batch that calls for the kernel's flushing form: launches of four moves, the form a CU holds more blocks of)
 *     0   one launch per learnable move
 *    < 0  an amc_status.
 * Only a pool of several classes can answer 0 for one-parameter policies: its several-move kernel form is asked of the run-time
 * compiler on first use, and where the compiler fails on it (hipcc 7.2 meets a back-end error on some pools; the compiler runs in a
 * child process, AMC_ERR_COMPILE) the calls fall back to the one-move form -- same samples, same sums, same bits.  `why` (may be
 * NULL) then receives the compiler's last words.  The call itself triggers that first build, so it doubles as a warm-up.
 * Policies with several parameters: 0, except n_learn == 1 (1, or 2 for the fused step on a single shard). */
int  amc_pg_route(amc_handle *h, int n_learn, int q_batch, int fused, char *why, int why_capacity);
/* The same with a script-defined ACTION -- the reference's Action interface (src/metropolis.jl:15-119; the displacement's
 * methods are example/particle_1d/particle_1d.jl:30-40) for a one-parameter action on the position:
 *     perform_expr   the position after perform_action!(system, action), from `x` and `delta`      (displacement: x + delta)
 *     invert_expr    the parameter of the inverted action, invert_action!(action, system), from `delta` and the NEW
 *                    position `x`                                                                 (displacement: -delta)
 * A rejected step re-applies the inverted action (perform_action_cached!, metropolis.jl:119,187).  Both NULL: the
 * displacement (= amc_create_proposal_model); they come together.  log_proposal_density must be the density of the move
 * in state space where the action is not a translation (e.g. scaling x -> x exp(delta): logq carries -log|x exp(delta)|). */
int  amc_create_action_model(const amc_config *cfg, const char *potential_expr, const char *reward_expr,
                             const char *sample_expr, const char *logq_expr, const char *dlogq_expr,
                             const char *perform_expr, const char *invert_expr, amc_handle **out);
/* Compile-only check of a potential expression (needs no GPU); the compiler log, if any, is copied to log. */
int  amc_potential_check(const char *potential_expr, char *log, int log_capacity);

/* initialise(): upload chains[c].x (and optionally a per-chain beta array,
 * Particle.beta).  e is not uploaded: e == potential(x) by construction
 * (particle_1d.jl:13-15) and is recomputed on device. */
int  amc_upload_state(amc_handle *h, const double *x, const double *beta_or_null);
/* Synthetic ensemble: x_c = lo + (hi-lo)*u_c, u from the INIT Philox stream keyed by
 * the GLOBAL chain id (MC_harmonic_oscillator.jl:13 uses 4rand(rng)-2). */
int  amc_init_uniform(amc_handle *h, double lo, double hi);
/* finalise(): chains[c].x / chains[c].e back to the host (either may be NULL). */
int  amc_download_state(amc_handle *h, double *x, double *e);
/* pools[c][k].accepted_calls / total_calls, move-major [k*n_chains + c].  Int (Int64) in the reference
 * (src/metropolis.jl:145-146).  The device counts in 32-bit arrays and CARRIES: before the launch that would count step 2^32
 * every counter is added into a 64-bit base of its own and the arrays restart at zero (once per 2^32 counted steps; what the
 * calls below return is base + array), so a run counts on -- up to 2^52 steps, the range in which callback_acceptance's
 * Int / Int is exact.  After the first carry the acceptance ratios are formed by a pass over arrays and bases instead of
 * inside the fold of the step log.  The pool-wide count of a K = 1 handle without per-chain counters is 64-bit.
 * (Storage detail, invisible here: with K <= 4 a counter is two u16 halves in separate arrays, and the high halves take
 * part in the folds only from the call that would count step 65 536 on; environment AMC_WIDE_COUNTERS=1 keeps plain u32.) */
int  amc_download_counters(amc_handle *h, int64_t *accepted, int64_t *total);
/* Pool-wide sums over local chains: accepted[k], total[k] (exact integers). */
int  amc_counter_totals(amc_handle *h, int64_t *accepted, int64_t *total);

/* Resume: restore pools[c][k].accepted_calls / total_calls ([k*n_chains + c], total may be NULL for
 * K = 1) on a handle with per-chain counters -- the total_calls of a chain must add up to the same number of steps on
 * every chain, as they do in the reference, where every chain takes the same steps (mc_sweep!, metropolis.jl:205-210); or, for a K = 1 handle without them, the pool-wide
 * accepted total and the number of counted steps.  Together with amc_upload_state, amc_set_step,
 * amc_set_estimator_step and amc_set_parameters this restores a run exactly (the reference's
 * StoreBackups, src/algorithms.jl:264-303, is write-only and saves neither RNG state nor counters). */
int  amc_upload_counters(amc_handle *h, const int64_t *accepted, const int64_t *total);
int  amc_set_counter_totals(amc_handle *h, const int64_t *accepted, uint64_t steps_counted);

/* Device-side stand-ins for the per-chain text trajectories (StoreTrajectories, src/algorithms.jl:154-210;
 * the pooled positions are what test/distribution_test.jl:33-37 and the density plot consume):
 * histogram over n_bins half-open bins of [lo, hi), bin = floor((x - lo) * (n_bins / (hi - lo)));
 * counts has n_bins + 3 entries: bins, then x < lo, x >= hi, NaN.  Local shard only (sum across shards). */
int  amc_histogram(amc_handle *h, double lo, double hi, int n_bins, uint64_t *counts);
/* The same histogram ACCUMULATED on the device over many calls (a density sampled every few sweeps): _accumulate queues one
 * pass over the positions as of this point of the stream and returns at once; _fetch waits, returns the running counts
 * (layout as above) and, with reset != 0, zeroes them.  The first _accumulate fixes (lo, hi, n_bins) until the next reset. */
int  amc_histogram_accumulate(amc_handle *h, double lo, double hi, int n_bins);
int  amc_histogram_fetch(amc_handle *h, uint64_t *counts, int n_bins, int reset);
/* x[first + i*stride], i < count: a strided binary snapshot of this shard. */
int  amc_download_strided(amc_handle *h, int64_t first, int64_t stride, int64_t count, double *x);

/* n x make_step!(simulation, ::Metropolis) (metropolis.jl:302-309): each is
 * `sweepstep` mc_step!s per chain.  The n*sweepstep steps run fused in one launch
 * (state stays in registers); results are identical to n separate calls. */
int  amc_sweep(amc_handle *h, int64_t n_sweeps);
/* n x make_step!(simulation, ::Metropolis) as n LAUNCHES of one sweep each, queued by one host call: what run!'s time loop issues
 * when every t is observed by nobody in between but the caller wants the state to make its HBM round trip per sweep (the
 * benchmark's definition of a step; src/simulation.jl:184-191 calls make_step! once per t).  Identical to n calls of
 * amc_sweep(h, 1), minus n - 1 crossings of the language boundary. */
int  amc_sweep_launches(amc_handle *h, int64_t n_launches);
/* MH steps done per chain so far (the Philox step index); settable for resume. */
int  amc_get_step(amc_handle *h, uint64_t *t);
int  amc_set_step(amc_handle *h, uint64_t t);
/* estimator make_step! calls done so far (the estimator stream's step index). */
int  amc_get_estimator_step(amc_handle *h, uint64_t *t);
int  amc_set_estimator_step(amc_handle *h, uint64_t t);

/* Reproducible sums.  Everything that crosses chains on this path is a sum -- callback_energy (particle_1d.jl:68-70: mean),
 * callback_acceptance (metropolis.jl:319-321: mean), the GradientData fold (estimator.jl:113-131: reducer(+, ...)) -- and the
 * reference adds Float64s in whatever order its reducer takes.  Here each such sum is defined so that the order of the
 * additions cannot enter the result (montecarlo_amd/csrc/amc_xsum.h): every summand is rounded once to
 * a multiple of a power of two fixed by order-independent facts (the move's sigma; the largest magnitude among the
 * summands), the multiples are added as integers, the total is rounded once to Float64.  The value is therefore the same
 * for every grid, every split of the chains into shards and every number of GPUs, bit for bit.
 * Between shards a partial sum travels as a RECORD of AMC_XSUM_WORDS doubles -- each word an integer below 2^53 in
 * magnitude, so a record survives any f64 channel unchanged; all-zero words are the empty sum.  amc_xsum_merge adds
 * records (into[i] += from[i], i < n_records), amc_xsum_round yields the Float64 of each; both are plain host functions.
 * amc_allreduce_xsum (below, with the communicator) leaves on every shard the merged records of all shards. */
#define AMC_XSUM_WORDS 12
int  amc_xsum_merge(double *into, const double *from, int n_records);
int  amc_xsum_round(const double *records, int n_records, double *out);

/* callback_energy (particle_1d.jl:68-70) / callback_acceptance (metropolis.jl:319-321)
 * / position moments as LOCAL sums (reproducible sums, above): per-block integer partial sums on the device, added up by the
 * host.  out: AMC_RED_HEADER + K doubles.  For one shard these are the final sums; across shards exchange the RECORDS
 * (amc_reduce_end_exact + amc_allreduce_xsum / amc_xsum_merge) -- adding the shards' rounded doubles would make the
 * result depend on the split.  Divide by the global chain count. */
int  amc_reduce(amc_handle *h, double *out);
/* The same reduction split in two so the host need not drain the stream: _begin enqueues the kernels
 * and returns; _end waits for THAT reduction only (sweeps queued after _begin keep
 * running) and returns the values as of _begin.  Up to TWO reductions may be in flight per handle; _end finishes the
 * oldest (a host that writes a callback's row while the next callback's sums are being formed never waits). */
int  amc_reduce_begin(amc_handle *h);
/* n make_step!s followed by amc_reduce_begin of the resulting state.  The sums over x are formed inside the last sweep
 * launch (K <= 4: no second pass over the chains); with per-chain counters the acceptance ratios come from the fold of the
 * step log that follows it. */
int  amc_sweep_reduce_begin(amc_handle *h, int64_t n_sweeps);
int  amc_reduce_end(amc_handle *h, double *out);
/* Which of the sums over x the reductions begun from now on form: callback_energy (particle_1d.jl:68-70) needs sum e alone,
 * the moments of test/distribution_test.jl:36-37 sum x and sum x^2; callback_acceptance (metropolis.jl:319-321) none of the
 * three.  Every sum costs the launch that forms it nine vector instructions per chain pair, so a caller that knows its
 * callbacks names them (default: all).  A sum that was not formed reads NaN (amc_reduce_end) / is an all-zero record
 * (amc_reduce_end_exact). */
enum { AMC_REDUCE_E = 1, AMC_REDUCE_X = 2, AMC_REDUCE_XX = 4, AMC_REDUCE_ALL = 7 };
int  amc_set_reduce_columns(amc_handle *h, int columns);
/* The same as records: (AMC_RED_HEADER + K) * AMC_XSUM_WORDS doubles, column order as above (the count is a record too).
 * steps_counted (may be NULL) receives the MH steps counted per chain at _begin.  On a K = 1 handle without per-chain
 * counters record AMC_RED_SUM_RATIO0 holds the pool-wide accepted TOTAL (an integer); sum_c accepted_c / total_c is its
 * Float64 divided by steps_counted (every chain has the same total_calls). */
int  amc_reduce_end_exact(amc_handle *h, double *records, uint64_t *steps_counted);

/* Move.parameters (shared by all chains, metropolis.jl:252-260): read / replace
 * sigma_k on the device copy, e.g. after learning_step! (update.jl:50-57). */
int  amc_set_parameters(amc_handle *h, int k, const double *p, int n);
int  amc_get_parameters(amc_handle *h, int k, double *p, int n);
/* The same read without making the caller wait for the queued steps (StoreParameters on a schedule, metropolis.jl:433-440,
 * beside device-resident learning steps): _begin queues a copy of sigma_0 .. sigma_{K-1} AS OF THIS POINT of the handle's
 * stream into pinned memory, _end returns them (sigma[K]) once that copy is done -- typically a callback period later, when
 * it long is.  One such read in flight per handle (AMC_ERR_STATE otherwise). */
int  amc_parameters_begin(amc_handle *h);
int  amc_parameters_end(amc_handle *h, double *sigma);
/* ... for a policy with several parameters (amc_create_vector_policy_model): all of them, parameters[k * P + p] of move k;
 * n = K * P.  (amc_parameters_end returns parameter 0 of every move.) */
int  amc_parameters_end_all(amc_handle *h, double *parameters, int n);

/* make_step!(simulation, ::PolicyGradientEstimator) (estimator.jl:111-134) for the
 * moves learn_ids[0..n_learn) (0-based), q_batch samples per chain per move, in the
 * reference's order (move-major, then sample).  Like the reference, every sample
 * leaves x at (x+delta)-delta (gradients.jl:98,103).  out: n_learn*AMC_GD_STRIDE. */
int  amc_pg_estimate(amc_handle *h, int n_learn, const int *learn_ids, int q_batch,
                     double *out);
/* The same fold as records (reproducible sums): n_learn * AMC_GD_STRIDE * AMC_XSUM_WORDS doubles, what shards exchange. */
int  amc_pg_estimate_exact(amc_handle *h, int n_learn, const int *learn_ids, int q_batch,
                           double *records);

/* Device-resident variant of the estimator/update pair (no host round trip per step):
 *   amc_pg_accumulate  = make_step!(::PolicyGradientEstimator): the same kernel as amc_pg_estimate, then
 *                        (if amc_comm_init was called) ONE in-place RCCL all-reduce on the engine's stream that gathers
 *                        the shards' records (4 n_learn AMC_XSUM_WORDS doubles per shard), then the integer merge, one
 *                        rounding and gradients_data[k] += gd on the device (estimator.jl:130): the same bits on every
 *                        shard as on a single shard holding all the chains.
 *   amc_pg_update      = make_step!(::PolicyGradientUpdate) (update.jl:50-57): average, learning_step! for
 *                        P = 1 with optimiser ids below and their two hyper-parameters (eta or delta, eps_id),
 *                        reset, refresh the device copy of sigma and its derived table.  Asynchronous.
 *   amc_pg_get_accumulated  reads the running sums (j, grad_j, grad_logq, g, n) per move (synchronises);
 *                        returns AMC_ERR_STATE if some step produced a sigma outside [1e-100, 1e100]. */
typedef enum amc_optimiser {      /* src/PolicyGuided/learning.jl:16-164 */
    AMC_OPT_STATIC = 0, AMC_OPT_VPG = 1, AMC_OPT_BLPG = 2, AMC_OPT_BLAPG = 3,
    AMC_OPT_NPG = 4, AMC_OPT_ANPG = 5, AMC_OPT_BLANPG = 6
} amc_optimiser;
int  amc_pg_accumulate(amc_handle *h, int n_learn, const int *learn_ids, int q_batch);
int  amc_pg_update(amc_handle *h, int n_learn, const int *learn_ids, const int *optimiser,
                   const double *hyper0, const double *hyper1);
int  amc_pg_get_accumulated(amc_handle *h, int n_learn, const int *learn_ids, double *out);
/* Resume: replace the running sums of the moves learn_ids[] by in[n_learn*AMC_GD_STRIDE] (what amc_pg_get_accumulated
 * returned when the run was checkpointed) -- gradients_data of estimator.jl:84 is part of the state whenever
 * PolicyGradientUpdate is scheduled less often than the estimator. */
int  amc_pg_set_accumulated(amc_handle *h, int n_learn, const int *learn_ids, const double *in);
/* n_steps x [ make_step!(::Metropolis); make_step!(::PolicyGradientEstimator); make_step!(::PolicyGradientUpdate)
 * if do_update ] -- the three algorithms run! calls back to back at one time step (src/simulation.jl:185-190,
 * PGMC_harmonic_oscillator.jl:24-33) -- enqueued by ONE host call: amc_sweep(h, 1), amc_pg_accumulate, amc_pg_update
 * in that order per step, results identical to the separate calls.  For hosts whose per-call cost (Julia ccall,
 * ctypes) would otherwise exceed the ~0.1 ms of device work per step. */
int  amc_pgmc_steps(amc_handle *h, int64_t n_steps, int n_learn, const int *learn_ids, int q_batch,
                    int do_update, const int *optimiser, const double *hyper0, const double *hyper1);
/* The same followed by amc_reduce_begin of the state the last time step leaves -- what callback_energy / callback_acceptance
 * scheduled at that t observe: run! calls them after the three algorithms (src/simulation.jl:185-190;
 * PGMC_harmonic_oscillator.jl:34 lists StoreCallbacks behind them).  When the time step is ONE launch (sweepstep = 1, at most
 * two learnable moves, K <= 4) that launch also forms the sums over x, and the acceptance ratios come from the fold of the
 * step log that follows it: no pass re-reads the chains, and amc_reduce_end returns the values as of this call however
 * many time steps have been queued since. */
int  amc_pgmc_steps_reduce_begin(amc_handle *h, int64_t n_steps, int n_learn, const int *learn_ids, int q_batch,
                                 int do_update, const int *optimiser, const double *hyper0, const double *hyper1);

int  amc_sync(amc_handle *h);
/* hipStream_t the handle launches on (for event timing / graph capture by the host). */
int  amc_get_stream(amc_handle *h, void **stream);
/* HIP-event timing on the handle's stream: begin records an event, end records a second
 * one, waits for it and returns the elapsed device time between the two in ms. */
int  amc_timing_begin(amc_handle *h);
int  amc_timing_end(amc_handle *h, double *elapsed_ms);
/* Optional: records the END event now, without waiting (amc_timing_end then only waits and reads), so that a host can
 * put its own synchronisation point between the two without paying for two blocking waits. */
int  amc_timing_mark(amc_handle *h);

/* Cross-shard sum over RCCL (xGMI) for hosts without torch.distributed (Julia):
 * nccl_unique_id is the 128-byte ncclUniqueId made by amc_comm_unique_id on rank 0
 * and shipped to the other ranks by the caller. */
/* RCCL is resolved with dlopen at the first of these calls (librccl.so.1, as any RCCL the host process has loaded already);
 * AMC_RCCL_LIBRARY=<file> names the library to use instead -- a site's own build, or the shared-memory stand-in of the
 * tests that lets several ranks share one GPU (tests/aux/fake_rccl.c). */
int  amc_comm_unique_id(void *id128);
int  amc_comm_init(amc_handle *h, int rank, int n_ranks, const void *id128);
/* The sum runs on a stream of its own (the host waits for it without draining the sweeps queued on the engine's stream),
 * ordered behind the collectives the estimator has queued on the engine's stream with the same communicator. */
int  amc_allreduce_sum(amc_handle *h, double *buf, int n);
/* records[i] <- the merged records i of ALL shards (reproducible sums): one ncclAllReduce in which every shard fills its own
 * slot of a zeroed buffer -- a gather that is exact whatever order RCCL adds in -- then amc_xsum_merge over the slots.
 * Every shard ends with the same bits, those a single shard holding all the chains would have.  Identity without a communicator. */
int  amc_allreduce_xsum(amc_handle *h, double *records, int n_records);
/* *forced = 1 when the environment variable AMC_RCCL_LIBRARY replaced librccl in this process (a site's own build, or the
 * tests' shared-memory stand-in): a multi-GPU figure obtained that way has to say so. */
int  amc_comm_library_forced(int *forced);
/* Drop the communicator: the handle is a single shard again (a later amc_comm_init may give it a new one).  For ranks whose
 * amc_comm_init succeeded in a launch where another rank's failed. */
int  amc_comm_destroy(amc_handle *h);
/* What the communicator says about itself (ncclCommCount, ncclCommUserRank), the RCCL version (ncclGetVersion) and the file
 * the RCCL symbols were resolved from -- so that a multi-GPU result can show that RCCL really spanned N ranks and which
 * library carried it.  Without a communicator: 1 rank, rank 0, version 0, empty path.  Any output pointer may be NULL. */
int  amc_comm_info(amc_handle *h, int *n_ranks, int *rank, int *rccl_version, char *librccl_path, int path_capacity);
/* hipRuntimeGetVersion of the HIP runtime this library is bound to in this process, and the file it was loaded from: a
 * process that imported torch first binds torch's bundled runtime, a bare one the system's (/opt/rocm). */
int  amc_runtime_info(int *hip_runtime_version, char *hip_runtime_path, int path_capacity);

/* Parity-test hooks: evaluate arithmetic-spec primitives on the device.
 * fn: 0 exp(a), 1 log(a), 2 sinpi(a), 3 cospi(a), 4 sqrt(a), 5 a/b (IEEE), 6 a/b by the kernel's
 * reciprocal-correction sequence (must equal 5 bit for bit), 7 the Box-Muller log, 8 the Box-Muller
 * radius sqrt (must equal 4 bit for bit on {0} U [2^-52, 80]); 9 log_proposal_density(delta = a, sigma = b)
 * (particle_1d.jl:52-54) and 10 its derivative with respect to sigma (withgrad_log_proposal_density!, gradients.jl:28-33),
 * both in the reference's operation order (ForwardDiff's dual rules written out) -- the values test/ad_backends_test.jl:31-32
 * pins, and what the host-side withgrad_log_proposal_density returns; 11 the same derivative as the estimator KERNEL forms
 * it (one fma chain with a split coefficient, within a few ulp of 10: the estimator's summands are tolerance-matched,
 * DESIGN.md section 3.6b).  Host buffers. */
int  amc_selftest_math(int device, int fn, const double *a, const double *b_or_null,
                       double *out, int64_t n);
/* The sweep kernel settles most accept decisions from a float estimate of exp(dlogp) whose error interval is rigorous
 * (DESIGN.md section 3.6); this hook walks EVERY float t in [t_to, t_from] (t_to <= t_from <= 0) on the device and
 * returns the largest relative deviation of that estimate from the arithmetic spec's f64 exp(t). */
int  amc_selftest_accept_filter(int device, float t_from, float t_to, double *max_rel_err);
/* out4[i] = Philox4x32-10(key = seed, counter of draw (pair[i], t[i], draw, stream)). */
int  amc_selftest_philox(int device, uint64_t seed, const uint64_t *pair, const uint64_t *t,
                         uint32_t draw, uint32_t stream, uint32_t *out4, int64_t n);

/* The kernels' wave-wide integer totals (every reproducible sum ends in one): values is [6][64] -- six 64-bit integers per lane
 * of one wavefront --, totals[0..5] their sums through the six-value form, [6..7] values 4, 1 through the two-value form,
 * [8..10] values 5, 0, 2 through the three-value form, [11] value 3 alone, [12] the largest low 32-bit word of row 0;
 * totals_plain[0..5]: the six sums by plain DPP rounds.  Sums are modulo 2^64. */
int  amc_selftest_wave_totals(int device, const int64_t *values, int64_t *totals, int64_t *totals_plain);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
