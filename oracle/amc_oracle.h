/*
 * amc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's many-chain Metropolis hot path
 * (Arianna.jl @ 2025-03-02: src/metropolis.jl, example/particle_1d/particle_1d.jl,
 * src/PolicyGuided/{gradients,estimator,update,learning}.jl, src/simulation.jl
 * schedule helpers).  Every function in amc_oracle.c cites the reference
 * file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (montecarlo_amd/, libamc.so) never does.
 *
 * PARITY STATUS: "parity unpinned" at the bit level.  The reference is Julia
 * (not runnable in this image) and none of its tests pins an RNG bitstream,
 * a trajectory or an accept count.  The oracle is pinned against everything
 * the reference's tests DO hold for this path:
 *   - test/ad_backends_test.jl:27-32  (logq, dlogq/dsigma closed form, 1e-10)
 *   - test/distribution_test.jl:36-37 (mean, std of sampled x, atol 1e-3)
 *   - test/pgmc_test.jl:45,50         (<e> = 0.25 +- 0.05, sigma* = 1.2 +- 0.2,
 *                                      Static optimiser leaves sigma untouched)
 * and against published known-answer vectors of its third-party pieces
 * (Random123 Philox4x32-10 KATs; rocRAND's uniform / Box-Muller maps).
 *
 * RNG: the reference uses one Xoshiro per chain, seeded seed+c-1
 * (metropolis.jl:262-263), through its public `R=` plug point.  Julia's
 * Xoshiro seeding and ziggurat randn cannot be reproduced without Julia, so
 * oracle and product both use the counter-based draw schedule of DESIGN.md §3
 * (Philox4x32-10 keyed by seed, counter = (step, draw, stream, chain pair)).
 */
#ifndef AMC_ORACLE_H
#define AMC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { AMO_POT_HARMONIC = 0, AMO_POT_DOUBLE_WELL = 1, AMO_POT_CUSTOM = 2 };
enum { AMO_STREAM_INIT = 0, AMO_STREAM_METROPOLIS = 1, AMO_STREAM_ESTIMATOR = 2 };
enum { AMO_DRAW_NORMAL = 0, AMO_DRAW_ACCEPT = 1 };
enum {
    AMO_OPT_STATIC = 0, AMO_OPT_VPG = 1, AMO_OPT_BLPG = 2, AMO_OPT_BLAPG = 3,
    AMO_OPT_NPG = 4, AMO_OPT_ANPG = 5, AMO_OPT_BLANPG = 6
};

typedef struct amo_sim amo_sim;

/* ---- arithmetic spec primitives (exported for known-answer tests) ---- */
void   amo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void   amo_counter(uint64_t pair, uint64_t t, uint32_t draw, uint32_t stream, uint32_t ctr[4]);
double amo_exp(double x);
double amo_log(double x);
void   amo_sincospi(double w, double *s, double *c);
void   amo_box_muller(const uint32_t v[4], double z[2]);
double amo_logbm(double u);                         /* table-driven log for the Box-Muller radius */
double amo_uniform_co(uint32_t lo, uint32_t hi);    /* [0,1), 52 bits: Julia's rand(Float64) construction */
double amo_uniform_oc(uint32_t lo, uint32_t hi);    /* (0,1] */
double amo_angle28(uint32_t w);                      /* (0,2], the top 28 bits of the word */
uint32_t amo_spare_accept12(const uint32_t v[4], int half);   /* spare bits of a normal draw: lead the chain's accept uniform */
uint32_t amo_spare_pick12(const uint32_t v[4], int half);     /* ... and its move-pick uniform */
double amo_uniform_accept(uint32_t accept12, uint32_t lo, uint32_t hi);   /* [0,1), 52 bits: top 12 from the normal draw */
double amo_uniform_pick(uint32_t pick12, uint32_t lo);       /* [0,1), 36 bits: categorical move pick */
void   amo_set_custom_proposal(double (*sample)(double, double, double), double (*logq)(double, double, double),
                               double (*dlogq)(double, double, double));   /* script-defined sample_action! / log_proposal_density */
void   amo_set_custom_action(double (*perform)(double, double), double (*invert)(double, double));   /* script-defined perform_action! / invert_action! */
double amo_potential(int pot, double x);
/* AMO_POT_CUSTOM: `potential` is a free GLOBAL function of the driver script in the reference
 * (MC_harmonic_oscillator.jl:4); the tests install the same C expression they hand to amc_create_custom,
 * compiled by gcc (-ffp-contract=off), here.  Process-global, like the reference's. */
void   amo_set_custom_potential(double (*fn)(double));
void   amo_set_custom_reward(double (*fn)(double delta, double x_new));   /* NULL: delta^2 (particle_1d.jl:42-44) */
/* Float32 state (Particle{Float32}, particle_1d.jl:9; promotion rules in amc_oracle.c): switch a simulation over
 * (rounds x, beta to Float32), the single-step form, and the script-defined functions in their Float32 form. */
void   amo_set_state_f32(amo_sim *s, int on);
int    amo_mc_step_explicit_f32(int pot, float beta, double sigma, double z, double u, float *x, float *e);
float  amo_potential_f32(int pot, float x);
void   amo_set_custom_potential_f32(float (*fn)(float));
void   amo_set_custom_reward_f32(double (*fn)(float delta, float x_new));
/* State-dependent proposal width sigma * scale(x) (a script-defined policy of the Gaussian-displacement family):
 * forward density at the old state, backward density at the new one.  NULL: scale == 1 (StandardGaussian). */
void   amo_set_custom_scale(double (*fn)(double x));
void   amo_set_custom_scale_f32(float (*fn)(float x));
double amo_log_proposal_density(double delta, double sigma);
double amo_grad_log_proposal_density(double delta, double sigma);
int    amo_categorical(const double *weights, int K, double r);
/* one mc_step! on a lone particle with explicit draws (z, u): returns 1/0, updates x,e */
int    amo_mc_step_explicit(int pot, double beta, double sigma, double z, double u,
                            double *x, double *e);

/* ---- simulation object: Vector{Particle} + per-chain pools + Metropolis ---- */
amo_sim *amo_create(int64_t n_chains, int64_t chain_offset, int potential, double beta,
                    int n_moves, const double *sigma, const double *weight,
                    uint64_t seed, int sweepstep);
void   amo_destroy(amo_sim *s);
void   amo_set_x(amo_sim *s, const double *x);            /* Particle(x, beta) ctor */
void   amo_set_beta(amo_sim *s, const double *beta);      /* per-chain beta */
void   amo_init_uniform(amo_sim *s, double lo, double hi);
void   amo_get_state(const amo_sim *s, double *x, double *e);
void   amo_get_counters(const amo_sim *s, int64_t *accepted, int64_t *total); /* [k*M + c] */
void   amo_set_sigma(amo_sim *s, int k, double sigma);
/* a policy with several parameters (Move.parameters as an array; amc_create_vector_policy_model is the engine's entry) */
void   amo_set_vector_policy(int np, double (*sample)(double, double, const double *), double (*logq)(double, double, const double *),
                             void (*dlogq)(double, double, const double *, double *));
/* pools that mix policy / action types: one class of functions per move (sample / logq / dlogq (z|delta, x, sigma), perform(x, delta),
 * invert(delta, x); dlogq, perform, invert arrays or entries may be 0); n_classes <= 1 restores the one-policy forms */
void   amo_set_policy_classes(int n_classes, const int *class_of_move, int n_moves, void *const *sample, void *const *logq,
                              void *const *dlogq, void *const *perform, void *const *invert);
void   amo_set_theta(amo_sim *s, int k, int p, double v);
double amo_get_theta(const amo_sim *s, int k, int p);
void   amo_pg_estimate_records_vec(amo_sim *s, int n_learn, const int *learn_ids, int q_batch, double *recs);
int    amo_inv_small(const double *A, int np, double *inv);
int    amo_learning_step_vec(int opt, double h0, double h1, int np, const double *gd, double *theta);
double amo_get_sigma(const amo_sim *s, int k);
uint64_t amo_get_step(const amo_sim *s);
void   amo_set_step(amo_sim *s, uint64_t t);

void   amo_make_step(amo_sim *s, int n_threads);          /* make_step!(::Metropolis) */
void   amo_make_steps(amo_sim *s, int64_t n, int n_threads);

double amo_callback_energy(const amo_sim *s);
void   amo_callback_acceptance(const amo_sim *s, double *out /* K */);
void   amo_moments(const amo_sim *s, double out[2]);      /* sum x, sum x^2 */

/* make_step!(::PolicyGradientEstimator): out[n_learn][5] = (j, dj, dlogq_fwd, g, n); the fold is a reproducible sum
 * (amc_oracle.c "Reproducible sums": the cross-chain sums are defined independently of the order of the additions). */
void   amo_pg_estimate(amo_sim *s, int n_learn, const int *learn_ids, int q_batch,
                       double *out);
/* ... as records (n_learn x 5 x 12 doubles: what shards exchange), and with the reference-ordered summands folded left
 * to right in Float64 (one of the orders the reference's reducer may take) */
void   amo_pg_estimate_records(amo_sim *s, int n_learn, const int *learn_ids, int q_batch, double *recs);
void   amo_pg_estimate_plain(amo_sim *s, int n_learn, const int *learn_ids, int q_batch, double *out);
/* one sample's summands (j, grad j, grad logq, g) of the Gaussian policy: by the arithmetic spec the engine follows
 *  and in the reference's operation order (gradients.jl:93-109); *x is updated like the chain's */
void   amo_pg_summands_spec(int pot, double beta, double sigma, double z, double *x, double out[4]);
void   amo_pg_summands_reference(int pot, double beta, double sigma, double z, double *x, double out[4]);

/* ---- reproducible sums: the definition applied to explicit operands, records of 12 doubles (include/amc.h) ---- */
void   amo_xsum_q(const double *v, int64_t n, int e, double *rec);
void   amo_xsum_q_product(const double *x, const double *y, int64_t n, int e, double *rec);
void   amo_xsum_r(const double *v, int64_t n, double *rec);
void   amo_xsum_merge(double *into, const double *from);
double amo_xsum_round(const double *rec);
void   amo_gd_exponents(double sigma, int e[4]);
/* the callbacks' sums as records, layout of amc_reduce: sum e, sum x, sum x^2, count, sum_c acc/tot per move */
void   amo_callback_records(const amo_sim *s, double *recs);
double amo_callback_energy_plain(const amo_sim *s);
void   amo_callback_acceptance_plain(const amo_sim *s, double *out);
/* learning_step! for P = 1: gd = averaged (j, dj, dlogq_fwd, g); returns new parameter */
double amo_learning_step(int opt, double hyper0, double hyper1, double theta,
                         const double gd[4]);

/* build_schedule: returns number of entries written (<= cap) */
int64_t amo_build_schedule_linear(int64_t steps, int64_t burn, int64_t dt,
                                  int64_t *out, int64_t cap);
int64_t amo_build_schedule_block(int64_t steps, int64_t burn, const int64_t *block,
                                 int n_block, int64_t *out, int64_t cap);
int64_t amo_build_schedule_log(int64_t steps, int64_t burn, double base,
                               int64_t *out, int64_t cap);

/* pooled-position statistic of test/distribution_test.jl (n, sum x, sum x^2, mean energy) */
void   amo_run_pooled_moments(amo_sim *s, int64_t steps, int64_t burn, int64_t dt, int n_threads,
                              double out[4]);

void   amo_set_counters(amo_sim *s, const int64_t *accepted, const int64_t *total);
uint64_t amo_get_estimator_step(const amo_sim *s);
void   amo_set_estimator_step(amo_sim *s, uint64_t t);

int    amo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
