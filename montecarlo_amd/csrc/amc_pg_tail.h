// amc_pg_tail.h -- what surrounds the estimator launch: its argument and tail records, the GradientData sample of every policy kind, the merge of
// shards' records, accumulate / update kernels, policies with several parameters, and the learning step a launch leaves pending.
// Part of the kernel sources of the many-chain Metropolis engine (gfx950 / CDNA4); amc_kernels.h includes all of them, in order.
#pragma once

#include "amc_reduce_pass.h"

namespace amc {

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
    // the launch's first learnable move in the estimator CALL -- the index that, with q, names a sample's draw.  AMC_NP > 1,
    // AMC_NCLASS > 1: a launch takes ONE learnable move (several parameters: its columns fill a row; classes: hipcc 7.2 fails on the
    // unrolled loop over moves with a class switch in it for some pools); built-in forms: a call over more than four learnable moves
    // whose samples need the flushing form goes in launches of four (amc_pg.hip pg_accumulate_impl)
    int32_t l_base;
};
// Rewrites the record in stream order: the value travels as a kernel argument (copied at launch), so no host buffer has
// to outlive the call and launches already queued keep reading the old record until they are done.
#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ void pg_tail_store_kernel(PgTail value, PgTail* dst)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = value;
}
#endif

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
    const double logq_f = user_logq_dlogq(delta, x, sigma, T, k, th, d_f);      // withgrad_log_proposal_density!, gradients.jl:97
    const real_t e1 = potential<POT>(x, T);
    const real_t xn = user_perform(x, delta, T, k);
    const real_t e2 = potential<POT>(xn, T);
    const real_t dlogp = ((-e2) * beta) - ((-e1) * beta);
    const double r = (POT == POT_CUSTOM) ? user_reward(delta, xn, T) : (double)(delta * delta);
    const real_t nd = user_invert(delta, xn, T, k);
    double logq_b = logq_f;
#if AMC_NCLASS > 1 && defined(AMC_CLASS_GAUSS_EST_MASK)
    // the launch's move belongs to a class that IS the built-in Gaussian displacement, derivative included (amc_rtc.hip): the backward
    // density and its derivative at -delta are the forward ones bit for bit ((-d)*(-d) == d*d) -- not formed again (wave-uniform branch)
    if ((((unsigned)AMC_CLASS_GAUSS_EST_MASK >> (k >> 8)) & 1u) != 0u) {
#pragma unroll
        for (int p = 0; p < AMC_NP; ++p) d_b[p] = d_f[p];
    } else
#endif
    {
        logq_b = user_logq_dlogq(nd, xn, sigma, T, k, th, d_b);                 // gradients.jl:102
    }
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

#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ void pg_accumulate_kernel(const double* recs, int n_ranks, int n_learn, PgIds ids, double n_samples, double* acc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double vals[AMC_MAX_LEARN * 4];
    pg_merge_slots(recs, n_ranks, n_learn * 4, vals);
    for (int l = 0; l < n_learn; ++l) pg_accumulate_one(vals, l, ids.v[l], n_samples, acc);
}
#endif

// Both in one launch, for shards connected by a communicator: what follows the in-place all-reduce of the estimator's sums
// when the time step also updates (estimator.jl:130, then update.jl:50-57) -- one tiny launch on the critical path instead of two.
// theta_used: see pg_update_all (the ring slot of the launch before, or nullptr)
#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ void pg_accumulate_update_kernel(const double* recs, int n_ranks, double* ptab, double* acc, int n_learn, PgIds ids,
                                                              double n_samples, PgOpts opt, int n_moves, int* status, const double* theta_used)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double vals[AMC_MAX_LEARN * 4];
    pg_merge_slots(recs, n_ranks, n_learn * 4, vals);
    pg_update_all(ptab, acc, n_learn, ids.v, opt, n_moves, status, vals, n_samples, theta_used);
}
#endif

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
#if AMC_PLAIN_KERNELS
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
#endif

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
#if AMC_PLAIN_KERNELS
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
#endif

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
// bits 24 .. 31: per learnable move l < 4, the GradientData column groups its optimiser does not read (bit 0 grad logq_forward, bit 1
// the metric g) -- set only where the launch's own tail takes the learning step, so that nothing of the skipped sums outlives it
__host__ __device__ inline int pg_skip_of(int tail_mode, int l) { return l < 4 ? (int)(((unsigned)tail_mode >> (24 + 2 * l)) & 3u) : 0; }
__host__ inline int pg_skip_bits(const int* kinds, int n_learn)
{
    unsigned bits = 0u;
    for (int l = 0; l < n_learn && l < 4; ++l) {
        const int k = kinds[l];
        const bool reads_dlogq = k == 2 || k == 3 || k == 6;         // BLPG, BLAPG, BLANPG: the baseline term j grad logq (learning.jl:50-52,77-79,160-164)
        const bool reads_g = k == 4 || k == 5 || k == 6;             // NPG, ANPG, BLANPG: the metric (learning.jl:103-105,130-134)
        bits |= (unsigned)((reads_dlogq ? 0 : 1) | (reads_g ? 0 : 2)) << (2 * l);
    }
    return (int)(bits << 24);
}

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
#if AMC_PLAIN_KERNELS
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
#endif
}  // namespace amc
