// amc_params.h -- the small kernels around the parameter table: the synthetic initial ensemble, per-move derived parameters and the pick
// table, and the device-resident learning step (learning.jl) of one-parameter policies.
// Part of the kernel sources of the many-chain Metropolis engine (gfx950 / CDNA4); amc_kernels.h includes all of them, in order.
#pragma once

#include "amc_sweep.h"

namespace amc {

// K0: synthetic initial ensemble, x_c = lo + (hi-lo)*u (MC_harmonic_oscillator.jl:13).
#if AMC_PLAIN_KERNELS
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
#endif

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

#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ void prepare_params_kernel(double* ptab, int n_moves)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    prepare_params(ptab, n_moves);
}
#endif

// The move-pick table (see AMC_PICK_CELLS): cell c covers the pick uniforms r in [c, c+1) 2^-12 (both ends exact).
// The walk's count #(cum[i] <= r), i < K-1, is monotone in r, so it is the same for every r of the cell iff it is the
// same at the two ends: #(cum[i] <= c 2^-12) == #(cum[i] < (c+1) 2^-12).  Launched after prepare_params (same stream)
// whenever the weights change.
#if AMC_PLAIN_KERNELS
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
#endif

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

#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ void pg_update_kernel(double* ptab, double* acc, int n_learn, PgIds ids, PgOpts opt, int n_moves, int* status)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    pg_update_all(ptab, acc, n_learn, ids.v, opt, n_moves, status);
}
#endif
}  // namespace amc
