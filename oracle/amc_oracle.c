/*
 * amc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See amc_oracle.h for scope, parity status and who may load this.
 *
 * Reference-shaped data model on purpose (array of Particle structs, one pool
 * of Move records per chain, parameters shared by all chains), arithmetic in
 * the reference's operation order.  Citations are relative to /root/reference.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -mfma -fopenmp).
 * -ffp-contract=off matters: every fused multiply-add below is an explicit
 * fma() and nothing else may be contracted, or host and device diverge.
 */
#include "amc_oracle.h"
#include "amc_tables.inc"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* Third-party piece 1: Philox4x32-10 (Random123; Salmon et al., SC'11).     */
/* The reference's RNG is Julia stdlib Xoshiro via the R= hook               */
/* (src/metropolis.jl:245,263); replaced by this counter-based generator,    */
/* bit-compatible with rocRAND's philox4x32_10 engine                        */
/* (/opt/rocm/include/rocrand/rocrand_philox4x32_10.h:270-303).              */
/* ------------------------------------------------------------------------ */
static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

void amo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)m1;
        uint32_t n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Draw schedule: the 128-bit counter names ONE draw of ONE
 * chain pair: x = step[31:0]; y = step[47:32] | draw<<16 | stream<<28;
 * (z,w) = global pair id (= rocRAND "subsequence").  Replaces the per-chain
 * sequential generators rngs[c] = R(seed + c - 1), metropolis.jl:262-263. */
void amo_counter(uint64_t pair, uint64_t t, uint32_t draw, uint32_t stream, uint32_t ctr[4])
{
    ctr[0] = (uint32_t)t;
    ctr[1] = ((uint32_t)(t >> 32) & 0xFFFFu) | ((draw & 0xFFFu) << 16) | ((stream & 0xFu) << 28);
    ctr[2] = (uint32_t)pair;
    ctr[3] = (uint32_t)(pair >> 32);
}

static void draw4(const amo_sim *s, uint64_t pair, uint64_t t, uint32_t draw,
                  uint32_t stream, uint32_t out[4]);

/* rand(rng)::Float64 in [0,1) (metropolis.jl:184).  Julia's own construction (Random stdlib,
 * rand(r, CloseOpen12()) - 1): 52 random bits become the significand of a double in [1,2), then
 * subtract 1.  The 64-bit word is (hi:lo), its top 52 bits are used. */
static inline double bits_to_12(uint32_t lo, uint32_t hi, uint64_t expo)
{
    uint64_t m = (((uint64_t)hi << 32) | lo) >> 12;
    return u2d(expo | m);
}

double amo_uniform_co(uint32_t lo, uint32_t hi)      /* [0,1): d - 1, d in [1,2) */
{
    return bits_to_12(lo, hi, 0x3ff0000000000000ull) - 1.0;
}

double amo_uniform_oc(uint32_t lo, uint32_t hi)      /* (0,1]: 2 - d (Box-Muller radius argument) */
{
    return 2.0 - bits_to_12(lo, hi, 0x3ff0000000000000ull);
}

/* Arithmetic spec v5.  How the 128 bits (x, y, z, w) of a step's NORMAL draw are used:
 *   radius  u in (0,1]   the top 52 bits of (y:x)                          (unchanged since v2)
 *   angle   w in (0,2]   the top 28 bits of word w: 2 - (w >> 4) 2^-27.  2^28 equally spaced angles: the two normals
 *                        s sin(pi w), s cos(pi w) keep a continuous radius, and an average over equally spaced angles is
 *                        the trapezoid rule on a periodic integrand -- exact for every Fourier mode below 2^28 --, so the
 *                        marginals are Gaussian to far below anything 10^20 samples could resolve
 *   spare   48 bits      x[11:0], z[31:0], w[3:0]: per chain of the pair 12 bits that lead its ACCEPT uniform and 12
 *                        bits that lead its move-PICK uniform
 *                          even chain: accept12 = x & 0xFFF          pick12 = z & 0xFFF
 *                          odd  chain: accept12 = (z >> 12) & 0xFFF  pick12 = (z >> 24) | ((w & 0xF) << 8)
 * and the chain's 64-bit word W of the step's ACCEPT draw (even: (y:x), odd: (w:z)):
 *   accept uniform (rand(rng), metropolis.jl:184): 52-bit significand = accept12, then the top 40 bits of W -- Julia's
 *                        rand(Float64) construction (d in [1,2) minus 1) on it
 *   pick uniform (rand(rng, Categorical(weights)), :206): (pick12 2^24 + (W & 0xFFFFFF)) 2^-36
 * Both uniforms are thus known to 2^-12 from the normal draw alone; the kernels form the accept draw only for the waves
 * in which that bracket leaves an accept decision or a move pick open. */
double amo_angle28(uint32_t w)                       /* (0,2]: Box-Muller angle / pi */
{
    return 2.0 - (double)(w >> 4) * 0x1.0p-27;
}

uint32_t amo_spare_accept12(const uint32_t v[4], int half)
{
    return half ? ((v[2] >> 12) & 0xFFFu) : (v[0] & 0xFFFu);
}

uint32_t amo_spare_pick12(const uint32_t v[4], int half)
{
    return half ? ((v[2] >> 24) | ((v[3] & 0xFu) << 8)) : (v[2] & 0xFFFu);
}

double amo_uniform_accept(uint32_t accept12, uint32_t lo, uint32_t hi)
{
    uint64_t m = ((uint64_t)(accept12 & 0xFFFu) << 40) | ((((uint64_t)hi << 32) | lo) >> 24);
    uint64_t b = 0x3ff0000000000000ull | m;
    double d;
    memcpy(&d, &b, sizeof d);
    return d - 1.0;
}

double amo_uniform_pick(uint32_t pick12, uint32_t lo)
{
    return (double)(((uint64_t)(pick12 & 0xFFFu) << 24) | (uint64_t)(lo & 0xFFFFFFu)) * 0x1.0p-36;
}

/* ------------------------------------------------------------------------ */
/* exp / log / sincospi: own implementations so host and device agree bit    */
/* for bit (libm and the GPU's ocml differ in the last ulp).  Constants from  */
/* tools/gen_math_constants.py.  Accuracy ~1 ulp; NOT correctly rounded, so   */
/* vs Julia's exp/log an accept decision can flip with probability ~1e-16.   */
/* ------------------------------------------------------------------------ */

double amo_exp(double x)
{
    /* exp(x) = 2^m * 2^(j/32) * e^r, k = 32 m + j = rint(x * 32/ln2), |r| <= ln2/64: table of 32
     * correctly rounded 2^(j/32) and a degree-6 Taylor polynomial (error < 4e-18). */
    if (x != x) return x;
    if (x > 709.0) return INFINITY;       /* documented domain cut (true overflow at 709.78) */
    if (x < -708.0) return 0.0;           /* flush: no subnormal results */
    const double INV_L = 0x1.71547652b82fep+5;      /* 32/ln2 */
    const double L_HI = 0x1.62e42fee00000p-6;       /* ln2/32 split hi/lo */
    const double L_LO = 0x1.a39ef35793c76p-38;
    const double SHIFT = 0x1.8p52;
    double t = fma(x, INV_L, SHIFT);      /* round-to-nearest-even integer in the low bits */
    double kd = t - SHIFT;
    int32_t ki = (int32_t)(uint32_t)d2u(t);
    double r = fma(-kd, L_HI, x);
    r = fma(-kd, L_LO, r);
    double p = 0x1.6c16c16c16c17p-10;     /* 1/720 */
    p = fma(p, r, 0x1.1111111111111p-7);
    p = fma(p, r, 0x1.5555555555555p-5);
    p = fma(p, r, 0x1.5555555555555p-3);
    p = fma(p, r, 0x1.0000000000000p-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    double y = AMC_TAB_EXP2[ki & 31] * p;
    int64_t m = (int64_t)(ki >> 5);       /* arithmetic shift: floor(k / 32) */
    return u2d(d2u(y) + ((uint64_t)m << 52));   /* y * 2^m, y in [0.98, 2.02), result normal */
}

/* log(x): argument reduction x = 2^k * m, m in [sqrt(2)/2, sqrt(2)), then the
 * classic s = f/(2+f) series with the 7-term minimax polynomial published in
 * FreeBSD/fdlibm e_log.c (Lg1..Lg7), always taking its "hfsq" form. */
double amo_log(double x)
{
    if (x != x || x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    const double LN2_HI = 0x1.62e42fee00000p-1;
    const double LN2_LO = 0x1.a39ef35793c76p-33;
    int64_t k = 0;
    uint64_t ux = d2u(x);
    if (ux < 0x0010000000000000ull) {     /* subnormal: scale by 2^54 */
        x = x * 0x1.0p54;
        ux = d2u(x);
        k = -54;
    }
    uint32_t hx = (uint32_t)(ux >> 32);
    k += (int64_t)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u;     /* 1 if m >= sqrt(2): halve it */
    k += (int64_t)(i >> 20);
    uint64_t um = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffull);
    double f = u2d(um) - 1.0;
    double s = f / (2.0 + f);
    double dk = (double)k;
    double z = s * s;
    double w = z * z;
    double t1 = w * fma(w, fma(w, 0x1.39a09d078c69fp-3, 0x1.c71c51d8e78afp-3), 0x1.999999997fa04p-2);
    double t2 = z * fma(w, fma(w, fma(w, 0x1.2f112df3e5244p-3, 0x1.7466496cb03dep-3),
                                 0x1.2492494229359p-2), 0x1.5555555555593p-1);
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    return dk * LN2_HI - ((hfsq - fma(s, hfsq + R, dk * LN2_LO)) - f);
}

/* log(u) for the Box-Muller radius, u positive normal (here u in [2^-52, 1]): division-free.
 * u = 2^k * m with m centred into [sqrt(2)/2, sqrt(2)) as in amo_log; table index = (low exponent
 * bit : top 7 significand bits) of m; r = m/c - 1 by one fma with INVC = RN(1/c), then
 * log m = LOGC + log1p(r), log1p by its degree-7 Taylor polynomial (|r| < 2^-7: error < 2e-18).
 * The two intervals touching 1 use c = 1, so log(1) = 0 exactly and -2 log u is never negative. */
double amo_logbm(double u)
{
    const double LN2_HI = 0x1.62e42fee00000p-1;
    const double LN2_LO = 0x1.a39ef35793c76p-33;
    uint64_t ux = d2u(u);
    uint32_t hx = (uint32_t)(ux >> 32);
    int32_t k = (int32_t)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u;       /* 1 if m >= sqrt(2): halve it */
    k += (int32_t)(i >> 20);
    hx |= i ^ 0x3ff00000u;                          /* exponent field 0x3ff (m >= 1) or 0x3fe (m < 1) */
    double m = u2d(((uint64_t)hx << 32) | (ux & 0xffffffffull));
    uint32_t idx = ((hx >> 13) & 0xffu) - AMC_TAB_LOG_IDX_MIN;    /* only 53..181 occur */
    double r = fma(m, AMC_TAB_LOG_INVC[idx], -1.0);
    double p = 0x1.2492492492492p-3;                /* 1/7 */
    p = fma(p, r, -0x1.5555555555555p-3);           /* -1/6 */
    p = fma(p, r, 0x1.999999999999ap-3);            /*  1/5 */
    p = fma(p, r, -0x1.0000000000000p-2);           /* -1/4 */
    p = fma(p, r, 0x1.5555555555555p-2);            /*  1/3 */
    p = fma(p, r, -0x1.0000000000000p-1);           /* -1/2 */
    p = fma(p, r, 1.0);
    double dk = (double)k;
    double hi = fma(dk, LN2_HI, AMC_TAB_LOG_LOGC[idx]);
    return fma(p, r, fma(dk, LN2_LO, hi));
}

/* sincospi(w) for the Box-Muller angle, |w| < 2^24: n = rint(64 w), r = w - n/64 (exact, |r| <= 1/128),
 * sin(pi r), cos(pi r) by short Taylor polynomials, rotated by the table angle pi*(n mod 128)/64. */
void amo_sincospi(double w, double *sp, double *cp)
{
    const double SHIFT = 0x1.8p52;
    double t = fma(w, 64.0, SHIFT);
    double nd = t - SHIFT;
    uint32_t j = (uint32_t)d2u(t) & 127u;
    double r = fma(nd, -0x1.0p-6, w);
    double z = r * r;
    double ps = -0x1.32d2cce62bd86p-1;              /* -pi^7/7! */
    ps = fma(ps, z, 0x1.466bc6775aae2p+1);          /*  pi^5/5! */
    ps = fma(ps, z, -0x1.4abbce625be53p+2);         /* -pi^3/3! */
    ps = fma(ps, z, 0x1.921fb54442d18p+1);          /*  pi      */
    double sr = ps * r;
    double pc = -0x1.55d3c7e3cbffap+0;              /* -pi^6/6! */
    pc = fma(pc, z, 0x1.03c1f081b5ac4p+2);          /*  pi^4/4! */
    pc = fma(pc, z, -0x1.3bd3cc9be45dep+2);         /* -pi^2/2! */
    double cr = fma(pc, z, 1.0);
    double S = AMC_TAB_SINPI[j], C = AMC_TAB_COSPI[j];
    *sp = fma(S, cr, C * sr);
    *cp = fma(C, cr, -(S * sr));
}

/* Third-party piece 2: randn.  The reference draws rand(rng, Normal(0, sigma))
 * (particle_1d.jl:57), i.e. Distributions.jl 0.25 `mu + sigma * randn(rng)`.
 * Julia's ziggurat is replaced by a Box-Muller pair in the shape of rocRAND's
 * box_muller_double(uint4) (rocrand_normal.h:78-98): u in (0,1] from words (x,y), w in (0,2]
 * from word w, z = sqrt(-2 log u) * (sinpi w, cospi w), with a 52-bit radius uniform, a 28-bit angle
 * (amo_angle28) and the table-driven log / sincospi above.  One call serves a chain PAIR. */
void amo_box_muller(const uint32_t v[4], double z[2])
{
    double u = amo_uniform_oc(v[0], v[1]);          /* (0,1] */
    double w = amo_angle28(v[3]);                   /* (0,2] */
    double s = sqrt(-2.0 * amo_logbm(u));
    double sn, cs;
    amo_sincospi(w, &sn, &cs);
    z[0] = sn * s;
    z[1] = cs * s;
}

/* ------------------------------------------------------------------------ */
/* L0 model: example/particle_1d/particle_1d.jl                              */
/* ------------------------------------------------------------------------ */

/* potential(x): free function defined by the driver script,
 * harmonic_oscillator/MC_harmonic_oscillator.jl:4  potential(x) = x^2 (== x*x).
 * The double well (x^2-1)^2 is BASELINE config 3's, not in the reference. */
static double (*g_custom_potential)(double) = 0;
static double (*g_custom_reward)(double, double) = 0;      /* reward(action, system) as f(delta, x_new); 0: delta^2 */

void amo_set_custom_reward(double (*fn)(double, double)) { g_custom_reward = fn; }

void amo_set_custom_potential(double (*fn)(double)) { g_custom_potential = fn; }

/* A script-defined policy of the Gaussian-displacement family: sample_action! / log_proposal_density receive `system`
 * (particle_1d.jl:52-59), so the proposal width may depend on the state: sigma * scale(x).  0: scale == 1, the
 * reference's StandardGaussian. */
static double (*g_custom_scale)(double) = 0;
static float (*g_custom_scale_f32)(float) = 0;
void amo_set_custom_scale(double (*fn)(double)) { g_custom_scale = fn; }
void amo_set_custom_scale_f32(float (*fn)(float)) { g_custom_scale_f32 = fn; }

/* A script-defined proposal in full: the model's own sample_action! / log_proposal_density (generic functions of
 * src/metropolis.jl:35-62) and, for the estimator, d logq / d sigma (what ForwardDiff returns, gradients.jl:28-33), as
 * functions compiled from the script's expressions: sample(z, x, sigma), logq(delta, x, sigma), dlogq(delta, x, sigma).
 * 0: the particle_1d model's Gaussian displacement. */
static double (*g_custom_sample)(double, double, double) = 0;
static double (*g_custom_logq)(double, double, double) = 0;
static double (*g_custom_dlogq)(double, double, double) = 0;
void amo_set_custom_proposal(double (*sample)(double, double, double), double (*logq)(double, double, double),
                             double (*dlogq)(double, double, double))
{
    g_custom_sample = sample; g_custom_logq = logq; g_custom_dlogq = dlogq;
}

/* A script-defined policy with SEVERAL parameters (Move.parameters is an array, src/metropolis.jl:140-147; GradientData keeps
 * grad j / grad logq_forward as arrays of its shape and g as their outer product, PolicyGuided/gradients.jl:41-61,104-108):
 * sample(z, x, theta), logq(delta, x, theta), dlogq(delta, x, theta, out[P]) compiled from the script's expressions.
 * 0: none (the one-parameter forms above). */
#define AMO_MAX_NP 4
static int g_np = 1;
static double (*g_vec_sample)(double, double, const double *) = 0;
static double (*g_vec_logq)(double, double, const double *) = 0;
static void (*g_vec_dlogq)(double, double, const double *, double *) = 0;
void amo_set_vector_policy(int np, double (*sample)(double, double, const double *), double (*logq)(double, double, const double *),
                           void (*dlogq)(double, double, const double *, double *))
{
    g_np = sample ? np : 1; g_vec_sample = sample; g_vec_logq = logq; g_vec_dlogq = dlogq;
}

/* A script-defined ACTION (the reference's Action interface, src/metropolis.jl:15-119; particle_1d.jl:30-40 are the
 * displacement's methods): perform(x, delta) = the position after perform_action!, invert(delta, x_new) = the parameter of
 * the inverted action.  0: the displacement (x + delta, -delta).  Used together with a script-defined proposal. */
static double (*g_custom_perform)(double, double) = 0;
static double (*g_custom_invert)(double, double) = 0;

/* Pools that mix policy / action types (every Move carries its own action and policy, src/metropolis.jl:140-162; the generic
 * functions dispatch on their types): up to AMO_MAX_CLASSES sets of functions, one class per move.  The class of the move at hand
 * is a thread-local the sweep / estimator loops set before they call mc_step / pgmc_sample (the chains run on OpenMP threads). */
#define AMO_MAX_CLASSES 4
static int g_n_classes = 1;
static int g_class_of_move[64];
static double (*g_cls_sample[AMO_MAX_CLASSES])(double, double, double);
static double (*g_cls_logq[AMO_MAX_CLASSES])(double, double, double);
static double (*g_cls_dlogq[AMO_MAX_CLASSES])(double, double, double);
static double (*g_cls_perform[AMO_MAX_CLASSES])(double, double);
static double (*g_cls_invert[AMO_MAX_CLASSES])(double, double);
static _Thread_local int tl_cls = 0;
#define CUR_SAMPLE  (g_n_classes > 1 ? g_cls_sample[tl_cls] : g_custom_sample)
#define CUR_LOGQ    (g_n_classes > 1 ? g_cls_logq[tl_cls] : g_custom_logq)
#define CUR_DLOGQ   (g_n_classes > 1 ? g_cls_dlogq[tl_cls] : g_custom_dlogq)
#define CUR_PERFORM (g_n_classes > 1 ? g_cls_perform[tl_cls] : g_custom_perform)
#define CUR_INVERT  (g_n_classes > 1 ? g_cls_invert[tl_cls] : g_custom_invert)
static inline void set_class_of(int k) { tl_cls = g_n_classes > 1 ? g_class_of_move[k] : 0; }
void amo_set_policy_classes(int n_classes, const int *class_of_move, int n_moves, void *const *sample, void *const *logq,
                            void *const *dlogq, void *const *perform, void *const *invert)
{
    g_n_classes = n_classes > 1 ? n_classes : 1;
    if (g_n_classes == 1) return;
    for (int k = 0; k < 64; ++k) g_class_of_move[k] = k < n_moves ? class_of_move[k] : 0;
    for (int c = 0; c < n_classes; ++c) {
        g_cls_sample[c] = (double (*)(double, double, double))sample[c];
        g_cls_logq[c] = (double (*)(double, double, double))logq[c];
        g_cls_dlogq[c] = dlogq ? (double (*)(double, double, double))dlogq[c] : 0;
        g_cls_perform[c] = perform ? (double (*)(double, double))perform[c] : 0;
        g_cls_invert[c] = invert ? (double (*)(double, double))invert[c] : 0;
    }
    /* the one-policy switches stay meaningful: a script-defined proposal is active, with or without its sigma-derivative */
    g_custom_sample = g_cls_sample[0]; g_custom_logq = g_cls_logq[0]; g_custom_dlogq = g_cls_dlogq[0];
}
void amo_set_custom_action(double (*perform)(double, double), double (*invert)(double, double))
{
    g_custom_perform = perform; g_custom_invert = invert;
}

double amo_potential(int pot, double x)
{
    if (pot == AMO_POT_CUSTOM) return g_custom_potential ? g_custom_potential(x) : (0.0 / 0.0);
    if (pot == AMO_POT_DOUBLE_WELL) {
        double q = x * x - 1.0;
        return q * q;
    }
    return x * x;
}

/* particle_1d.jl:9-16: mutable struct Particle (x, beta, e), e = potential(x). */
typedef struct { double x, beta, e; } particle_t;

/* particle_1d.jl:26-28 Displacement(delta) + metropolis.jl:140-147 Move's
 * per-chain mutable part.  policy/parameters/weight are aliased across chains
 * (metropolis.jl:252-260) and live once in amo_sim. */
typedef struct { double delta; int64_t total_calls, accepted_calls; } move_t;

struct amo_sim {
    int64_t M, offset;
    int pot, K, sweepstep;
    int f32;                  /* Particle{Float32} / Displacement{Float32}: see the Float32 section below */
    double *sigma, *weight;   /* shared parameters / weights, length K */
    double *theta_more;       /* parameters 1 .. 3 of every move (a policy with several): [K][AMO_MAX_NP - 1] */
    particle_t *chains;       /* Vector{Particle} */
    move_t *pools;            /* pools[c*K + k] */
    uint64_t seed;
    uint64_t t;               /* MH steps done so far per chain */
    uint64_t t_est;           /* estimator make_step! calls so far */
};

/* particle_1d.jl:20-22: unnormalised_log_target_density(state) = -state[1]*state[2] */
static inline double unnormalised_log_target_density(double e, double beta)
{
    return (-e) * beta;
}

/* metropolis.jl:74 (generic fallback): logp(x2) - logp(x1) */
static inline double delta_log_target_density(double e1, double b1, double e2, double b2)
{
    return unnormalised_log_target_density(e2, b2) - unnormalised_log_target_density(e1, b1);
}

/* particle_1d.jl:52-54:  -(d)^2 / (2s^2) - log(2pi * s^2) / 2   (2pi = 2*Float64(pi)) */
double amo_log_proposal_density(double delta, double sigma)
{
    const double TWO_PI = 0x1.921fb54442d18p+2;
    double s2 = sigma * sigma;
    return (-(delta * delta)) / (2.0 * s2) - amo_log(TWO_PI * s2) / 2.0;
}

/* gradients.jl:28-33 withgrad_log_proposal_density!: d/dsigma of the above.
 * ForwardDiff 0.10's dual-number rules restated by hand (sigma^2 -> (s2, s+s);
 * c/Dual -> -(v/den)*dden; log -> da/a).  The reference pins only the VALUE
 * (-5.0 at delta=0, sigma=0.2, atol 1e-10: test/ad_backends_test.jl:31-32). */
double amo_grad_log_proposal_density(double delta, double sigma)
{
    const double TWO_PI = 0x1.921fb54442d18p+2;
    double s2 = sigma * sigma, ds2 = sigma + sigma;
    double den = 2.0 * s2, dden = 2.0 * ds2;
    double q1 = (-(delta * delta)) / den;
    double dq1 = -(q1 / den) * dden;
    double a = TWO_PI * s2, da = TWO_PI * ds2;
    double dl = da / a;
    return dq1 - dl / 2.0;
}

/* The same gradient for a width w = sigma * scale(x) (dw = scale(x) = dw/dsigma): ForwardDiff's rules in the
 * function's order, w*w -> (w2, dw*w + w*dw).  d2neg = -(delta^2), already formed in the action's type. */
static double grad_log_proposal_density_w(double d2neg, double w, double dw)
{
    const double TWO_PI = 0x1.921fb54442d18p+2;
    double w2 = w * w, dw2 = dw * w + w * dw;
    double den = 2.0 * w2, dden = 2.0 * dw2;
    double q1 = d2neg / den;
    double dq1 = -(q1 / den) * dden;
    double a = TWO_PI * w2, da = TWO_PI * dw2;
    double dl = da / a;
    return dq1 - dl / 2.0;
}

/* particle_1d.jl:30-35 perform_action!: e1 = e; x += delta; e = potential(x). */
static inline void perform_action(particle_t *p, const move_t *m, int pot, double *e1, double *e2)
{
    *e1 = p->e;
    p->x += m->delta;
    p->e = amo_potential(pot, p->x);
    *e2 = p->e;
}

/* perform_action! of a script-defined action (amo_set_custom_action); the displacement when none is set. */
static inline void script_perform_action(particle_t *p, const move_t *m, int pot, double *e1, double *e2)
{
    *e1 = p->e;
    p->x = CUR_PERFORM ? CUR_PERFORM(p->x, m->delta) : p->x + m->delta;
    p->e = amo_potential(pot, p->x);
    *e2 = p->e;
}

/* Julia's min(a, b): NaN if either is NaN (C fmin would return the non-NaN). */
static inline double julia_min(double a, double b)
{
    if (a != a) return a;
    if (b != b) return b;
    return b < a ? b : a;
}

/* the move's parameter array: theta[0] is what the one-parameter forms call sigma */
static inline void move_theta(const struct amo_sim *s, int k, double th[AMO_MAX_NP])
{
    th[0] = s->sigma[k];
    for (int p = 1; p < AMO_MAX_NP; ++p) th[p] = s->theta_more[k * (AMO_MAX_NP - 1) + p - 1];
}

static inline void script_perform_action(particle_t *p, const move_t *m, int pot, double *e1, double *e2);

/* metropolis.jl:176-190 mc_step! for a policy with several parameters */
static inline int mc_step_vec(particle_t *p, move_t *m, const double *theta, int pot, double z, double u)
{
    m->delta = g_vec_sample(z, p->x, theta);                           /* :177 sample_action! */
    double logq_f = g_vec_logq(m->delta, p->x, theta);                 /* :178 at the old state */
    double e1c, e2c;
    script_perform_action(p, m, pot, &e1c, &e2c);                      /* :179 */
    double dlogp_c = delta_log_target_density(e1c, p->beta, e2c, p->beta); /* :180 */
    m->delta = CUR_INVERT ? CUR_INVERT(m->delta, p->x) : -m->delta;  /* :181 */
    double logq_b = g_vec_logq(m->delta, p->x, theta);                 /* :182 at the new state */
    double alpha_c = julia_min(1.0, amo_exp(dlogp_c + logq_b - logq_f)); /* :183 */
    if (alpha_c > u) return 1;                                         /* :184 */
    script_perform_action(p, m, pot, &e1c, &e2c);                      /* :187 perform_action_cached! */
    return 0;
}

/* metropolis.jl:176-190 mc_step!, with sample_action! (particle_1d.jl:56-59)
 * fed an explicit standard normal z and the accept uniform u. */
static inline int mc_step(particle_t *p, move_t *m, double sigma, int pot, double z, double u)
{
    if (g_custom_logq) {                                           /* script-defined proposal */
        m->delta = CUR_SAMPLE(z, p->x, sigma);                    /* :177 sample_action! */
        double logq_f = CUR_LOGQ(m->delta, p->x, sigma);          /* :178 at the old state */
        double e1c, e2c;
        script_perform_action(p, m, pot, &e1c, &e2c);                  /* :179 */
        double dlogp_c = delta_log_target_density(e1c, p->beta, e2c, p->beta); /* :180 */
        m->delta = CUR_INVERT ? CUR_INVERT(m->delta, p->x) : -m->delta;  /* :181 invert_action!(action, system) */
        double logq_b = CUR_LOGQ(m->delta, p->x, sigma);          /* :182 at the new state */
        double alpha_c = julia_min(1.0, amo_exp(dlogp_c + logq_b - logq_f)); /* :183 */
        if (alpha_c > u) return 1;                                     /* :184 */
        script_perform_action(p, m, pot, &e1c, &e2c);                  /* :187 perform_action_cached! */
        return 0;
    }
    /* the policy's width at the state it is asked about: the old one for sample_action! and the forward density, the
     * new one for the backward density (system has moved by then) */
    double s_f = g_custom_scale ? sigma * g_custom_scale(p->x) : sigma;
    m->delta = 0.0 + s_f * z;                                      /* :177 -> particle_1d.jl:57 */
    double logq_forward = amo_log_proposal_density(m->delta, s_f); /* :178 */
    double e1, e2;
    perform_action(p, m, pot, &e1, &e2);                           /* :179 */
    double dlogp = delta_log_target_density(e1, p->beta, e2, p->beta); /* :180 */
    m->delta = -m->delta;                                          /* :181 invert_action! */
    double s_b = g_custom_scale ? sigma * g_custom_scale(p->x) : sigma;
    double logq_backward = amo_log_proposal_density(m->delta, s_b);    /* :182 */
    double alpha = julia_min(1.0, amo_exp(dlogp + logq_backward - logq_forward)); /* :183 */
    if (alpha > u)                                                 /* :184 */
        return 1;
    perform_action(p, m, pot, &e1, &e2);   /* :187 perform_action_cached! -> :90 perform_action! */
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Float32 state.  Particle{T} and Displacement{T} are generic in T <: AbstractFloat
 * (particle_1d.jl:9,26); with T = Float32 Julia's promotion rules give:
 *   x, beta, e, delta              Float32 (struct fields of type T; assignments convert)
 *   sample_action!  :56-59         rand(rng, Normal(zero(T), sigma::Float64)) is Float64 -> delta = Float32(.)
 *   log_proposal_density :52-54    (delta)^2 and its negation in Float32, then / (2 sigma^2) promotes: Float64
 *   perform_action! :30-35         x += delta, e = potential(x) in Float32
 *   delta_log_target_density       (-e2*beta) - (-e1*beta) in Float32
 *   alpha, rand(rng), reward*alpha Float64 (Float32 + Float64 promotes)
 * The policy parameters stay Float64 (setup_parameters :50, ComponentArray(sigma = ...)).  The chain fields below
 * keep doubles that always hold Float32 values; every Float32 operation is written with C floats. */
static float (*g_custom_potential_f32)(float) = 0;
static double (*g_custom_reward_f32)(float, float) = 0;

void amo_set_custom_potential_f32(float (*fn)(float)) { g_custom_potential_f32 = fn; }
void amo_set_custom_reward_f32(double (*fn)(float, float)) { g_custom_reward_f32 = fn; }

float amo_potential_f32(int pot, float x)
{
    if (pot == AMO_POT_CUSTOM) return g_custom_potential_f32 ? g_custom_potential_f32(x) : (0.0f / 0.0f);
    if (pot == AMO_POT_DOUBLE_WELL) {
        float q = x * x - 1.0f;
        return q * q;
    }
    return x * x;
}

static double log_proposal_density_f32(float delta, double sigma)
{
    const double TWO_PI = 0x1.921fb54442d18p+2;
    double s2 = sigma * sigma;
    float d2 = -(delta * delta);                                   /* Float32 */
    return (double)d2 / (2.0 * s2) - amo_log(TWO_PI * s2) / 2.0;
}

static double grad_log_proposal_density_f32(float delta, double sigma)
{
    const double TWO_PI = 0x1.921fb54442d18p+2;
    double s2 = sigma * sigma, ds2 = sigma + sigma;
    double den = 2.0 * s2, dden = 2.0 * ds2;
    float d2 = -(delta * delta);
    double q1 = (double)d2 / den;
    double dq1 = -(q1 / den) * dden;
    double a = TWO_PI * s2, da = TWO_PI * ds2;
    double dl = da / a;
    return dq1 - dl / 2.0;
}

static inline void perform_action_f32(particle_t *p, float delta, int pot, float *e1, float *e2)
{
    *e1 = (float)p->e;
    float x = (float)p->x + delta;
    float e = amo_potential_f32(pot, x);
    p->x = (double)x;
    p->e = (double)e;
    *e2 = e;
}

static inline float delta_log_target_density_f32(float e1, float e2, float beta)
{
    return ((-e2) * beta) - ((-e1) * beta);
}

/* Float32 state under a script-defined proposal / action.  The generic functions are the model's own (metropolis.jl:35-62); with
 * Particle{Float32} and Displacement{Float32} their bodies run under Julia's promotion rules, which C's usual arithmetic
 * conversions restate for the same text: x and delta are Float32, the parameters and the normal variate Float64, a Float32 x
 * Float32 product stays Float32, anything that meets a Float64 is Float64.  The hooks keep their double signatures (the chain
 * fields are doubles that hold Float32 values): the functions installed for a Float32 simulation take x and delta as floats
 * inside and return a Float32 value where the result is assigned to a field of type T (delta, x).  theta: the move's parameter
 * array (one entry for the one-parameter forms).  Lines: metropolis.jl:176-190. */
static inline void script_perform_action_f32(particle_t *p, float delta, int pot, float *e1, float *e2)
{
    *e1 = (float)p->e;
    float x = CUR_PERFORM ? (float)CUR_PERFORM(p->x, (double)delta) : (float)p->x + delta;
    float e = amo_potential_f32(pot, x);
    p->x = (double)x;
    p->e = (double)e;
    *e2 = e;
}
static inline float script_sample_f32(const particle_t *p, const double *theta, double z)
{
    return (float)(g_vec_logq ? g_vec_sample(z, p->x, theta) : CUR_SAMPLE(z, p->x, theta[0]));
}
static inline double script_logq_f32(float delta, const particle_t *p, const double *theta)
{
    return g_vec_logq ? g_vec_logq((double)delta, p->x, theta) : CUR_LOGQ((double)delta, p->x, theta[0]);
}
static inline void script_dlogq_f32(float delta, const particle_t *p, const double *theta, double *out)
{
    if (g_vec_logq) g_vec_dlogq((double)delta, p->x, theta, out);
    else out[0] = g_custom_dlogq ? CUR_DLOGQ((double)delta, p->x, theta[0]) : (0.0 / 0.0);
}
static inline float script_invert_f32(float delta, const particle_t *p)
{
    return CUR_INVERT ? (float)CUR_INVERT((double)delta, p->x) : -delta;
}
static inline int mc_step_script_f32(particle_t *p, move_t *m, const double *theta, int pot, double z, double u)
{
    float delta = script_sample_f32(p, theta, z);                  /* :177 sample_action! */
    double logq_f = script_logq_f32(delta, p, theta);              /* :178 at the old state */
    float e1, e2, beta = (float)p->beta;
    script_perform_action_f32(p, delta, pot, &e1, &e2);            /* :179 */
    float dlogp = delta_log_target_density_f32(e1, e2, beta);      /* :180 */
    delta = script_invert_f32(delta, p);                           /* :181 */
    double logq_b = script_logq_f32(delta, p, theta);              /* :182 at the new state */
    double alpha = julia_min(1.0, amo_exp((double)dlogp + logq_b - logq_f)); /* :183 */
    m->delta = (double)delta;
    if (alpha > u) return 1;                                       /* :184 */
    script_perform_action_f32(p, delta, pot, &e1, &e2);            /* :187 perform_action_cached! */
    return 0;
}

static inline int mc_step_f32(particle_t *p, move_t *m, double sigma, int pot, double z, double u)
{
    if (g_custom_logq) {                                           /* script-defined proposal */
        const double th[1] = { sigma };
        return mc_step_script_f32(p, m, th, pot, z, u);
    }
    double s_f = g_custom_scale_f32 ? sigma * (double)g_custom_scale_f32((float)p->x) : sigma;
    float delta = (float)(0.0 + s_f * z);                          /* :177; the field converts */
    double logq_forward = log_proposal_density_f32(delta, s_f);    /* :178 */
    float e1, e2, beta = (float)p->beta;
    perform_action_f32(p, delta, pot, &e1, &e2);                   /* :179 */
    float dlogp = delta_log_target_density_f32(e1, e2, beta);      /* :180 */
    delta = -delta;                                                /* :181 */
    double s_b = g_custom_scale_f32 ? sigma * (double)g_custom_scale_f32((float)p->x) : sigma;
    double logq_backward = log_proposal_density_f32(delta, s_b);   /* :182 */
    double alpha = julia_min(1.0, amo_exp((double)dlogp + logq_backward - logq_forward)); /* :183 */
    m->delta = (double)delta;
    if (alpha > u)                                                 /* :184 */
        return 1;
    perform_action_f32(p, delta, pot, &e1, &e2);                   /* :187 */
    return 0;
}

int amo_mc_step_explicit_f32(int pot, float beta, double sigma, double z, double u, float *x, float *e)
{
    particle_t p = { (double)*x, (double)beta, (double)*e };
    move_t m = { 0.0, 0, 0 };
    int a = mc_step_f32(&p, &m, sigma, pot, z, u);
    *x = (float)p.x; *e = (float)p.e;
    return a;
}

/* Switches a simulation to Float32 state: x, beta and e are rounded to Float32 now (Particle(Float32(x), Float32(beta)))
 * and stay Float32 values from here on. */
void amo_set_state_f32(amo_sim *s, int on)
{
    s->f32 = on ? 1 : 0;
    if (!s->f32) return;
    for (int64_t c = 0; c < s->M; ++c) {
        s->chains[c].x = (double)(float)s->chains[c].x;
        s->chains[c].beta = (double)(float)s->chains[c].beta;
        s->chains[c].e = (double)amo_potential_f32(s->pot, (float)s->chains[c].x);
    }
}

int amo_mc_step_explicit(int pot, double beta, double sigma, double z, double u,
                         double *x, double *e)
{
    particle_t p = { *x, beta, *e };
    move_t m = { 0.0, 0, 0 };
    int a = mc_step(&p, &m, sigma, pot, z, u);
    *x = p.x; *e = p.e;
    return a;
}

/* Third-party piece 3: rand(rng, Categorical(weights)) at metropolis.jl:206 is
 * Distributions.jl 0.25's DiscreteNonParametric sampler: one uniform `draw`,
 * cp = p[1]; i = 1; while cp <= draw && i < n: cp += p[i += 1].  0-based here. */
int amo_categorical(const double *weights, int K, double r)
{
    double cp = weights[0];
    int i = 0;
    while (cp <= r && i < K - 1) {
        i += 1;
        cp += weights[i];
    }
    return i;
}

static void draw4(const amo_sim *s, uint64_t pair, uint64_t t, uint32_t draw,
                  uint32_t stream, uint32_t out[4])
{
    uint32_t ctr[4], key[2] = { (uint32_t)s->seed, (uint32_t)(s->seed >> 32) };
    amo_counter(pair, t, draw, stream, ctr);
    amo_philox4x32_10(ctr, key, out);
}

/* metropolis.jl:203-212 mc_sweep! for chain c (global id g), steps [t0, t0+mc_steps). */
static void mc_sweep(amo_sim *s, int64_t c, uint64_t t0, int mc_steps)
{
    uint64_t g = (uint64_t)(s->offset + c);
    uint64_t pair = g >> 1;
    int half = (int)(g & 1u);
    particle_t *p = &s->chains[c];
    move_t *pool = &s->pools[c * s->K];
    const double *weights = s->weight;                             /* :204 */
    for (int i = 0; i < mc_steps; ++i) {                           /* :205 */
        uint64_t t = t0 + (uint64_t)i;
        uint32_t v[4], va[4];
        int id = 0;
        /* the step's two Philox draws are pure functions of (seed, pair, t): formed first, consumed in the
         * reference's order -- categorical pick (:206), proposal normal, accept uniform */
        draw4(s, pair, t, AMO_DRAW_NORMAL, AMO_STREAM_METROPOLIS, v);
        draw4(s, pair, t, AMO_DRAW_ACCEPT, AMO_STREAM_METROPOLIS, va);
        if (s->K > 1)                                              /* :206 */
            id = amo_categorical(weights, s->K, amo_uniform_pick(amo_spare_pick12(v, half), va[2 * half]));
        double zz[2];
        amo_box_muller(v, zz);
        double u = amo_uniform_accept(amo_spare_accept12(v, half), va[2 * half], va[2 * half + 1]);
        move_t *move = &pool[id];                                  /* :207 */
        set_class_of(id);
        if (g_vec_logq) {
            double th[AMO_MAX_NP];
            move_theta(s, id, th);
            move->accepted_calls += s->f32 ? mc_step_script_f32(p, move, th, s->pot, zz[half], u)
                                           : mc_step_vec(p, move, th, s->pot, zz[half], u);
        } else
        move->accepted_calls += s->f32 ? mc_step_f32(p, move, s->sigma[id], s->pot, zz[half], u)
                                       : mc_step(p, move, s->sigma[id], s->pot, zz[half], u); /* :208 */
        move->total_calls += 1;                                    /* :209 */
    }
}

amo_sim *amo_create(int64_t n_chains, int64_t chain_offset, int potential, double beta,
                    int n_moves, const double *sigma, const double *weight,
                    uint64_t seed, int sweepstep)
{
    if (n_chains < 0 || n_moves < 1 || sweepstep < 1) return NULL;
    amo_sim *s = (amo_sim *)calloc(1, sizeof(*s));
    s->M = n_chains; s->offset = chain_offset; s->pot = potential; s->K = n_moves;
    s->sweepstep = sweepstep; s->seed = seed;
    s->sigma = (double *)malloc(sizeof(double) * (size_t)n_moves);
    s->weight = (double *)malloc(sizeof(double) * (size_t)n_moves);
    s->theta_more = (double *)calloc((size_t)n_moves * (AMO_MAX_NP - 1), sizeof(double));
    memcpy(s->sigma, sigma, sizeof(double) * (size_t)n_moves);
    memcpy(s->weight, weight, sizeof(double) * (size_t)n_moves);
    s->chains = (particle_t *)calloc((size_t)(n_chains > 0 ? n_chains : 1), sizeof(particle_t));
    s->pools = (move_t *)calloc((size_t)(n_chains > 0 ? n_chains : 1) * (size_t)n_moves, sizeof(move_t));
    for (int64_t c = 0; c < n_chains; ++c) {
        s->chains[c].x = 0.0;
        s->chains[c].beta = beta;
        s->chains[c].e = amo_potential(potential, 0.0);
    }
    return s;
}

void amo_destroy(amo_sim *s)
{
    if (!s) return;
    free(s->sigma); free(s->theta_more); free(s->weight); free(s->chains); free(s->pools); free(s);
}

/* particle_1d.jl:13-15: Particle(x, beta) sets e = potential(x). */
void amo_set_x(amo_sim *s, const double *x)
{
    for (int64_t c = 0; c < s->M; ++c) {
        if (s->f32) {
            s->chains[c].x = (double)(float)x[c];
            s->chains[c].e = (double)amo_potential_f32(s->pot, (float)x[c]);
            continue;
        }
        s->chains[c].x = x[c];
        s->chains[c].e = amo_potential(s->pot, x[c]);
    }
}

void amo_set_beta(amo_sim *s, const double *beta)
{
    for (int64_t c = 0; c < s->M; ++c) s->chains[c].beta = s->f32 ? (double)(float)beta[c] : beta[c];
}

/* MC_harmonic_oscillator.jl:13  chains = [System(4rand(rng) - 2, beta) ...],
 * generalised to lo + (hi-lo)*u, u from the INIT stream of the chain's pair. */
void amo_init_uniform(amo_sim *s, double lo, double hi)
{
    for (int64_t c = 0; c < s->M; ++c) {
        uint64_t g = (uint64_t)(s->offset + c);
        int half = (int)(g & 1u);
        uint32_t v[4];
        draw4(s, g >> 1, 0, 0, AMO_STREAM_INIT, v);
        double u = amo_uniform_co(v[2 * half], v[2 * half + 1]);
        double x = lo + (hi - lo) * u;
        if (s->f32) {                      /* System(Float32(4rand(rng) - 2), beta) */
            s->chains[c].x = (double)(float)x;
            s->chains[c].e = (double)amo_potential_f32(s->pot, (float)x);
            continue;
        }
        s->chains[c].x = x;
        s->chains[c].e = amo_potential(s->pot, x);
    }
}

void amo_get_state(const amo_sim *s, double *x, double *e)
{
    for (int64_t c = 0; c < s->M; ++c) {
        if (x) x[c] = s->chains[c].x;
        if (e) e[c] = s->chains[c].e;
    }
}

void amo_get_counters(const amo_sim *s, int64_t *accepted, int64_t *total)
{
    for (int k = 0; k < s->K; ++k)
        for (int64_t c = 0; c < s->M; ++c) {
            if (accepted) accepted[k * s->M + c] = s->pools[c * s->K + k].accepted_calls;
            if (total) total[k * s->M + c] = s->pools[c * s->K + k].total_calls;
        }
}

void amo_set_sigma(amo_sim *s, int k, double sigma) { s->sigma[k] = sigma; }
void amo_set_theta(amo_sim *s, int k, int p, double v) { if (p == 0) s->sigma[k] = v; else s->theta_more[k * (AMO_MAX_NP - 1) + p - 1] = v; }
double amo_get_theta(const amo_sim *s, int k, int p) { return p == 0 ? s->sigma[k] : s->theta_more[k * (AMO_MAX_NP - 1) + p - 1]; }
double amo_get_sigma(const amo_sim *s, int k) { return s->sigma[k]; }
uint64_t amo_get_step(const amo_sim *s) { return s->t; }
void amo_set_step(amo_sim *s, uint64_t t) { s->t = t; }

int amo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* metropolis.jl:302-309 make_step!(simulation, ::Metropolis): map mc_sweep! over
 * chains; n_threads = 1 is `collect` (parallel=false), > 1 is tcollect (:265). */
void amo_make_step(amo_sim *s, int n_threads)
{
    uint64_t t0 = s->t;
    int steps = s->sweepstep;
    if (n_threads <= 1) {
        for (int64_t c = 0; c < s->M; ++c) mc_sweep(s, c, t0, steps);
    } else {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
        for (int64_t c = 0; c < s->M; ++c) mc_sweep(s, c, t0, steps);
    }
    s->t = t0 + (uint64_t)steps;
}

void amo_make_steps(amo_sim *s, int64_t n, int n_threads)
{
    /* n make_step!s.  Chains never interact (metropolis.jl:303-307), so with threads each chain
     * runs its n sweeps back to back (cache-resident); the result is identical to n barriers. */
    if (n_threads <= 1 || n <= 1) {
        for (int64_t i = 0; i < n; ++i) amo_make_step(s, n_threads);
        return;
    }
    const uint64_t t0 = s->t;
    const int steps = s->sweepstep;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
    for (int64_t c = 0; c < s->M; ++c)
        for (int64_t i = 0; i < n; ++i) mc_sweep(s, c, t0 + (uint64_t)i * (uint64_t)steps, steps);
    s->t = t0 + (uint64_t)n * (uint64_t)steps;
}

/* ------------------------------------------------------------------------ */
/* Reproducible sums.                                 */
/* The reference folds across chains with `+` in whatever order its reducer   */
/* takes (mean: particle_1d.jl:68-70, metropolis.jl:319-321; foldxl / foldxt: */
/* estimator.jl:94,113), so the order is not part of its semantics.  The      */
/* engine defines each cross-chain sum so that no order can enter: summands   */
/* are rounded once to multiples of a power of two fixed by order-independent */
/* facts, the multiples are added as integers, the total is rounded once.     */
/* This is that definition restated with integer arithmetic on the operands'  */
/* bit patterns (the kernels form the same integers with floating-point       */
/* additions into accumulators of a fixed binade).  The plain left-to-right   */
/* sums stay beside it (amo_*_plain) to pin the difference.                   */
/* ------------------------------------------------------------------------ */
typedef __int128 i128_t;
typedef unsigned __int128 u128_t;
enum { XS_WORDS = 12, XS_EMPTY = 0, XS_Q = 1, XS_R = 2, XS_PLAIN = 3, XS_NAN = 1, XS_PINF = 2, XS_NINF = 4,
       XS_LEVEL_BITS = 50, XS_LMIN = -20, XS_LMAX = 19, XS_E_RATIO = -50 };
typedef struct { int kind, e; unsigned flags; i128_t k1, k2; double plain; } xs_t;

static double lsb1(double v) { return u2d(d2u(v) | 1ull); }

/* round-half-even of |m| 2^e / 2^q for an unsigned integer m; *overflow set when the result needs more than 62 bits */
static int64_t rn_u128(u128_t m, int e, int q, int *overflow)
{
    int s = q - e;                       /* divide by 2^s */
    u128_t r;
    if (m == 0) return 0;
    if (s <= 0) {
        int bits = 0;
        for (u128_t t = m; t; t >>= 1) ++bits;
        if (bits - s > 62) { *overflow = 1; return 0; }
        r = m << (-s);
    } else if (s >= 128) {
        r = 0;
    } else {
        u128_t rem = m & ((((u128_t)1) << s) - 1), half = ((u128_t)1) << (s - 1);
        r = m >> s;
        if (rem > half || (rem == half && (r & 1))) r += 1;
        if (r >> 62) { *overflow = 1; return 0; }
    }
    return (int64_t)r;
}

/* finite x = (-1)^neg m 2^e with m an integer below 2^53 */
static void split_double(double x, int *neg, uint64_t *m, int *e)
{
    uint64_t b = d2u(x);
    int be = (int)((b >> 52) & 0x7FF);
    *neg = (int)(b >> 63);
    *m = b & ((1ull << 52) - 1);
    if (be == 0) *e = -1074;
    else { *m |= 1ull << 52; *e = be - 1075; }
}

/* RN(x / 2^q), x finite */
static int64_t rn_scaled(double x, int q, int *overflow)
{
    int neg, e;
    uint64_t m;
    split_double(x, &neg, &m, &e);
    int64_t r = rn_u128((u128_t)m, e, q, overflow);
    return neg ? -r : r;
}

/* RN(a b / 2^q): the exact product of two finite doubles, rounded once */
static int64_t rn_scaled_product(double a, double b, int q, int *overflow)
{
    int na, nb, ea, eb;
    uint64_t ma, mb;
    split_double(a, &na, &ma, &ea);
    split_double(b, &nb, &mb, &eb);
    int64_t r = rn_u128((u128_t)ma * (u128_t)mb, ea + eb, q, overflow);
    return (na != nb) ? -r : r;
}

static int is_finite(double v) { return ((d2u(v) >> 52) & 0x7FF) != 0x7FF; }

static void xs_flag_nonfinite(xs_t *a, double v)
{
    if (v != v) a->flags |= XS_NAN;
    else a->flags |= (v > 0) ? XS_PINF : XS_NINF;
}

/* kind Q: a->e is the quantum exponent.  Any summand that is not finite makes the sum NaN. */
static void xs_q_add(xs_t *a, double v)
{
    int ov = 0;
    if (!is_finite(v)) { a->flags |= XS_NAN; return; }
    a->k1 += rn_scaled(lsb1(v), a->e, &ov);
    if (ov) a->flags |= XS_NAN;
}
static void xs_q_add_product(xs_t *a, double x, double y)
{
    int ov = 0;
    if (!is_finite(x) || !is_finite(y)) { a->flags |= XS_NAN; return; }
    a->k1 += rn_scaled_product(lsb1(x), lsb1(y), a->e, &ov);
    if (ov) a->flags |= XS_NAN;
}

/* kind R: the level a finite value needs, max(LMIN, floor((ilogb(v) + 1) / 50)); zero and subnormals: LMIN */
static int xs_level_of(double v)
{
    int be = (int)((d2u(v) >> 52) & 0x7FF);
    if (be == 0) return XS_LMIN;
    int num = be - 1023 + 1;                          /* ilogb + 1 */
    int l = (num >= 0) ? num / XS_LEVEL_BITS : -((-num + XS_LEVEL_BITS - 1) / XS_LEVEL_BITS);
    return l < XS_LMIN ? XS_LMIN : l;
}
static void xs_r_raise(xs_t *a, int top)
{
    int d = top - a->e;
    if (d <= 0) return;
    a->k2 = (d == 1) ? a->k1 : 0;      /* one level up the level-1 multiples ARE the level-2 multiples; further up all round to 0 */
    a->k1 = 0;
    a->e = top;
}
static void xs_r_add(xs_t *a, double v)
{
    int ov = 0;
    if (!is_finite(v)) { xs_flag_nonfinite(a, v); return; }
    int l = xs_level_of(v);
    if (l > XS_LMAX) { a->flags |= (v > 0) ? XS_PINF : XS_NINF; return; }    /* |v| >= 2^999: beyond the last level */
    if (l > a->e) xs_r_raise(a, l);
    double v1 = lsb1(v);
    int64_t k1 = rn_scaled(v1, XS_LEVEL_BITS * a->e, &ov);
    double r = v1 - ldexp((double)k1, XS_LEVEL_BITS * a->e);      /* exact: the low part of v1 */
    int64_t k2 = rn_scaled(lsb1(r), XS_LEVEL_BITS * (a->e - 1), &ov);
    a->k1 += k1;
    a->k2 += k2;
    if (ov) a->flags |= XS_NAN;
}
static xs_t xs_new(int kind, int e)
{
    xs_t a;
    memset(&a, 0, sizeof(a));
    a.kind = kind;
    a.e = (kind == XS_R) ? XS_LMIN : e;
    return a;
}

/* K rounded to 53 significant bits (ties to even), times 2^e */
static double round_i128(i128_t K, int e)
{
    int neg = K < 0;
    u128_t m = neg ? (u128_t)(-K) : (u128_t)K;
    int bits = 0;
    for (u128_t t = m; t; t >>= 1) ++bits;
    double v;
    if (bits <= 53) v = ldexp((double)(uint64_t)m, e);
    else {
        int s = bits - 53;
        u128_t rem = m & ((((u128_t)1) << s) - 1), half = ((u128_t)1) << (s - 1);
        uint64_t q = (uint64_t)(m >> s);
        if (rem > half || (rem == half && (q & 1))) q += 1;
        v = ldexp((double)q, e + s);
    }
    return neg ? -v : v;
}
static double xs_value(const xs_t *a)
{
    if (a->kind == XS_EMPTY) return 0.0;
    if (a->kind == XS_PLAIN) return a->plain;
    if (a->flags) {
        if ((a->flags & XS_NAN) || ((a->flags & XS_PINF) && (a->flags & XS_NINF))) return 0.0 / 0.0;
        return (a->flags & XS_PINF) ? 1.0 / 0.0 : -1.0 / 0.0;
    }
    if (a->kind == XS_Q) return round_i128(a->k1, a->e);
    return round_i128(a->k1 * (((i128_t)1) << XS_LEVEL_BITS) + a->k2, XS_LEVEL_BITS * (a->e - 1));
}
static void xs_merge(xs_t *a, const xs_t *b_in)
{
    xs_t b = *b_in;
    if (b.kind == XS_EMPTY) return;
    if (a->kind == XS_EMPTY) { *a = b; return; }
    if (a->kind != b.kind || (a->kind == XS_Q && a->e != b.e)) { a->flags |= XS_NAN; return; }
    if (a->kind == XS_PLAIN) { a->plain += b.plain; return; }
    if (a->kind == XS_R) {
        if (b.e > a->e) xs_r_raise(a, b.e);
        else xs_r_raise(&b, a->e);
    }
    a->k1 += b.k1;
    a->k2 += b.k2;
    a->flags |= b.flags;
}
/* records: the transport form of include/amc.h (AMC_XSUM_WORDS doubles, 32-bit limbs, the top one signed) */
static void limbs_out(double *w, i128_t k)
{
    u128_t u = (u128_t)k;
    w[0] = (double)(uint32_t)u;
    w[1] = (double)(uint32_t)(u >> 32);
    w[2] = (double)(uint32_t)(u >> 64);
    w[3] = (double)(int32_t)(uint32_t)(u >> 96);
}
static i128_t limbs_in(const double *w)
{
    i128_t r = 0;
    for (int i = 3; i >= 0; --i) r = r * (((i128_t)1) << 32) + (i128_t)(int64_t)w[i];
    return r;
}
static void xs_to_record(const xs_t *a, double *rec)
{
    for (int i = 0; i < XS_WORDS; ++i) rec[i] = 0.0;
    rec[0] = (double)a->kind;
    if (a->kind == XS_PLAIN) { rec[11] = a->plain; return; }
    rec[1] = (double)a->e;
    rec[2] = (double)a->flags;
    if (a->flags) return;                  /* NaN / infinite: the integers mean nothing, canonically zero */
    limbs_out(rec + 3, a->k1);
    if (a->kind == XS_R) limbs_out(rec + 7, a->k2);
}
static xs_t xs_from_record(const double *rec)
{
    xs_t a;
    memset(&a, 0, sizeof(a));
    a.kind = (int)rec[0];
    a.e = (int)rec[1];
    a.flags = (unsigned)rec[2];
    a.k1 = limbs_in(rec + 3);
    a.k2 = limbs_in(rec + 7);
    a.plain = rec[11];
    return a;
}

/* exported: the definition applied to explicit operands (known-answer tests against an independent big-integer model) */
void amo_xsum_q(const double *v, int64_t n, int e, double *rec)
{
    xs_t a = xs_new(XS_Q, e);
    for (int64_t i = 0; i < n; ++i) xs_q_add(&a, v[i]);
    xs_to_record(&a, rec);
}
void amo_xsum_q_product(const double *x, const double *y, int64_t n, int e, double *rec)
{
    xs_t a = xs_new(XS_Q, e);
    for (int64_t i = 0; i < n; ++i) xs_q_add_product(&a, x[i], y[i]);
    xs_to_record(&a, rec);
}
void amo_xsum_r(const double *v, int64_t n, double *rec)
{
    xs_t a = xs_new(XS_R, 0);
    for (int64_t i = 0; i < n; ++i) xs_r_add(&a, v[i]);
    xs_to_record(&a, rec);
}
void amo_xsum_merge(double *into, const double *from)
{
    xs_t a = xs_from_record(into), b = xs_from_record(from);
    xs_merge(&a, &b);
    xs_to_record(&a, into);
}
double amo_xsum_round(const double *rec)
{
    xs_t a = xs_from_record(rec);
    return xs_value(&a);
}
/* GradientData of the Gaussian displacement policy: quantum exponents of (j, grad j, grad logq, g) from
 * 2^(es-1) <= sigma < 2^es and z^2 <= 72.1, alpha <= 1: the bounds 2^(2es+7), 2^(es+13), 2^(8-es), 2^(15-2es) of one
 * summand, each minus 46 (a lane of the engine adds 2^5 summands into an accumulator of 2^51 quanta). */
void amo_gd_exponents(double sigma, int e[4])
{
    int es = (int)((d2u(sigma) >> 52) & 0x7FF) - 1023 + 1;
    e[0] = 2 * es + 7 - 46;
    e[1] = es + 13 - 46;
    e[2] = 8 - es - 46;
    e[3] = 15 - 2 * es - 46;
}

/* The summands of the three sums over the state: the chain PAIRS' sums -- global chains 2p and 2p + 1 (shards begin at even
 * ids, so a pair never straddles two of them), a lone last chain by itself: fl(e_2p + e_2p+1), fl(x_2p + x_2p+1),
 * fl(fl(x_2p^2) + fl(x_2p+1^2)).  One of the orders in which the reference's `mean` may add, fixed by the chain ids alone. */
static void state_sums(const amo_sim *s, xs_t *se, xs_t *sx, xs_t *sxx)
{
    for (int64_t c = 0; c < s->M; c += 2) {
        const int two = c + 1 < s->M;
        const double x0 = s->chains[c].x, x1 = two ? s->chains[c + 1].x : 0.0;
        const double e0 = s->chains[c].e, e1 = two ? s->chains[c + 1].e : 0.0;
        if (se) xs_r_add(se, e0 + e1);
        if (sx) xs_r_add(sx, x0 + x1);
        if (sxx) xs_r_add(sxx, x0 * x0 + x1 * x1);
    }
}

/* The callbacks' sums over this simulation's chains as records, in the layout of amc_reduce: sum e, sum x, sum x^2 (kind R),
 * the count (plain), per move sum_c accepted_c / total_c (kind Q, quantum 2^-50; 0/0 = NaN like the reference). */
void amo_callback_records(const amo_sim *s, double *recs)
{
    xs_t se = xs_new(XS_R, 0), sx = xs_new(XS_R, 0), sxx = xs_new(XS_R, 0), cnt = xs_new(XS_PLAIN, 0);
    state_sums(s, &se, &sx, &sxx);
    cnt.plain = (double)s->M;
    xs_to_record(&se, recs);
    xs_to_record(&sx, recs + XS_WORDS);
    xs_to_record(&sxx, recs + 2 * XS_WORDS);
    xs_to_record(&cnt, recs + 3 * XS_WORDS);
    for (int k = 0; k < s->K; ++k) {
        xs_t r = xs_new(XS_Q, XS_E_RATIO);
        for (int64_t c = 0; c < s->M; ++c) {
            const move_t *m = &s->pools[c * s->K + k];
            xs_q_add(&r, (double)m->accepted_calls / (double)m->total_calls);
        }
        xs_to_record(&r, recs + (4 + k) * XS_WORDS);
    }
}

/* particle_1d.jl:68-70 callback_energy: mean(system.e for system in chains) -- the sum as defined above, divided by M. */
double amo_callback_energy(const amo_sim *s)
{
    xs_t se = xs_new(XS_R, 0);
    state_sums(s, &se, 0, 0);
    return xs_value(&se) / (double)s->M;
}
/* ... and as a left-to-right Float64 sum, one of the orders the reference's `mean` may take */
double amo_callback_energy_plain(const amo_sim *s)
{
    double acc = 0.0;
    for (int64_t c = 0; c < s->M; ++c) acc += s->chains[c].e;
    return acc / (double)s->M;
}

/* metropolis.jl:319-321 callback_acceptance: mean over chains of the per-chain
 * vector [accepted_calls / total_calls for move in pool]; 0/0 -> NaN. */
void amo_callback_acceptance(const amo_sim *s, double *out)
{
    for (int k = 0; k < s->K; ++k) {
        xs_t r = xs_new(XS_Q, XS_E_RATIO);
        for (int64_t c = 0; c < s->M; ++c) {
            const move_t *m = &s->pools[c * s->K + k];
            xs_q_add(&r, (double)m->accepted_calls / (double)m->total_calls);
        }
        out[k] = xs_value(&r) / (double)s->M;
    }
}
void amo_callback_acceptance_plain(const amo_sim *s, double *out)
{
    for (int k = 0; k < s->K; ++k) {
        double acc = 0.0;
        for (int64_t c = 0; c < s->M; ++c) {
            const move_t *m = &s->pools[c * s->K + k];
            acc += (double)m->accepted_calls / (double)m->total_calls;
        }
        out[k] = acc / (double)s->M;
    }
}

/* Statistic of test/distribution_test.jl:33-37 (mean/std of positions). */
void amo_moments(const amo_sim *s, double out[2])
{
    xs_t sx = xs_new(XS_R, 0), sxx = xs_new(XS_R, 0);
    state_sums(s, 0, &sx, &sxx);
    out[0] = xs_value(&sx); out[1] = xs_value(&sxx);
}

/* ------------------------------------------------------------------------ */
/* Policy-guided Monte Carlo: src/PolicyGuided/                              */
/* ------------------------------------------------------------------------ */

/* gradients.jl:93-109 pgmc_estimate after gradients.jl:117-121
 * sample_gradient_data, P = 1.  gd = (j, dj, dlogq_forward, g). */
static void pgmc_sample(particle_t *p, move_t *m, double sigma, int pot, double z, double gd[4])
{
    if (g_custom_logq) {
        /* script-defined proposal: value and sigma-derivative of the forward density at the old state (:97), of the
         * backward density at the new state (:102) */
        m->delta = CUR_SAMPLE(z, p->x, sigma);
        double logq_f = CUR_LOGQ(m->delta, p->x, sigma);
        double dlogq_f = g_custom_dlogq ? CUR_DLOGQ(m->delta, p->x, sigma) : (0.0 / 0.0);
        double e1, e2;
        script_perform_action(p, m, pot, &e1, &e2);
        double dlogp = delta_log_target_density(e1, p->beta, e2, p->beta);
        double r = g_custom_reward ? g_custom_reward(m->delta, p->x) : m->delta * m->delta;
        m->delta = CUR_INVERT ? CUR_INVERT(m->delta, p->x) : -m->delta;
        double logq_b = CUR_LOGQ(m->delta, p->x, sigma);
        double dlogq_b = g_custom_dlogq ? CUR_DLOGQ(m->delta, p->x, sigma) : (0.0 / 0.0);
        script_perform_action(p, m, pot, &e1, &e2);
        double alpha = julia_min(1.0, amo_exp(dlogp + logq_b - logq_f));
        double j = r * alpha;
        gd[0] = j;
        gd[1] = j * (alpha == 1.0 ? dlogq_f : dlogq_b);
        gd[2] = dlogq_f;
        gd[3] = dlogq_f * dlogq_f;
        return;
    }
    if (g_custom_scale) {
        /* state-dependent width: forward density / gradient at the old state, backward at the new one */
        double s_f = g_custom_scale(p->x), w_f = sigma * s_f;
        m->delta = 0.0 + w_f * z;
        double logq_f = amo_log_proposal_density(m->delta, w_f);
        double dlogq_f = grad_log_proposal_density_w(-(m->delta * m->delta), w_f, s_f);
        double e1, e2;
        perform_action(p, m, pot, &e1, &e2);
        double dlogp = delta_log_target_density(e1, p->beta, e2, p->beta);
        double r = g_custom_reward ? g_custom_reward(m->delta, p->x) : m->delta * m->delta;
        m->delta = -m->delta;
        double s_b = g_custom_scale(p->x), w_b = sigma * s_b;
        double logq_b = amo_log_proposal_density(m->delta, w_b);
        double dlogq_b = grad_log_proposal_density_w(-(m->delta * m->delta), w_b, s_b);
        perform_action(p, m, pot, &e1, &e2);
        double alpha = julia_min(1.0, amo_exp(dlogp + logq_b - logq_f));
        double j = r * alpha;
        gd[0] = j;
        gd[1] = j * (alpha == 1.0 ? dlogq_f : dlogq_b);
        gd[2] = dlogq_f;
        gd[3] = dlogq_f * dlogq_f;
        return;
    }
    m->delta = 0.0 + sigma * z;                                     /* :119 sample_action! */
    double logq_f = amo_log_proposal_density(m->delta, sigma);     /* :97 */
    double dlogq_f = amo_grad_log_proposal_density(m->delta, sigma);
    double e1, e2;
    perform_action(p, m, pot, &e1, &e2);                            /* :98 */
    double dlogp = delta_log_target_density(e1, p->beta, e2, p->beta); /* :99 */
    double r = g_custom_reward ? g_custom_reward(m->delta, p->x)    /* :100 reward(action, system): script-defined */
                               : m->delta * m->delta;               /*      particle_1d.jl:42-44 */
    m->delta = -m->delta;                                           /* :101 */
    double logq_b = amo_log_proposal_density(m->delta, sigma);     /* :102 */
    double dlogq_b = amo_grad_log_proposal_density(m->delta, sigma);
    perform_action(p, m, pot, &e1, &e2);                            /* :103 always reverts */
    double alpha = julia_min(1.0, amo_exp(dlogp + logq_b - logq_f)); /* :104 */
    double j = r * alpha;                                           /* :105 */
    gd[0] = j;
    gd[1] = j * (alpha == 1.0 ? dlogq_f : dlogq_b);                 /* :106 */
    gd[2] = dlogq_f;
    gd[3] = dlogq_f * dlogq_f;                                      /* :107 */
}

/* The same sample with Float32 state (promotion rules in the Float32 section above): reward (delta)^2 in Float32,
 * j = r * alpha in Float64. */
/* gradients.jl:93-109 pgmc_estimate with Float32 state under a script-defined policy (one parameter or several):
 * gd = [j, grad j [P], grad logq_forward [P], g [P][P] row by row]; P = 1: (j, dj, dlogq, g). */
static void pgmc_sample_script_f32(particle_t *p, move_t *m, const double *theta, int pot, double z, double *gd)
{
    const int P = g_vec_logq ? g_np : 1;
    double d_f[AMO_MAX_NP], d_b[AMO_MAX_NP];
    float delta = script_sample_f32(p, theta, z);
    double logq_f = script_logq_f32(delta, p, theta);              /* :97 forward, at the old state */
    script_dlogq_f32(delta, p, theta, d_f);
    float e1, e2, beta = (float)p->beta;
    script_perform_action_f32(p, delta, pot, &e1, &e2);            /* :98 */
    float dlogp = delta_log_target_density_f32(e1, e2, beta);      /* :99 */
    double r = g_custom_reward_f32 ? g_custom_reward_f32(delta, (float)p->x) : (double)(delta * delta);   /* :100, (delta)^2 in T */
    delta = script_invert_f32(delta, p);                           /* :101 */
    double logq_b = script_logq_f32(delta, p, theta);              /* :102 backward, at the new state */
    script_dlogq_f32(delta, p, theta, d_b);
    script_perform_action_f32(p, delta, pot, &e1, &e2);            /* :103 */
    m->delta = (double)delta;
    double alpha = julia_min(1.0, amo_exp((double)dlogp + logq_b - logq_f));   /* :104 */
    double j = r * alpha;                                          /* :105 */
    gd[0] = j;
    for (int a = 0; a < P; ++a) {
        gd[1 + a] = j * (alpha == 1.0 ? d_f[a] : d_b[a]);          /* :106 */
        gd[1 + P + a] = d_f[a];
        for (int b = 0; b < P; ++b) gd[1 + 2 * P + a * P + b] = d_f[a] * d_f[b];   /* :107 */
    }
}

static void pgmc_sample_f32(particle_t *p, move_t *m, double sigma, int pot, double z, double gd[4])
{
    if (g_custom_logq) {
        const double th[1] = { sigma };
        pgmc_sample_script_f32(p, m, th, pot, z, gd);
        return;
    }
    if (g_custom_scale_f32) {
        double s_f = (double)g_custom_scale_f32((float)p->x), w_f = sigma * s_f;
        float delta = (float)(0.0 + w_f * z);
        double logq_f = log_proposal_density_f32(delta, w_f);
        double dlogq_f = grad_log_proposal_density_w((double)(-(delta * delta)), w_f, s_f);
        float e1, e2, beta = (float)p->beta;
        perform_action_f32(p, delta, pot, &e1, &e2);
        float dlogp = delta_log_target_density_f32(e1, e2, beta);
        double r = g_custom_reward_f32 ? g_custom_reward_f32(delta, (float)p->x) : (double)(delta * delta);
        delta = -delta;
        double s_b = (double)g_custom_scale_f32((float)p->x), w_b = sigma * s_b;
        double logq_b = log_proposal_density_f32(delta, w_b);
        double dlogq_b = grad_log_proposal_density_w((double)(-(delta * delta)), w_b, s_b);
        perform_action_f32(p, delta, pot, &e1, &e2);
        m->delta = (double)delta;
        double alpha = julia_min(1.0, amo_exp((double)dlogp + logq_b - logq_f));
        double j = r * alpha;
        gd[0] = j;
        gd[1] = j * (alpha == 1.0 ? dlogq_f : dlogq_b);
        gd[2] = dlogq_f;
        gd[3] = dlogq_f * dlogq_f;
        return;
    }
    float delta = (float)(0.0 + sigma * z);
    double logq_f = log_proposal_density_f32(delta, sigma);
    double dlogq_f = grad_log_proposal_density_f32(delta, sigma);
    float e1, e2, beta = (float)p->beta;
    perform_action_f32(p, delta, pot, &e1, &e2);
    float dlogp = delta_log_target_density_f32(e1, e2, beta);
    double r = g_custom_reward_f32 ? g_custom_reward_f32(delta, (float)p->x) : (double)(delta * delta);
    delta = -delta;
    double logq_b = log_proposal_density_f32(delta, sigma);
    double dlogq_b = grad_log_proposal_density_f32(delta, sigma);
    perform_action_f32(p, delta, pot, &e1, &e2);
    m->delta = (double)delta;
    double alpha = julia_min(1.0, amo_exp((double)dlogp + logq_b - logq_f));
    double j = r * alpha;
    gd[0] = j;
    gd[1] = j * (alpha == 1.0 ? dlogq_f : dlogq_b);
    gd[2] = dlogq_f;
    gd[3] = dlogq_f * dlogq_f;
}

/* ---- the estimator's summands by the ARITHMETIC SPEC of DESIGN.md section 3.6b --------------------------------
 * The Gaussian policy's four GradientData summands enter nothing but `+` folds whose order the reference leaves open,
 * and the engine forms them with two of the reference's sub-expressions replaced by forms that agree to a few ulp
 * (amc_kernels.h pg_sample): alpha = exp(min(dlogp, 0)) without the detour through logq_b - logq_f (which cancels bit
 * for bit), and d logq / d sigma = d^2 c3 - 1/sigma as one fma chain with c3 = dden / den^2 split hi + lo.  For the fold
 * to be bit-reproducible against the engine the oracle restates THAT arithmetic here, operation for operation;
 * pgmc_sample above is the reference-ordered form (gradients.jl:93-109), and the tests pin the two within a few ulp per
 * sample (test_pg_sample_summands_within_ulps).  The position update is the reference's in both. */
typedef struct { double sigma, c3hi, c3lo, c1; } spec_consts_t;
static spec_consts_t spec_consts(double sigma)
{
    const double TWO_PI = 0x1.921fb54442d18p+2;
    spec_consts_t k;
    double s2 = sigma * sigma, ds2 = sigma + sigma;
    double den = 2.0 * s2, dden = 2.0 * ds2;
    double av = TWO_PI * s2;
    k.sigma = sigma;
    k.c1 = ((TWO_PI * ds2) / av) / 2.0;                 /* d/dsigma of log(2 pi sigma^2) / 2 by ForwardDiff's rules */
    double d_hi = den * den, d_lo = fma(den, den, -d_hi);
    k.c3hi = dden / d_hi;
    double res = fma(-k.c3hi, d_hi, dden) - k.c3hi * d_lo;
    k.c3lo = res / d_hi;
    return k;
}
static double spec_alpha(double arg)
{
    /* min(1, exp(arg)) with Julia's NaN-keeping min; the spec's exp has exp(0) == 1 and exp(arg <= 0) <= 1 exactly */
    if (arg != arg) return arg;
    if (arg >= 0.0) return 1.0;
    if (arg >= -708.0) return amo_exp(arg);
    return 0.0;
}
/* gd = (j, d logq / d sigma): the two numbers the four summands are made of (grad j = j dlogq, g = dlogq^2) */
static void pgmc_sample_spec(particle_t *p, move_t *m, const spec_consts_t *k, int pot, double z, double *j, double *dlogq)
{
    m->delta = fma(k->sigma, z, 0.0);                               /* 0.0 + sigma*z, bit for bit */
    double e1, e2;
    perform_action(p, m, pot, &e1, &e2);
    double dlogp = delta_log_target_density(e1, p->beta, e2, p->beta);
    double d2 = m->delta * m->delta;
    double r = g_custom_reward ? g_custom_reward(m->delta, p->x) : d2;
    m->delta = -m->delta;
    perform_action(p, m, pot, &e1, &e2);                            /* always reverts: x = (x + d) + (-d) */
    *dlogq = fma(d2, k->c3hi, fma(d2, k->c3lo, -k->c1));
    *j = r * spec_alpha(dlogp);
}
static void pgmc_sample_spec_f32(particle_t *p, move_t *m, const spec_consts_t *k, int pot, double z, double *j, double *dlogq)
{
    float delta = (float)fma(k->sigma, z, 0.0);
    float e1, e2, beta = (float)p->beta;
    perform_action_f32(p, delta, pot, &e1, &e2);
    float dlogp = delta_log_target_density_f32(e1, e2, beta);
    float d2t = delta * delta;
    double d2 = (double)d2t;
    double r = g_custom_reward_f32 ? g_custom_reward_f32(delta, (float)p->x) : d2;
    delta = -delta;
    perform_action_f32(p, delta, pot, &e1, &e2);
    m->delta = (double)delta;
    *dlogq = fma(d2, k->c3hi, fma(d2, k->c3lo, -k->c1));
    *j = r * spec_alpha((double)dlogp);
}
/* exported for the per-sample comparison with the reference-ordered form */
void amo_pg_summands_spec(int pot, double beta, double sigma, double z, double *x, double out[4])
{
    particle_t p = { *x, beta, amo_potential(pot, *x) };
    move_t m = { 0.0, 0, 0 };
    spec_consts_t k = spec_consts(sigma);
    double j, d;
    pgmc_sample_spec(&p, &m, &k, pot, z, &j, &d);
    out[0] = j; out[1] = j * d; out[2] = d; out[3] = d * d;
    *x = p.x;
}
void amo_pg_summands_reference(int pot, double beta, double sigma, double z, double *x, double out[4])
{
    particle_t p = { *x, beta, amo_potential(pot, *x) };
    move_t m = { 0.0, 0, 0 };
    pgmc_sample(&p, &m, sigma, pot, z, out);
    *x = p.x;
}

/* Which kind of reproducible sum the fold is (amc_kernels.h PgKind): quanta from sigma for the Gaussian displacement
 * policy on a built-in potential with the model's reward -- every summand is then bounded by a function of sigma --,
 * running top as soon as a script-defined expression takes part. */
static int pg_fold_bounded(const amo_sim *s)
{
    return !(g_custom_logq || g_custom_scale || g_custom_scale_f32 || g_custom_reward || g_custom_reward_f32 ||
             s->pot == AMO_POT_CUSTOM);
}

/* estimator.jl:111-134 make_step!(::PolicyGradientEstimator): for each learnable
 * move, the `+` fold of GradientData (gradients.jl:68-76) over chains x q_batch
 * samples -- as a reproducible sum (above).  The caller owns the running accumulators (:130-131).  Draws come
 * from the ESTIMATOR stream (the reference replays the sampler's seeds,
 * estimator.jl:91-92,107 -- a quirk that is deliberately not reproduced).
 * recs: n_learn x 5 records (j, grad j, grad logq, g, n). */
void amo_pg_estimate_records(amo_sim *s, int n_learn, const int *learn_ids, int q_batch, double *recs)
{
    uint64_t t = s->t_est;
    const int bounded = pg_fold_bounded(s);
    const int script = (g_custom_logq || g_custom_scale || g_custom_scale_f32) ? 1 : 0;
    for (int l = 0; l < n_learn; ++l) {
        int lid = learn_ids[l];
        int ex[4] = { 0, 0, 0, 0 };
        if (bounded) amo_gd_exponents(s->sigma[lid], ex);
        xs_t col[4];
        for (int i = 0; i < 4; ++i) col[i] = xs_new(bounded ? XS_Q : XS_R, ex[i]);
        spec_consts_t k = spec_consts(s->sigma[lid]);
        int64_t n = 0;
        /* Running-top (kind R) columns take ONE summand per chain PAIR, sample and column: fl(s_even + s_odd), the two chains that
         * share a Box-Muller draw -- as the callback sums take theirs (xs_r_add(se, e0 + e1) above).  `held`: the even chain's
         * summands of this move, waiting for its partner's (a shard starts on an even global id; a lone last chain adds + 0.0). */
        double (*held)[4] = (double (*)[4])malloc((size_t)(q_batch > 0 ? q_batch : 1) * sizeof(double[4]));
        for (int64_t c = 0; c < s->M; ++c) {
            uint64_t g = (uint64_t)(s->offset + c);
            int half = (int)(g & 1u);
            const int last_alone = !half && c + 1 == s->M;
            for (int q = 0; q < q_batch; ++q) {
                uint32_t v[4];
                double zz[2];
                draw4(s, g >> 1, t, (uint32_t)(l * q_batch + q), AMO_STREAM_ESTIMATOR, v);
                amo_box_muller(v, zz);
                particle_t *p = &s->chains[c];
                move_t *m = &s->pools[c * s->K + lid];
                set_class_of(lid);
                if (script) {
                    /* script-defined policies: the reference's operations (the kernels mirror them), products rounded */
                    double gd[4];
                    if (s->f32) pgmc_sample_f32(p, m, s->sigma[lid], s->pot, zz[half], gd);
                    else pgmc_sample(p, m, s->sigma[lid], s->pot, zz[half], gd);
                    for (int i = 0; i < 4; ++i) {
                        if (!half) held[q][i] = gd[i];
                        if (half) xs_r_add(&col[i], (c > 0 ? held[q][i] : 0.0) + gd[i]);       /* (c == 0 odd: a shard that starts inside a pair) */
                        else if (last_alone) xs_r_add(&col[i], gd[i] + 0.0);
                    }
                } else {
                    double j, d;
                    if (s->f32) pgmc_sample_spec_f32(p, m, &k, s->pot, zz[half], &j, &d);
                    else pgmc_sample_spec(p, m, &k, s->pot, zz[half], &j, &d);
                    if (bounded) {
                        xs_q_add(&col[0], j);
                        xs_q_add_product(&col[1], j, d);
                        xs_q_add(&col[2], d);
                        xs_q_add_product(&col[3], d, d);
                    } else {
                        const double gd[4] = { j, j * d, d, d * d };
                        for (int i = 0; i < 4; ++i) {          /* the pair's sum, as above */
                            if (!half) held[q][i] = gd[i];
                            if (half) xs_r_add(&col[i], (c > 0 ? held[q][i] : 0.0) + gd[i]);
                            else if (last_alone) xs_r_add(&col[i], gd[i] + 0.0);
                        }
                    }
                }
                n += 1;
            }
        }
        free(held);
        for (int i = 0; i < 4; ++i) xs_to_record(&col[i], recs + (size_t)(l * 5 + i) * XS_WORDS);
        xs_t cnt = xs_new(XS_PLAIN, 0);
        cnt.plain = (double)n;
        xs_to_record(&cnt, recs + (size_t)(l * 5 + 4) * XS_WORDS);
    }
    s->t_est = t + 1;
}

void amo_pg_estimate(amo_sim *s, int n_learn, const int *learn_ids, int q_batch, double *out)
{
    double *recs = (double *)malloc((size_t)(n_learn > 0 ? n_learn : 1) * 5 * XS_WORDS * sizeof(double));
    amo_pg_estimate_records(s, n_learn, learn_ids, q_batch, recs);
    for (int i = 0; i < n_learn * 5; ++i) out[i] = amo_xsum_round(recs + (size_t)i * XS_WORDS);
    free(recs);
}

/* The same call with the reference-ordered summands (pgmc_sample) folded left to right in Float64: one of the orders the
 * reference's reducer may take (foldxl).  Pins the reproducible fold above at rtol 1e-10 in the tests. */
void amo_pg_estimate_plain(amo_sim *s, int n_learn, const int *learn_ids, int q_batch, double *out)
{
    uint64_t t = s->t_est;
    for (int l = 0; l < n_learn; ++l) {
        int lid = learn_ids[l];
        double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
        int64_t n = 0;
        for (int64_t c = 0; c < s->M; ++c) {
            uint64_t g = (uint64_t)(s->offset + c);
            int half = (int)(g & 1u);
            for (int q = 0; q < q_batch; ++q) {
                uint32_t v[4];
                double zz[2], gd[4];
                draw4(s, g >> 1, t, (uint32_t)(l * q_batch + q), AMO_STREAM_ESTIMATOR, v);
                amo_box_muller(v, zz);
                set_class_of(lid);
                if (s->f32)
                    pgmc_sample_f32(&s->chains[c], &s->pools[c * s->K + lid], s->sigma[lid], s->pot, zz[half], gd);
                else
                    pgmc_sample(&s->chains[c], &s->pools[c * s->K + lid], s->sigma[lid], s->pot, zz[half], gd);
                for (int i = 0; i < 4; ++i) acc[i] += gd[i];
                n += 1;
            }
        }
        for (int i = 0; i < 4; ++i) out[l * 5 + i] = acc[i];
        out[l * 5 + 4] = (double)n;
    }
    s->t_est = t + 1;
}

/* ---- a policy with several parameters ----------------------------------------------------------------------------- */
/* gradients.jl:93-109 pgmc_estimate: gd = [j, grad j [P], grad logq_forward [P], g [P][P] row by row] */
static void pgmc_sample_vec(particle_t *p, move_t *m, const double *theta, int pot, double z, double *gd)
{
    const int P = g_np;
    double d_f[AMO_MAX_NP], d_b[AMO_MAX_NP];
    m->delta = g_vec_sample(z, p->x, theta);
    double logq_f = g_vec_logq(m->delta, p->x, theta);                 /* :97 forward, at the old state */
    g_vec_dlogq(m->delta, p->x, theta, d_f);
    double e1, e2;
    script_perform_action(p, m, pot, &e1, &e2);                        /* :98 */
    double dlogp = delta_log_target_density(e1, p->beta, e2, p->beta); /* :99 */
    double r = g_custom_reward ? g_custom_reward(m->delta, p->x) : m->delta * m->delta;   /* :100 */
    m->delta = CUR_INVERT ? CUR_INVERT(m->delta, p->x) : -m->delta;              /* :101 */
    double logq_b = g_vec_logq(m->delta, p->x, theta);                 /* :102 backward, at the new state */
    g_vec_dlogq(m->delta, p->x, theta, d_b);
    script_perform_action(p, m, pot, &e1, &e2);                        /* :103 perform_action_cached! */
    double alpha = julia_min(1.0, amo_exp(dlogp + logq_b - logq_f));   /* :104 */
    double j = r * alpha;                                              /* :105 */
    gd[0] = j;
    for (int a = 0; a < P; ++a) {
        gd[1 + a] = j * (alpha == 1.0 ? d_f[a] : d_b[a]);              /* :106 */
        gd[1 + P + a] = d_f[a];
        for (int b = 0; b < P; ++b) gd[1 + 2 * P + a * P + b] = d_f[a] * d_f[b];   /* :107 */
    }
}

/* estimator.jl:111-134 for such a policy: recs = n_learn x (2 + 2P + P^2) records, fields as above and n last; every field a
 * running-top reproducible sum (the two triangles of g are the same sums: a product commutes). */
void amo_pg_estimate_records_vec(amo_sim *s, int n_learn, const int *learn_ids, int q_batch, double *recs)
{
    const int P = g_np, nf = 1 + 2 * P + P * P, stride = nf + 1;
    uint64_t t = s->t_est;
    for (int l = 0; l < n_learn; ++l) {
        int lid = learn_ids[l];
        double th[AMO_MAX_NP];
        move_theta(s, lid, th);
        xs_t col[1 + 2 * AMO_MAX_NP + AMO_MAX_NP * AMO_MAX_NP];
        for (int i = 0; i < nf; ++i) col[i] = xs_new(XS_R, 0);
        int64_t n = 0;
        /* one summand per chain PAIR, sample and field: fl(s_even + s_odd) (see amo_pg_estimate_records) */
        enum { NF_MAX = 1 + 2 * AMO_MAX_NP + AMO_MAX_NP * AMO_MAX_NP };
        double (*held)[NF_MAX] = (double (*)[NF_MAX])malloc((size_t)(q_batch > 0 ? q_batch : 1) * sizeof(double[NF_MAX]));
        for (int64_t c = 0; c < s->M; ++c) {
            uint64_t g = (uint64_t)(s->offset + c);
            int half = (int)(g & 1u);
            const int last_alone = !half && c + 1 == s->M;
            for (int q = 0; q < q_batch; ++q) {
                uint32_t v[4];
                double zz[2], gd[NF_MAX];
                draw4(s, g >> 1, t, (uint32_t)(l * q_batch + q), AMO_STREAM_ESTIMATOR, v);
                amo_box_muller(v, zz);
                if (s->f32) pgmc_sample_script_f32(&s->chains[c], &s->pools[c * s->K + lid], th, s->pot, zz[half], gd);
                else pgmc_sample_vec(&s->chains[c], &s->pools[c * s->K + lid], th, s->pot, zz[half], gd);
                for (int i = 0; i < nf; ++i) {
                    if (!half) held[q][i] = gd[i];
                    if (half) xs_r_add(&col[i], (c > 0 ? held[q][i] : 0.0) + gd[i]);
                    else if (last_alone) xs_r_add(&col[i], gd[i] + 0.0);
                }
                n += 1;
            }
        }
        free(held);
        for (int i = 0; i < nf; ++i) xs_to_record(&col[i], recs + (size_t)(l * stride + i) * XS_WORDS);
        xs_t cnt = xs_new(XS_PLAIN, 0);
        cnt.plain = (double)n;
        xs_to_record(&cnt, recs + (size_t)(l * stride + nf) * XS_WORDS);
    }
    s->t_est = t + 1;
}

/* inv(A), P x P, P <= 4: Gauss-Jordan elimination with partial pivoting -- the sequence of amc::pg_inv_small (amc_kernels.h),
 * operation for operation.  (Julia's inv is LAPACK getrf / getri: the same pivots, another order of the eliminations.) */
int amo_inv_small(const double *A, int np, double *inv)
{
    double m[AMO_MAX_NP][2 * AMO_MAX_NP];
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < np; ++j) { m[i][j] = A[i * np + j]; m[i][np + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < np; ++c) {
        int piv = c;
        double best = fabs(m[c][c]);
        for (int r = c + 1; r < np; ++r)
            if (fabs(m[r][c]) > best) { best = fabs(m[r][c]); piv = r; }
        if (!(best > 0.0) || !(best <= 1.7976931348623157e308)) return 0;
        if (piv != c)
            for (int j = 0; j < 2 * np; ++j) { double t = m[c][j]; m[c][j] = m[piv][j]; m[piv][j] = t; }
        double d = m[c][c];
        for (int j = 0; j < 2 * np; ++j) m[c][j] = m[c][j] / d;
        for (int r = 0; r < np; ++r) {
            if (r == c) continue;
            double f = m[r][c];
            for (int j = 0; j < 2 * np; ++j) m[r][j] = m[r][j] - f * m[c][j];
        }
    }
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < np; ++j) inv[i * np + j] = m[i][np + j];
    return 1;
}

/* learning.jl:32-34,50-52,77-79,103-105,130-134,160-164 learning_step! on arrays: gd = the AVERAGED GradientData
 * [j, grad j, grad logq_forward, g]; theta is updated in place.  `eta * inv(F) * v` is (eta * inv(F)) * v; matrix-vector products
 * and dot add their terms in index order.  0: F = g + eps I is singular (nothing applied). */
int amo_learning_step_vec(int opt, double h0, double h1, int np, const double *gd, double *theta)
{
    double j = gd[0];
    const double *dj = gd + 1, *dl = gd + 1 + np, *g = gd + 1 + 2 * np;
    double v[AMO_MAX_NP], step[AMO_MAX_NP], eta = h0;
    int baseline = opt == AMO_OPT_BLPG || opt == AMO_OPT_BLAPG || opt == AMO_OPT_BLANPG;
    for (int p = 0; p < np; ++p) v[p] = baseline ? dj[p] - j * dl[p] : dj[p];
    if (opt == AMO_OPT_VPG || opt == AMO_OPT_BLPG || opt == AMO_OPT_BLAPG) {
        if (opt == AMO_OPT_BLAPG) {
            double dot = dj[0] * dj[0];
            for (int p = 1; p < np; ++p) dot = dot + dj[p] * dj[p];
            eta = sqrt(2.0 * h0 / (dot + h1));
        }
        for (int p = 0; p < np; ++p) step[p] = eta * v[p];
    } else if (opt == AMO_OPT_NPG || opt == AMO_OPT_ANPG || opt == AMO_OPT_BLANPG) {
        double F[AMO_MAX_NP * AMO_MAX_NP], Fi[AMO_MAX_NP * AMO_MAX_NP];
        for (int a = 0; a < np; ++a)
            for (int b = 0; b < np; ++b) F[a * np + b] = a == b ? g[a * np + b] + h1 * 1.0 : g[a * np + b];
        if (!amo_inv_small(F, np, Fi)) return 0;
        if (opt != AMO_OPT_NPG) {
            double w[AMO_MAX_NP] = { 0.0, 0.0, 0.0, 0.0 };
            for (int a = 0; a < np; ++a) {
                double t = Fi[a * np] * v[0];
                for (int b = 1; b < np; ++b) t = t + Fi[a * np + b] * v[b];
                w[a] = t;
            }
            double dot = v[0] * w[0];
            for (int p = 1; p < np; ++p) dot = dot + v[p] * w[p];
            eta = sqrt(2.0 * h0 / dot);
        }
        for (int a = 0; a < np; ++a) {
            double t = (eta * Fi[a * np]) * v[0];
            for (int b = 1; b < np; ++b) t = t + (eta * Fi[a * np + b]) * v[b];
            step[a] = t;
        }
    } else {
        return 1;
    }
    for (int p = 0; p < np; ++p) theta[p] = theta[p] + step[p];
    return 1;
}

/* learning.jl:32-34,50-52,77-79,103-105,130-134,160-164 learning_step! for P = 1;
 * gd is the AVERAGED GradientData (gradients.jl:83-85).  inv(g + eps*I) is a
 * scalar reciprocal here. */
double amo_learning_step(int opt, double h0, double h1, double theta, const double gd[4])
{
    double j = gd[0], dj = gd[1], dlogq = gd[2], g = gd[3];
    switch (opt) {
    case AMO_OPT_VPG:                       /* eta = h0 */
        return theta + h0 * dj;
    case AMO_OPT_BLPG:
        return theta + h0 * (dj - j * dlogq);
    case AMO_OPT_BLAPG: {                   /* delta = h0, eps = h1 */
        double eta = sqrt(2.0 * h0 / (dj * dj + h1));
        return theta + eta * (dj - j * dlogq);
    }
    case AMO_OPT_NPG: {                     /* eta = h0, eps = h1 */
        double Finv = 1.0 / (g + h1 * 1.0);
        return theta + h0 * Finv * dj;
    }
    case AMO_OPT_ANPG: {
        double Finv = 1.0 / (g + h1 * 1.0);
        double eta = sqrt(2.0 * h0 / (dj * (Finv * dj)));
        return theta + eta * Finv * dj;
    }
    case AMO_OPT_BLANPG: {
        double Finv = 1.0 / (g + h1 * 1.0);
        double bj = dj - j * dlogq;
        double eta = sqrt(2.0 * h0 / (bj * (Finv * bj)));
        return theta + eta * Finv * bj;
    }
    default:                                /* Static: never in learn_ids, estimator.jl:72 */
        return theta;
    }
}

/* ------------------------------------------------------------------------ */
/* simulation.jl:95-117 build_schedule (three methods)                       */
/* ------------------------------------------------------------------------ */
int64_t amo_build_schedule_linear(int64_t steps, int64_t burn, int64_t dt, int64_t *out, int64_t cap)
{
    int64_t n = 0, last = -1;
    for (int64_t t = burn; t <= steps; t += dt) {          /* collect(burn:dt:steps) */
        if (n < cap) out[n] = t;
        n++; last = t;
    }
    if (last != steps) {                                   /* ∪ [steps] */
        if (n < cap) out[n] = steps;
        n++;
    }
    return n;
}

static int64_t push_unique(int64_t *out, int64_t n, int64_t cap, int64_t v)
{
    for (int64_t i = (n > 64 ? n - 64 : 0); i < n && i < cap; ++i)
        if (out[i] == v) return n;          /* duplicates only ever sit next to each other */
    if (n < cap) out[n] = v;
    return n + 1;
}

int64_t amo_build_schedule_block(int64_t steps, int64_t burn, const int64_t *block, int n_block,
                                 int64_t *out, int64_t cap)
{
    int64_t last = block[n_block - 1];
    int64_t nblock = (steps - burn) / last;                /* :114 */
    int64_t n = 0;
    for (int64_t m = 1; m <= nblock; ++m)                  /* :115 */
        for (int b = 0; b < n_block; ++b) {
            int64_t v = block[b] + burn + (m - 1) * last;
            if (v <= steps) n = push_unique(out, n, cap, v);   /* :116 filter + unique */
        }
    n = push_unique(out, n, cap, steps);
    return n;
}

int64_t amo_build_schedule_log(int64_t steps, int64_t burn, double base, int64_t *out, int64_t cap)
{
    int64_t n = 0;
    n = push_unique(out, n, cap, burn);                    /* :105 */
    int nmax = (int)floor(log((double)(steps - burn)) / log(base));
    for (int k = 0; k <= nmax; ++k) {
        double p = pow(base, (double)k);
        if (p != floor(p)) return -1;                      /* Int(base^n) throws InexactError */
        n = push_unique(out, n, cap, burn + (int64_t)p);
    }
    n = push_unique(out, n, cap, steps);
    return n;
}

/* Statistic of test/distribution_test.jl:9-39: run `steps` sweeps, pool the positions of all
 * chains at every scheduled sample time (burn, burn+dt, ..., StoreTrajectories' rows) and
 * return n, sum x, sum x^2 of the pooled sample.  Also returns the time average of
 * callback_energy over the same schedule (pgmc_test.jl:45 style) in out[3]. */
void amo_run_pooled_moments(amo_sim *s, int64_t steps, int64_t burn, int64_t dt, int n_threads, double out[4])
{
    /* chains are independent (metropolis.jl:303-307), so each chain runs all its steps in turn */
    double n = 0.0, sx = 0.0, sxx = 0.0, se = 0.0;
    const uint64_t t0 = s->t;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 1 ? n_threads : 1) reduction(+ : n, sx, sxx, se)
#endif
    for (int64_t c = 0; c < s->M; ++c) {
        for (int64_t t = 1; t <= steps; ++t) {
            mc_sweep(s, c, t0 + (uint64_t)(t - 1) * (uint64_t)s->sweepstep, s->sweepstep);
            if (t >= burn && (t - burn) % dt == 0) {
                double x = s->chains[c].x;
                sx += x; sxx += x * x; n += 1.0; se += s->chains[c].e;
            }
        }
    }
    s->t = t0 + (uint64_t)steps * (uint64_t)s->sweepstep;
    out[0] = n; out[1] = sx; out[2] = sxx; out[3] = se / n;
}

/* Resume helpers (the reference's StoreBackups is write-only: src/algorithms.jl:264-303). */
void amo_set_counters(amo_sim *s, const int64_t *accepted, const int64_t *total)
{
    for (int k = 0; k < s->K; ++k)
        for (int64_t c = 0; c < s->M; ++c) {
            s->pools[c * s->K + k].accepted_calls = accepted[k * s->M + c];
            s->pools[c * s->K + k].total_calls = total[k * s->M + c];
        }
}

uint64_t amo_get_estimator_step(const amo_sim *s) { return s->t_est; }
void amo_set_estimator_step(amo_sim *s, uint64_t t) { s->t_est = t; }
