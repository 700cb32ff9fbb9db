// amc_xsum.h -- reproducible cross-chain sums: host + device.
//
// Everything that crosses chains on the path is a SUM: callback_energy (example/particle_1d/particle_1d.jl:68-70),
// callback_acceptance (src/metropolis.jl:319-321), the GradientData fold (src/PolicyGuided/estimator.jl:113-131,
// gradients.jl:68-76).  The reference adds Float64s in whatever order its reducer takes (foldxl / foldxt / mean), so its own
// result depends on the thread count.  Here every such sum is defined so that it does NOT depend on the order of the additions,
// hence not on the grid, the number of shards or the number of GPUs: each summand is rounded ONCE to a multiple of a power of
// two (its "quantum") by a rule that depends on the summand and on order-independent facts alone, the multiples are added as
// INTEGERS (exact, associative), and the total is rounded once to Float64.
//
//   lsb1(v)     v with the last bit of its significand set (bits | 1): no summand can then lie exactly half way between two
//               multiples of a quantum (ties would be broken by the parity of the running sum, i.e. by the order)
//   RN(.)       round to nearest integer
//   RN53(K)     the integer K rounded to 53 significant bits, ties to even
//
// kind Q ("quantum known a priori", exponent E): for columns whose summands are bounded by a function of the parameters --
//     K = sum_i RN(lsb1(v_i) / 2^E)            plain columns
//     K = sum_i RN(lsb1(a_i) lsb1(b_i) / 2^E)  product columns (the EXACT product, rounded once)
//     result = RN53(K) 2^E; NaN as soon as one summand is not finite.
//   Lanes form it with ONE f64 add (or fma) per summand: an accumulator S that starts at C = 1.5 * 2^(E+52) has ulp 2^E while
//   it stays in [2^(E+52), 2^(E+53)), so S + v rounds v to a multiple of 2^E and (bits(S) - bits(C)) IS the integer sum.
//   E of each column: xs_gd_exponents() below (GradientData of the Gaussian policy, from sigma) and XS_E_RATIO.
// kind R ("running top"): for unbounded columns (sum e, sum x, sum x^2, script-defined estimators) -- two integers on an
//   ABSOLUTE grid of levels, level l having quantum q_l = 2^(50 l) and taking |v| < 2^(50 l + 49) = q_(l+1) / 2:
//     L      = max(XS_LMIN, max_i floor((ilogb(v_i) + 1) / 50))           the top level: a function of max |v_i| alone
//     k1_i   = RN(lsb1(v_i) / q_L),  r_i = lsb1(v_i) - k1_i q_L  (exact),  k2_i = RN(lsb1(r_i) / q_(L-1))
//     result = RN53(2^50 sum k1_i + sum k2_i) q_(L-1);  NaN / +-Inf summands: NaN if a NaN or both infinities occur, else +-Inf;
//     a finite summand of magnitude 2^999 or more (beyond the last level) counts as an infinity of its sign.
//   Lanes form k1_i and k2_i with four f64 additions: t = C1 + lsb1(v) rounds v to a multiple of q_L (C1 = 1.5 * 2^(50 L + 52) has
//   that ulp), q = t - C1 and r = lsb1(v) - q are exact, t2 = C2 + lsb1(r) likewise one level down; the BIT PATTERNS of t and t2
//   are added up as 64-bit integers (minus n times the constants' bits at the end).
//   A partial sum formed while the running top was T < L is brought to L exactly: one level up, its k1 total becomes the k2 total
//   (a summand taken at T is below q_(T+1) / 2, so its level-(T+1) multiple is 0 and its level-T multiple is what k2 would have
//   been) and its old k2 total is dropped; two or more levels up it contributes nothing (all its summands round to 0 there).
//   Worst-case quantum relative to max |v_i|: 2^-49; typical 2^-74.
// Partial sums travel as RECORDS of XS_WORDS doubles, every word an integer below 2^53 in magnitude: they survive any f64
// channel (an all-reduce(sum) over disjoint slots, a store, a file) bit for bit.  xs_merge adds two records, xs_round yields
// the Float64.
#pragma once

#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#define AMC_XS_HD __host__ __device__ inline

namespace amc {
namespace xs {

enum { XS_WORDS = 12 };                   // doubles per record
enum { XS_EMPTY = 0, XS_Q = 1, XS_R = 2, XS_PLAIN = 3 };
enum { XS_F_NAN = 1, XS_F_PINF = 2, XS_F_NINF = 4 };
enum { XS_W = 50, XS_B = 49 };            // bits per level; a level takes |v| < 2^(50 l + 49)
enum { XS_LMIN = -20, XS_LMAX = 19 };     // q_(LMIN-1) = 2^-1050 keeps 1.5 * 2^(q+52) a normal number; level 19 takes |v| < 2^999
enum { XS_LANE_CAP = 4096 };              // kind R: summands a lane adds into one pair of 64-bit integers between two flushes
enum { XS_GD_CAP_BITS = 5, XS_GD_LANE_CAP = 1 << XS_GD_CAP_BITS };   // kind Q (GradientData): 32 summands between two flushes
enum { XS_E_RATIO = -50 };                // accepted / total in [0, 1]: quantum 2^-50.  Lanes add the bit patterns of 6.0 + lsb1(ratio)
enum { XS_RATIO_LANE_CAP = 4096 };        // (6.0 = 1.5 * 2^2 has ulp 2^-50) minus those of 6.0 as 64-bit integers: 2^12 ratios fit

// ---- 128-bit two's complement integers (no __int128 on the device side of every toolchain this is compiled by) ----
struct i128 {
    uint64_t lo;
    int64_t hi;
};
AMC_XS_HD i128 i128_of(int64_t v) { return i128{(uint64_t)v, v < 0 ? (int64_t)-1 : (int64_t)0}; }
AMC_XS_HD i128 i128_add(i128 a, i128 b)
{
    i128 r;
    r.lo = a.lo + b.lo;
    r.hi = (int64_t)((uint64_t)a.hi + (uint64_t)b.hi + (r.lo < a.lo ? 1u : 0u));
    return r;
}
AMC_XS_HD i128 i128_neg(i128 a)
{
    i128 r;
    r.lo = ~a.lo + 1u;
    r.hi = (int64_t)(~(uint64_t)a.hi + (r.lo == 0 ? 1u : 0u));
    return r;
}
AMC_XS_HD i128 i128_shl(i128 a, int n)      // 0 < n < 64
{
    i128 r;
    r.hi = (int64_t)(((uint64_t)a.hi << n) | (a.lo >> (64 - n)));
    r.lo = a.lo << n;
    return r;
}
AMC_XS_HD bool i128_is_zero(i128 a) { return a.lo == 0 && a.hi == 0; }

// RN53(K) * 2^e: K rounded to 53 significant bits (ties to even), then scaled (ldexp rounds once more only where the
// result is subnormal).
AMC_XS_HD double i128_round_scaled(i128 K, int e)
{
    const bool negative = K.hi < 0;
    const i128 m = negative ? i128_neg(K) : K;              // |K| < 2^127
    uint64_t hi = (uint64_t)m.hi, lo = m.lo;
    if (hi == 0 && lo < (1ull << 53)) {
        const double v = __builtin_ldexp((double)lo, e);
        return negative ? -v : v;
    }
    // position of the leading bit
    const int p = hi ? 127 - __builtin_clzll(hi) : 63 - __builtin_clzll(lo);
    const int s = p - 52;                                    // bits to drop, >= 1
    // mant = |K| >> s (53 bits), rem = dropped bits compared with half
    uint64_t mant, half_bit, below;
    if (s >= 64) {
        mant = hi >> (s - 64);
        const int hs = s - 64;                               // bits of hi dropped
        half_bit = hs > 0 ? (hi >> (hs - 1)) & 1u : (lo >> 63) & 1u;
        const uint64_t hi_below = hs > 1 ? (hi & ((1ull << (hs - 1)) - 1)) : 0;
        const uint64_t lo_below = hs > 0 ? lo : (lo & 0x7FFFFFFFFFFFFFFFull);
        below = (hi_below | lo_below) ? 1u : 0u;
    } else {
        mant = (s == 0) ? lo : ((lo >> s) | (hi << (64 - s)));
        half_bit = (lo >> (s - 1)) & 1u;
        below = (s > 1 && (lo & ((1ull << (s - 1)) - 1))) ? 1u : 0u;
    }
    if (half_bit && (below || (mant & 1u))) mant += 1;       // may reach 2^53: exact in a double
    const double v = __builtin_ldexp((double)mant, e + s);
    return negative ? -v : v;
}

// ---- accumulator constants ----
AMC_XS_HD double xs_bits_double(uint64_t b)
{
    union { uint64_t u; double d; } c;
    c.u = b;
    return c.d;
}
AMC_XS_HD uint64_t xs_double_bits(double d)
{
    union { uint64_t u; double d; } c;
    c.d = d;
    return c.u;
}
// 1.5 * 2^(e + 52): the accumulator whose ulp is 2^e
AMC_XS_HD uint64_t xs_c_bits(int e) { return ((uint64_t)(e + 52 + 1023) << 52) | (1ull << 51); }
AMC_XS_HD double xs_c(int e) { return xs_bits_double(xs_c_bits(e)); }
// level l of a running-top column: the accumulator constant, and the bound 2^(50 l + 49) its summands stay below (2^999 for the
// last level: infinities, NaN -- which compares false -- and finite values of 2^999 or more lie beyond every level)
AMC_XS_HD uint64_t xs_level_c_bits(int l) { return xs_c_bits(XS_W * l); }
AMC_XS_HD double xs_level_cap(int l) { return xs_bits_double((uint64_t)(XS_W * l + XS_B + 1023) << 52); }
// the level a finite value needs: max(LMIN, floor((ilogb(v) + 1) / 50)), which exceeds LMAX for |v| >= 2^999; zero and subnormals: LMIN
AMC_XS_HD int xs_level_of_exponent(int be)                    // be: the biased exponent field, 0 .. 0x7FF
{
    if (be == 0) return XS_LMIN;
    const int l = (be - 1023 + 1 + 1050) / XS_W - 21;        // numerator >= 29: plain integer division is the floor
    return l < XS_LMIN ? XS_LMIN : l;
}
AMC_XS_HD int xs_level_of(double v) { return xs_level_of_exponent((int)((xs_double_bits(v) >> 52) & 0x7FFu)); }
// the smallest biased exponent that lies beyond the last level (|v| >= 2^999; infinities and NaN have 0x7FF)
enum { XS_BE_BEYOND = 999 + 1023 };
AMC_XS_HD double xs_lsb1(double v) { return xs_bits_double(xs_double_bits(v) | 1ull); }

// GradientData of the Gaussian displacement policy (gradients.jl:104-108 with particle_1d.jl:42-59): the quantum exponents
// of its four columns from sigma, 2^(es-1) <= sigma < 2^es (es = ilogb(sigma) + 1), and |z| < 8.5 (the Box-Muller radius of
// a 52-bit uniform: z^2 <= 72.1), alpha <= 1:
//   j = delta^2 alpha                 <= 72.1 sigma^2          < 2^(2 es + 7)
//   grad j = j d logq                 <= 72.1 * 71.1 sigma     < 2^(es + 13)
//   d logq / d sigma = (z^2 - 1)/sigma   |.| <= 71.1 / sigma   < 2^(8 - es)
//   g = (d logq)^2                    <= 5056 / sigma^2        < 2^(15 - 2 es)
// A lane adds XS_GD_LANE_CAP = 2^5 summands between two flushes, and a column's lane sum has to stay below 2^51 quanta:
// E = bound exponent + 5 - 51.  (Each summand is thus rounded to 2^-46 of its bound -- for a typical summand, 2^-29 .. 2^-39
// of itself; over the 1e7+ chains of a fold the rounding errors average out to far below the Float64 rounding of the total.)
struct GdExponents { int e[4]; };
AMC_XS_HD int xs_gd_es(double sigma) { return (int)((xs_double_bits(sigma) >> 52) & 0x7FFu) - 1023 + 1; }
AMC_XS_HD GdExponents xs_gd_exponents_es(int es)
{
    const int drop = 51 - XS_GD_CAP_BITS;
    GdExponents g;
    g.e[0] = 2 * es + 7 - drop;
    g.e[1] = es + 13 - drop;
    g.e[2] = 8 - es - drop;
    g.e[3] = 15 - 2 * es - drop;
    return g;
}
// column i's exponent by itself (an index that is not a constant would put the four of them in private memory)
AMC_XS_HD int xs_gd_exponent_of(double sigma, int i)
{
    const int es = xs_gd_es(sigma), drop = 51 - XS_GD_CAP_BITS;
    return i == 0 ? 2 * es + 7 - drop : i == 1 ? es + 13 - drop : i == 2 ? 8 - es - drop : 15 - 2 * es - drop;
}
AMC_XS_HD GdExponents xs_gd_exponents(double sigma)
{
    const int es = xs_gd_es(sigma);
    const int drop = 51 - XS_GD_CAP_BITS;
    GdExponents g;
    g.e[0] = 2 * es + 7 - drop;
    g.e[1] = es + 13 - drop;
    g.e[2] = 8 - es - drop;
    g.e[3] = 15 - 2 * es - drop;
    return g;
}

// ---- partial sums in integer form ----
struct PartQ {            // kind Q
    i128 k;
    uint32_t flags;
};
struct PartR {            // kind R
    int32_t top;
    uint32_t flags;
    i128 k1, k2;
};
AMC_XS_HD PartR part_r_empty() { return PartR{XS_LMIN, 0u, i128{0, 0}, i128{0, 0}}; }
// bring a partial to a higher top (exact, see the header comment)
AMC_XS_HD void part_r_raise(PartR& a, int top)
{
    const int d = top - a.top;
    if (d <= 0) return;
    a.k2 = (d == 1) ? a.k1 : i128{0, 0};
    a.k1 = i128{0, 0};
    a.top = top;
}
AMC_XS_HD void part_r_merge(PartR& a, PartR b)
{
    if (b.top > a.top) part_r_raise(a, b.top);
    else part_r_raise(b, a.top);
    a.k1 = i128_add(a.k1, b.k1);
    a.k2 = i128_add(a.k2, b.k2);
    a.flags |= b.flags;
}
AMC_XS_HD double flags_value(uint32_t flags)
{
    if ((flags & XS_F_NAN) || ((flags & XS_F_PINF) && (flags & XS_F_NINF))) return __builtin_nan("");
    return (flags & XS_F_PINF) ? __builtin_huge_val() : -__builtin_huge_val();
}
AMC_XS_HD double part_r_round(const PartR& a)
{
    if (a.flags) return flags_value(a.flags);
    return i128_round_scaled(i128_add(i128_shl(a.k1, XS_W), a.k2), XS_W * (a.top - 1));
}
AMC_XS_HD double part_q_round(const PartQ& a, int e)
{
    if (a.flags) return flags_value(a.flags);
    return i128_round_scaled(a.k, e);
}

// ---- records: XS_WORDS doubles, each an integer below 2^53 in magnitude ----
//   [0] kind  [1] E (kind Q) or top level (kind R)  [2] flags  [3..6] k1 / k as 32-bit limbs, least significant first, the
//   last one signed  [7..10] k2 likewise  [11] the value of a PLAIN record (counts: integers, exact under +)
AMC_XS_HD void limbs_store(double* w, i128 k)
{
    w[0] = (double)(uint32_t)k.lo;
    w[1] = (double)(uint32_t)(k.lo >> 32);
    w[2] = (double)(uint32_t)(uint64_t)k.hi;
    w[3] = (double)(int32_t)(k.hi >> 32);
}
// limbs that may have been ADDED word by word (each then below 2^53 in magnitude): carries are propagated here
AMC_XS_HD i128 limbs_load(const double* w)
{
    i128 r = i128{0, 0};
    for (int i = 3; i >= 0; --i) {
        // r = r * 2^32 + w[i]
        i128 s;
        s.hi = (int64_t)(((uint64_t)r.hi << 32) | (r.lo >> 32));
        s.lo = r.lo << 32;
        r = i128_add(s, i128_of((int64_t)w[i]));
    }
    return r;
}
AMC_XS_HD void rec_clear(double* rec) { for (int i = 0; i < XS_WORDS; ++i) rec[i] = 0.0; }
AMC_XS_HD void rec_from_q(double* rec, const PartQ& p, int e)
{
    rec_clear(rec);
    rec[0] = (double)XS_Q; rec[1] = (double)e; rec[2] = (double)p.flags;
    if (p.flags) return;                      // NaN: the integer means nothing, canonically zero
    limbs_store(rec + 3, p.k);
}
AMC_XS_HD void rec_from_r(double* rec, const PartR& p)
{
    rec_clear(rec);
    rec[0] = (double)XS_R; rec[1] = (double)p.top; rec[2] = (double)p.flags;
    if (p.flags) return;                      // NaN / infinite: the integers mean nothing, canonically zero
    limbs_store(rec + 3, p.k1);
    limbs_store(rec + 7, p.k2);
}
AMC_XS_HD void rec_from_plain(double* rec, double v)
{
    rec_clear(rec);
    rec[0] = (double)XS_PLAIN; rec[11] = v;
}
AMC_XS_HD PartR rec_to_r(const double* rec)
{
    PartR p;
    p.top = (int32_t)rec[1]; p.flags = (uint32_t)rec[2];
    p.k1 = limbs_load(rec + 3); p.k2 = limbs_load(rec + 7);
    return p;
}
AMC_XS_HD PartQ rec_to_q(const double* rec) { return PartQ{limbs_load(rec + 3), (uint32_t)rec[2]}; }

// into += from.  Records of different kinds (or kind Q records of different quanta) do not add: the result is NaN.
AMC_XS_HD void rec_merge(double* into, const double* from)
{
    const int kf = (int)from[0], ki = (int)into[0];
    if (kf == XS_EMPTY) return;
    if (ki == XS_EMPTY) { for (int i = 0; i < XS_WORDS; ++i) into[i] = from[i]; return; }
    if (ki != kf || (ki == XS_Q && into[1] != from[1])) {
        into[2] = (double)((uint32_t)into[2] | XS_F_NAN);
        return;
    }
    if (ki == XS_PLAIN) { into[11] += from[11]; return; }
    if (ki == XS_Q) {
        PartQ a = rec_to_q(into);
        const PartQ b = rec_to_q(from);
        a.k = i128_add(a.k, b.k);
        a.flags |= b.flags;
        rec_from_q(into, a, (int)into[1]);
        return;
    }
    PartR a = rec_to_r(into);
    part_r_merge(a, rec_to_r(from));
    rec_from_r(into, a);
}
AMC_XS_HD double rec_round(const double* rec)
{
    switch ((int)rec[0]) {
    case XS_Q: return part_q_round(rec_to_q(rec), (int)rec[1]);
    case XS_R: return part_r_round(rec_to_r(rec));
    case XS_PLAIN: return rec[11];
    default: return 0.0;
    }
}

}  // namespace xs
}  // namespace amc
