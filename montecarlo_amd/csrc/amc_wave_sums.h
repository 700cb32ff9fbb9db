// amc_wave_sums.h -- the device side of the reproducible cross-chain sums (amc_xsum.h): integer wave totals through DPP, the lanes' kind-Q /
// kind-R accumulators and their flushes, block rows, the callback columns a REDUCE launch forms (red_add_pair / red_finish).
// Part of the kernel sources of the many-chain Metropolis engine (gfx950 / CDNA4); amc_kernels.h includes all of them, in order.
#pragma once

#include "amc_model.h"

namespace amc {

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
// EONLY (round 5): the form of a launch whose callbacks read sum e alone (callback_energy: BASELINE configs 2 - 5; RED_FORM_E) -- one lane
// column, compiled in: the launch keeps 81 instead of 89 VGPRs and loses the other columns' branches (K = 2 sweep with sums 40.3 ->
// 38.2 us, fused PGMC step with sums 71.6 -> 70.2, same-box A/B).  Its rows have the same layout; the columns nobody asked for repeat column 0
// and the host, which knows what was asked for, leaves their records empty.
enum { RED_FORM_NONE = 0, RED_FORM_COLS = 1, RED_FORM_E = 2 };        // the REDUCE template argument of sweep_kernel / pg_estimate_kernel
template <int POT, bool EONLY = false>
struct RedCols {
    static constexpr bool X2_IS_E = POT == POT_HARMONIC && sizeof(real_t) == 8;
    static constexpr int NC = EONLY ? 1 : X2_IS_E ? 2 : 3;
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
template <int POT, bool EONLY = false>
__device__ __forceinline__ void red_add_pair(RLanes<RedCols<POT, EONLY>::NC>& L, real2 xv, bool v0, bool v1, const double* s_math,
                                             xs::PartR* slot, int cols)
{
    const double x0 = v0 ? (double)xv.x : 0.0, x1 = v1 ? (double)xv.y : 0.0;
    if (cols & (RedCols<POT>::X2_IS_E ? (RED_WANT_E | RED_WANT_XX) : RED_WANT_E)) {
        const double e0 = v0 ? (double)potential<POT>(xv.x, s_math) : 0.0, e1 = v1 ? (double)potential<POT>(xv.y, s_math) : 0.0;
        r_deposit(L, 0, e0 + e1, slot);
    }
    if (EONLY) return;                  // (the host picks this form only when nothing else is asked for)
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
template <int POT, bool EONLY = false>
__device__ __forceinline__ void red_finish(RLanes<RedCols<POT, EONLY>::NC>& L, xs::PartR (*slots)[RedCols<POT, EONLY>::NC], xs_word* row, bool compact,
                                           int cols)
{
    constexpr int NC = RedCols<POT, EONLY>::NC;
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
        if (EONLY || (cols & ~(RedCols<POT>::X2_IS_E ? (RED_WANT_E | RED_WANT_XX) : RED_WANT_E)) == 0) {
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
            auto lane_col = [](int rc) { return (EONLY || (RedCols<POT>::X2_IS_E && rc == 2)) ? 0 : rc; };
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
        const int c = (EONLY || (RedCols<POT>::X2_IS_E && threadIdx.x == 2)) ? 0 : (int)threadIdx.x;
        xs_store_r_row(s_row + threadIdx.x * XS_ROW_R, r_block_total<NC>(slots, c));
        if (threadIdx.x == 0) s_row[RED_ROW_COUNT] = 0ull;          // (unused: the host knows the chains a launch covers)
    }
    __syncthreads();
    if (threadIdx.x <= RED_ROW_COUNT) row[threadIdx.x] = s_row[threadIdx.x];
}
}  // namespace amc
