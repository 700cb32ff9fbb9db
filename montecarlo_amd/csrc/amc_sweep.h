// amc_sweep.h -- K1, the sweep (make_step!(::Metropolis)): launch arguments, the draws of a step, the move pick, pair_steps, sweep_kernel,
// and the fold of the step log into the per-chain counters (fold_log_kernel).
// Part of the kernel sources of the many-chain Metropolis engine (gfx950 / CDNA4); amc_kernels.h includes all of them, in order.
#pragma once

#include "amc_wave_sums.h"

namespace amc {

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
                                           const UserTheta& th1 = UserTheta{0.0, 0.0, 0.0, 0.0, 0.0, 0.0})
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
//         callback_acceptance needs no second pass over x.  RED_FORM_COLS: the sums SweepArgs.red_cols names; RED_FORM_E: sum e
//         alone, compiled in (RedCols) -- what the host launches when nothing else is asked for
template <int POT, bool MULTI, int LOG, bool BETA, bool SINGLE, int REDUCE = RED_FORM_NONE>
__global__ __launch_bounds__(AMC_BLOCK) void sweep_kernel(const SweepArgs a)
{
    // REDUCE with LOG (per-chain counters): rows carry the sums over x only; the acceptance ratios of the same
    // callback come from the fold of the step log that follows (fold_log_kernel<KS, true>)
    static_assert(!MULTI || LOG, "K > 1 always keeps per-chain counters");
    // the callback sums: reproducible (amc_xsum.h); the count of full trips lives on the scalar unit
    constexpr bool RED_E = REDUCE == RED_FORM_E;
    constexpr int RNC = RedCols<POT, RED_E>::NC;
    RLanes<RNC> red;
    __shared__ xs::PartR s_red[REDUCE ? AMC_BLOCK / 64 : 1][RNC];
    if (REDUCE) r_init(red, s_red[threadIdx.x >> 6]);
    __shared__ double s_tab[MULTI ? (5 + AMC_SIGMA_MEMO) * AMC_MAX_MOVES : 1];
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
            if (AMC_SIGMA_MEMO) s_tab[(AMC_SIGMA_MEMO ? 5 : 0) * AMC_MAX_MOVES + i] = log_f64(a.ptab[PT_SIGMA * AMC_MAX_MOVES + i]);      // (SigmaArg)
        }
        // visible to the block after the barrier that ends stage_math_tables below
    }
    // K == 1: wave-uniform scalars (s_load)
    const double sigma1 = a.ptab[PT_SIGMA * AMC_MAX_MOVES];
    const double den1 = a.ptab[PT_DEN * AMC_MAX_MOVES];
    const double logc1 = a.ptab[PT_LOGC * AMC_MAX_MOVES];
    const double rden1 = a.ptab[PT_RDEN * AMC_MAX_MOVES];
    UserTheta th1 = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
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
    stage_user_theta(a.ptab, MULTI);
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
            red_add_pair<POT, RED_E>(red, xv, true, true, s_math, s_red[threadIdx.x >> 6], a.red_cols);
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
        if (REDUCE) red_add_pair<POT, RED_E>(red, xv, v0, v1, s_math, s_red[threadIdx.x >> 6], a.red_cols);
    }
    if (REDUCE)
        red_finish<POT, RED_E>(red, s_red, a.red_partials + (int64_t)blockIdx.x * a.red_stride, a.red_stride == RED_COMPACT_WORDS, a.red_cols);
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
}  // namespace amc
