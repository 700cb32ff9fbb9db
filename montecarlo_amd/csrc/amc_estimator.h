// amc_estimator.h -- K3, make_step!(::PolicyGradientEstimator) -- optionally fused with one make_step!(::Metropolis) and the callback sums --
// with the in-kernel fold, gradients_data +=, and learning_step! in its tail.
// Part of the kernel sources of the many-chain Metropolis engine (gfx950 / CDNA4); amc_kernels.h includes all of them, in order.
#pragma once
// script-defined forms (run-time compiled): the waves per SIMD the register allocator is held to (amdgpu_waves_per_eu); the run-time
// compiler sets it (amc_rtc.hip; AMC_RTC_WAVES in the environment is the A/B knob), 1 = no constraint
#ifndef AMC_RTC_WAVES
#define AMC_RTC_WAVES 1
#endif

#include "amc_pg_tail.h"

namespace amc {

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
template <int POT, int NL, bool BETA, int SWEEP = 0, int REDUCE = RED_FORM_NONE, bool MIDFLUSH = false>
// (amdgpu_waves_per_eu: the fused time step that also leaves sum e -- RED_FORM_E, built-in potentials -- needs 97 VGPRs, one more than five
// waves per SIMD allow; held to 96 it spills ONE register outside the loop and the launch runs on five resident blocks per CU instead
// of four: 69.9 -> 67.2 us at 1e7 chains, same-box A/B.  The same squeeze on the K = 2 sweep with sums, 81 -> 80 VGPRs for six waves,
// costs a microsecond instead: profiles/r05_reduce_occupancy_ab.txt.)
__global__ __launch_bounds__(AMC_BLOCK) __attribute__((amdgpu_waves_per_eu((REDUCE == RED_FORM_E && SWEEP != 0 && PgKind<POT>::Q) ? 5 : (PgKind<POT>::Q ? 1 : AMC_RTC_WAVES))))
void pg_estimate_kernel(const PgArgs a, const SweepArgs sw)
{
    static_assert(!REDUCE || SWEEP != 0, "the callback sums ride on the fused time step");
    constexpr bool QK = PgKind<POT>::Q;
    constexpr int ROW = PgKind<POT>::ROW;
    constexpr int NC = AMC_PG_NC;               // GradientData columns per learnable move (4 for one parameter)
    static_assert(!QK || NC == 4, "the quanta of xs_gd_exponents are those of the one-parameter Gaussian policy");
    constexpr int NV = NL * NC;
    static_assert(NV <= 32, "the tail's wave 0 owns a row's columns");
    // the callback sums: reproducible (amc_xsum.h); the count of full trips lives on the scalar unit
    constexpr bool RED_E = REDUCE == RED_FORM_E;
    constexpr int RNC = RedCols<POT, RED_E>::NC;
    RLanes<RNC> red;
    __shared__ xs::PartR s_red[REDUCE ? AMC_BLOCK / 64 : 1][RNC];
    if (REDUCE) r_init(red, s_red[threadIdx.x >> 6]);
    // the pool-wide accepted total this block can see before the launch (see sweep_kernel)
    unsigned long long slots_before = 0;
    if (REDUCE && SWEEP == 3 && threadIdx.x == 0)
        for (int sl = (int)blockIdx.x; sl < sw.n_slots; sl += (int)gridDim.x) slots_before += sw.acc_total[sl];
    __shared__ double s_math[TAB_DOUBLES];
    __shared__ double s_tab[SWEEP == 2 ? (5 + AMC_SIGMA_MEMO) * AMC_MAX_MOVES : 1];
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
            if (AMC_SIGMA_MEMO) s_tab[(AMC_SIGMA_MEMO ? 5 : 0) * AMC_MAX_MOVES + i] = log_f64(sw.ptab[PT_SIGMA * AMC_MAX_MOVES + i]);      // (SigmaArg)
        }
    }
    double sw_sigma1 = SWEEP ? sw.ptab[PT_SIGMA * AMC_MAX_MOVES] : 0.0;
    double sw_den1 = SWEEP ? sw.ptab[PT_DEN * AMC_MAX_MOVES] : 0.0;
    double sw_logc1 = SWEEP ? sw.ptab[PT_LOGC * AMC_MAX_MOVES] : 0.0;
    double sw_rden1 = SWEEP ? sw.ptab[PT_RDEN * AMC_MAX_MOVES] : 0.0;
    // script-defined policies: the further parameters of the sweep's only move (K == 1) and of the learnable moves, wave-uniform
    UserTheta sw_th1 = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    UserTheta c_th[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) c_th[l] = UserTheta{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#ifdef AMC_USER_LOGQ
    if (SWEEP == 1 || SWEEP == 3) sw_th1 = user_theta_uniform(sw.ptab, 0);
#pragma unroll
    for (int l = 0; l < NL; ++l)
        if (l < a.n_learn) c_th[l] = user_theta_uniform(a.ptab, a.learn_ids[l]);
#endif
    // a learning step the previous launch left pending (pg_apply_pending): wave-uniform
    constexpr bool CAN_DEFER = QK && NL <= 2 && AMC_NP == 1;
    // the pending branch below rewrites the learnable moves' rows 0, 1, 2, 4 of s_tab from sigma', NOT row 5 (the memoised
    // log(sigma_k) of script-defined proposals): forms with the memo are kind R and never defer -- keep it that way, or add the row
    static_assert(!(CAN_DEFER && AMC_SIGMA_MEMO), "a deferred learning step would leave the memoised log(sigma) one step behind");
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
                    // (l_base: the launch's first move in the estimator CALL -- a call may be taken in several launches, the draw is
                    // named by the move's index in the call)
                    const uint32_t sample_id = (uint32_t)((a.l_base + l) * a.q_batch + q);
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
                    // kind R: ONE summand per lane, sample and column -- the chain PAIR's sum fl(s_even + s_odd), as the callback
                    // sums take theirs (round 6: half the deposits, -7 .. -10 % per script-defined time step)
                    // ... and none for a column nobody will read: in a launch whose tail takes the learning step (tail modes 3 / groups:
                    // gradients_data is consumed and reset there), VPG reads j and grad j alone, BLPG / BLAPG grad logq too, NPG / ANPG
                    // the metric g -- the host names, per learnable move, the column groups the move's optimiser leaves unread
                    // (pg_skip_of: wave-uniform, from the launch's arguments)
                    const int skip = pg_skip_of(a.tail_mode, l);
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        if (i > AMC_NP && i <= 2 * AMC_NP && (skip & 1)) continue;          // grad logq_forward [P]
                        if (i > 2 * AMC_NP && (skip & 2)) continue;                         // g, upper triangle
                        r_deposit(gr, QK ? 0 : l * NC + i, (v0 ? s0[i] : 0.0) + s1[i], s_gr[threadIdx.x >> 6]);
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
                        for (int i = 0; i < 4; ++i)           // (the pair's sum: one deposit, see above)
                            r_deposit(gr, QK ? 0 : l * 4 + i, (v0 ? s0[i] : 0.0) + s1[i], s_gr[threadIdx.x >> 6]);
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
    stage_user_theta(a.ptab, SWEEP == 2);
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
        if (REDUCE) red_add_pair<POT, RED_E>(red, xv, true, true, s_math, s_red[threadIdx.x >> 6], sw.red_cols);
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
        if (REDUCE) red_add_pair<POT, RED_E>(red, xv, v0, v1, s_math, s_red[threadIdx.x >> 6], sw.red_cols);
    }
    if (REDUCE)
        red_finish<POT, RED_E>(red, s_red, sw.red_partials + (int64_t)blockIdx.x * sw.red_stride, sw.red_stride == RED_COMPACT_WORDS, sw.red_cols);
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
            pg_tail_np(s_tot, AMC_NP, tl->learn_ids[a.l_base], tl->n_samples, tail_mode >= 3, tl->opt.kind[a.l_base], tl->opt.h0[a.l_base],
                       tl->opt.h1[a.l_base], tl->ptab_rw, tl->gd_acc, tl->status);      // (the record describes the estimator call; l_base: this launch's move)
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
}  // namespace amc
