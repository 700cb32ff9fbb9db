// amc_reduce_pass.h -- K2a, the callback reductions as a pass of their own, and the passes over the per-chain counters (rebase, totals).
// Part of the kernel sources of the many-chain Metropolis engine (gfx950 / CDNA4); amc_kernels.h includes all of them, in order.
#pragma once

#include "amc_params.h"

namespace amc {

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
}  // namespace amc
