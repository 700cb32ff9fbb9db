// amc_aux_kernels.h -- histogram / energy / Float32 conversion / strided gather kernels and the parity-test hooks.
// Part of the kernel sources of the many-chain Metropolis engine (gfx950 / CDNA4); amc_kernels.h includes all of them, in order.
#pragma once

#include "amc_estimator.h"

namespace amc {

// Device-side replacement for the per-chain text trajectories (StoreTrajectories, src/algorithms.jl:154-210):
// histogram of the chain positions over half-open bins [lo + i w, lo + (i+1) w), i < n_bins, with
// bin = floor((x - lo) * inv_w) in this exact f64 form; counts[n_bins..n_bins+2] = below lo, >= hi, NaN.
// Per-block LDS histogram (u32 LDS atomics), flushed with one u64 global atomic per non-empty bin.
#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void histogram_kernel(const double* x, int64_t n_chains, double lo, double hi,
                                                               double inv_w, int n_bins, unsigned long long* counts)
{
    extern __shared__ unsigned int s_hist[];
    for (int i = threadIdx.x; i < n_bins + 3; i += AMC_BLOCK) s_hist[i] = 0u;
    __syncthreads();
    auto count = [&](double v) {
        int b;
        if (v != v) b = n_bins + 2;
        else if (v < lo) b = n_bins;
        else if (v >= hi) b = n_bins + 1;
        else {
            b = (int)((v - lo) * inv_w);
            b = b < n_bins ? b : n_bins - 1;       // (hi - ulp - lo) * inv_w can round up to n_bins
        }
        atomicAdd(&s_hist[b], 1u);
    };
    // four positions per lane and trip, both 16-byte loads issued before the first is used (one 8-byte load per trip left the
    // pass waiting for latency: ~50 us for 80 MB)
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    const int64_t n_quads = n_chains >> 2;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    for (int64_t q = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; q < n_quads; q += stride) {
        const double2 a = x2[2 * q], b = x2[2 * q + 1];
        count(a.x); count(a.y); count(b.x); count(b.y);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_chains & 3)) count(x[4 * n_quads + threadIdx.x]);
    __syncthreads();
    for (int i = threadIdx.x; i < n_bins + 3; i += AMC_BLOCK)
        if (s_hist[i]) atomicAdd(&counts[i], (unsigned long long)s_hist[i]);
}
#endif

// e[c] = potential(x[c]) (Particle.e, particle_1d.jl:13-15,33) for amc_download_state when the host cannot
// evaluate the potential itself (POT_CUSTOM).
template <int POT>
__global__ __launch_bounds__(AMC_BLOCK) void energy_kernel(const real_t* x, int64_t n_chains, double* e)
{
    __shared__ double s_math[TAB_DOUBLES];
    stage_math_tables(s_math, threadIdx.x, AMC_BLOCK);
    const int64_t stride = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t c = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; c < n_chains; c += stride)
        e[c] = (double)potential<POT>(x[c], s_math);
}

// Float32 state (AMC_STATE_F32 builds only): the C ABI moves positions as doubles whatever the state type, so uploads
// are narrowed (T(x), round to nearest even -- what Particle(Float32(x), ...) does) and downloads widened (exact).
// The kernels that only READ positions for host-side consumers (histogram, strided snapshots) run on the widened copy.
#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void narrow_state_kernel(const double* in, int64_t n, real_t* out)
{
    const int64_t gs = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; i < n; i += gs) out[i] = (real_t)in[i];
}
#endif

#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void widen_state_kernel(const real_t* in, int64_t n, double* out)
{
    const int64_t gs = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; i < n; i += gs) out[i] = (double)in[i];
}
#endif

// Strided snapshot: out[i] = x[first + i*stride] (binary stand-in for a subset of trajectory files).
#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(AMC_BLOCK) void gather_strided_kernel(const double* x, int64_t first, int64_t stride,
                                                                    int64_t count, double* out)
{
    const int64_t gs = (int64_t)gridDim.x * AMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * AMC_BLOCK + threadIdx.x; i < count; i += gs) out[i] = x[first + i * stride];
}
#endif

// Parity-test hooks (amc_selftest_*): the arithmetic-spec primitives, one value per thread.
#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ void selftest_math_kernel(int fn, const double* a, const double* b, double* out, int64_t n)
{
    __shared__ double s_math[TAB_DOUBLES];
    stage_math_tables(s_math, threadIdx.x, blockDim.x);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = a[i];
    double s, c, r = 0.0;
    switch (fn) {
    case 0: r = exp_f64(v, s_math); break;
    case 1: r = log_f64(v); break;
    case 2: sincospi_f64(v, s, c, s_math); r = s; break;
    case 3: sincospi_f64(v, s, c, s_math); r = c; break;
    case 4: r = __builtin_sqrt(v); break;
    case 5: r = v / b[i]; break;
    case 6: r = div_by_const(v, b[i], 1.0 / b[i]); break;
    case 7: r = logbm_f64(v, s_math); break;
    case 8: r = sqrt_radius_f64(v); break;
    case 9:
    case 10:
    case 11: {
        // log_proposal_density(delta = a, sigma = b) / its sigma-derivative, through the code the estimator runs:
        // prepare_params for a one-move pool, then log_proposal_density_withgrad
        double tab[PT_ROWS * AMC_MAX_MOVES];
        tab[PT_SIGMA * AMC_MAX_MOVES] = b[i];
        tab[PT_WEIGHT * AMC_MAX_MOVES] = 1.0;
        prepare_params(tab, 1);
        const LogQ lq = log_proposal_density_withgrad((real_t)v, tab[PT_DEN * AMC_MAX_MOVES], tab[PT_RDEN * AMC_MAX_MOVES],
                                                      tab[PT_LOGC * AMC_MAX_MOVES], tab[PT_DDEN * AMC_MAX_MOVES],
                                                      tab[PT_DLHALF * AMC_MAX_MOVES]);
        // 9: the reference-ordered log density; 10: d logq / d sigma in the reference's order (ForwardDiff's dual rules:
        // what withgrad_log_proposal_density! returns); 11: d logq / d sigma as the estimator kernel forms it (pg_sample)
        const double d2 = (double)((real_t)v * (real_t)v);
        const double dq = __builtin_fma(d2, tab[PT_C3HI * AMC_MAX_MOVES], __builtin_fma(d2, tab[PT_C3LO * AMC_MAX_MOVES], -tab[PT_DLHALF * AMC_MAX_MOVES]));
        r = fn == 9 ? lq.logq : (fn == 10 ? lq.dlogq : dq);
        break;
    }
    default: break;
    }
    out[i] = r;
}
#endif

// Exhaustive check of the accept filter's float estimate (accept_filter): for EVERY float t with bit pattern in
// [bits_lo, bits_hi] the relative deviation of v_exp_f32(max(t, -17) * log2e) from the spec's f64 exp(t); the maximum
// over the range lands in out_max_bits (bits of a non-negative double compare like integers).
#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(256) void selftest_filter_kernel(uint32_t bits_lo, uint64_t count, unsigned long long* out_max_bits)
{
    __shared__ double s_math[TAB_DOUBLES];
    stage_math_tables(s_math, threadIdx.x, 256);
    double worst = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const float t = __uint_as_float(bits_lo + (uint32_t)i);
        const float ex = __builtin_amdgcn_exp2f(__builtin_fmaxf(t, -17.0f) * 0x1.715476p+0f);
        const double ref = exp_f64((double)__builtin_fmaxf(t, -17.0f), s_math);
        const double rel = __builtin_fabs((double)ex - ref) / ref;
        worst = (rel > worst) ? rel : worst;                  // NaN never enters (ref is finite and positive here)
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(worst, off, 64);
        worst = (o > worst) ? o : worst;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_max_bits, (unsigned long long)__double_as_longlong(worst));
}
#endif

// The wave-total primitives (wave_total_i64 by folding, wave_max_u32) on one wave's worth of arbitrary lane values: in is
// [6][64] 64-bit integers; out[0..5] the six totals through wave_total_i64<6>, out[6..7] two of them through <2>, out[8..10]
// three through <3>, out[11] one through <1>, out[12] wave_max_u32 of the low words of row 0; ref[0..5] the totals by the plain
// DPP form of round 4.  The host compares both with its own sums.
#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ __launch_bounds__(64) void selftest_wave_totals_kernel(const long long* in, long long* out, long long* ref)
{
    const int lane = threadIdx.x & 63;
    long long v[6], r[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = r[i] = in[i * 64 + lane];
    long long two[2] = {v[4], v[1]}, three[3] = {v[5], v[0], v[2]}, one[1] = {v[3]};
    const uint32_t m = wave_max_u32((uint32_t)(unsigned long long)v[0]);
    wave_total_i64<6>(v);
    wave_total_i64<2>(two);
    wave_total_i64<3>(three);
    wave_total_i64<1>(one);
    wave_total_i64_dpp<6>(r);
    if (lane == 17) {        // any lane: the totals are wave-uniform
#pragma unroll
        for (int i = 0; i < 6; ++i) { out[i] = v[i]; ref[i] = r[i]; }
        out[6] = two[0]; out[7] = two[1];
        out[8] = three[0]; out[9] = three[1]; out[10] = three[2];
        out[11] = one[0];
        out[12] = (long long)m;
    }
}
#endif

#if AMC_PLAIN_KERNELS
AMC_KERNEL_LINKAGE __global__ void selftest_philox_kernel(uint32_t key0, uint32_t key1, const uint64_t* pair, const uint64_t* t,
                                       uint32_t draw, uint32_t stream, uint32_t* out4, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 v = philox4x32_10(draw_counter(pair[i], t[i], draw, stream), key0, key1);
    out4[4 * i + 0] = v.x; out4[4 * i + 1] = v.y; out4[4 * i + 2] = v.z; out4[4 * i + 3] = v.w;
}
#endif
}  // namespace amc
