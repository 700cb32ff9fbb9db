// amc_api.hip -- C ABI (include/amc.h) over the HIP kernels of amc_kernels.h.
//
// Host side of the engine: owns device memory, the stream and the step counter;
// validates arguments the way the reference's constructors assert them
// (src/metropolis.jl:248-251, Distributions.Categorical's probability-vector check).
// No CPU fallback: every entry point either runs on the GPU or returns an error.
#include "amc_internal.h"

static thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int grid_for(const amc_handle* h, int64_t n_items, int blocks_per_cu)
{
    // memory-streaming shape: <= 8 blocks of 256 per CU, grid-stride the rest
    int64_t blocks = (n_items + AMC_BLOCK - 1) / AMC_BLOCK;
    const int64_t cap = (int64_t)h->n_cu * (blocks_per_cu > 0 ? blocks_per_cu : h->blocks_per_cu);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

static int push_params(amc_handle* h, const double* sigma, const double* weight)
{
    { const int rc = pg_resolve(h); if (rc != AMC_OK) return rc; }
    std::vector<double> tab((size_t)amc::PT_ROWS * AMC_MAX_MOVES, 0.0);
    AMC_HIP(hipMemcpyAsync(tab.data(), h->d_ptab, tab.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));
    for (int k = 0; k < h->K; ++k) {
        if (sigma) tab[amc::PT_SIGMA * AMC_MAX_MOVES + k] = sigma[k];
        if (weight) tab[amc::PT_WEIGHT * AMC_MAX_MOVES + k] = weight[k];
    }
    AMC_HIP(hipMemcpyAsync(h->d_ptab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));   // tab is a stack-scoped host buffer
    hipLaunchKernelGGL(amc::prepare_params_kernel, dim3(1), dim3(64), 0, h->stream, h->d_ptab, h->K);
    AMC_HIP(hipGetLastError());
    if (weight && h->K > 1) {      // the cumulative weights changed: rebuild the 12-bit move-pick table from them
        hipLaunchKernelGGL(amc::prepare_pick_kernel, dim3(AMC_PICK_CELLS / AMC_BLOCK), dim3(AMC_BLOCK), 0, h->stream, h->d_ptab, h->K,
                           h->d_pick);
        AMC_HIP(hipGetLastError());
    }
    return AMC_OK;
}

// The step log's form (store_log_pair): none without per-chain counters, two chains per byte while the move index fits three
// bits, one byte per chain beyond.
int log_form(const amc_handle* h)
{
    return !h->counters ? AMC_LOG_NONE : (h->K <= AMC_PACKED_LOG_MOVES ? AMC_LOG_PACKED : AMC_LOG_BYTES);
}

template <int POT, bool SINGLE>
int launch_sweep_s(amc_handle* h, const amc::SweepArgs& a, int grid)
{
    const bool multi = h->K > 1;
#define AMC_SWEEP(MULTI, LOG)                                                                                              \
    do {                                                                                                                   \
        if (h->beta_arr)                                                                                                   \
            hipLaunchKernelGGL((amc::sweep_kernel<POT, MULTI, LOG, true, SINGLE>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, a); \
        else                                                                                                               \
            hipLaunchKernelGGL((amc::sweep_kernel<POT, MULTI, LOG, false, SINGLE>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, a); \
    } while (0)
    // K > 1 always keeps per-chain counters (callback_acceptance is a mean of per-chain ratios); the step log's form is
    // part of the instantiation (log_form)
    if (multi && log_form(h) == AMC_LOG_PACKED) AMC_SWEEP(true, AMC_LOG_PACKED);
    else if (multi) AMC_SWEEP(true, AMC_LOG_BYTES);
    else if (h->counters) AMC_SWEEP(false, AMC_LOG_PACKED);
    else AMC_SWEEP(false, AMC_LOG_NONE);
#undef AMC_SWEEP
    AMC_HIP(hipGetLastError());
    return AMC_OK;
}

// Adds the pending rows of the step log into the per-chain counters (on the stream).  Everything that reads or
// replaces d_acc / d_tot calls this first.
// with_ratio (K <= 4): the launch also leaves callback_acceptance's per-move sums as block partials in h_ratio
// (rows = its grid; *ratio_rows receives the count) and runs even when no log row is pending.
// (Round 3 measured the callback's fold on a second stream beside the sweeps queued after it, the log a ring of rows:
// no gain -- config 3: 37.3 against 37.6 us per time step with the callback read a period late, 41.0 against 38.9 read at
// once; config 5: 72.7 against 70.7 either way.  The fold's waves do not fit beside five 96-register waves of the fused
// kernel, so they take whole wave slots from it, and the cross-stream events cost more than the overlap returns.)
static int fold_log(amc_handle* h, bool with_ratio = false, int* ratio_rows = nullptr, amc::xs_word* ratio_dst = nullptr)
{
    if (!h->d_log || (h->log_fill == 0 && !with_ratio)) return AMC_OK;
    // tiles of AMC_FOLD_TILE chains, dealt evenly: every block takes the same number of tiles (a grid of 2048 over 2442
    // tiles would leave 80 % of the blocks idle for the second half of the launch)
    const int64_t n_tiles = (h->M + AMC_FOLD_TILE - 1) / AMC_FOLD_TILE;
    const int64_t cap = (int64_t)h->n_cu * h->blocks_per_cu;
    const int64_t rounds = (n_tiles + cap - 1) / cap;
    const int grid = (int)((n_tiles + rounds - 1) / rounds);
    uint16_t* const no_hi = nullptr;
#define AMC_FOLD_W(KS, RATIO)                                                                                         \
    hipLaunchKernelGGL((amc::fold_log_kernel<KS, RATIO, uint32_t, false>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, \
                       h->d_log, h->log_fill, h->d_acc, h->d_tot, no_hi, no_hi, h->M, h->M_pad, 0, h->t_counted, ratio_dst, RATIO_STRIDE)
#define AMC_FOLD_N(KS, RATIO, HIGH)                                                                                   \
    hipLaunchKernelGGL((amc::fold_log_kernel<KS, RATIO, uint16_t, HIGH>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, \
                       h->d_log, h->log_fill, h->d_acc16, h->d_tot16, h->d_acc_hi, h->d_tot_hi, h->M, h->M_pad, 0, h->t_counted, \
                       ratio_dst, RATIO_STRIDE)
#define AMC_FOLD(KS, RATIO)                                                                                           \
    do {                                                                                                              \
        if (!h->narrow) AMC_FOLD_W(KS, RATIO);                                                                        \
        else if (h->use_high) AMC_FOLD_N(KS, RATIO, true);                                                            \
        else AMC_FOLD_N(KS, RATIO, false);                                                                            \
    } while (0)
    if (with_ratio) {
        switch (h->K) {
        case 1: AMC_FOLD(1, true); break;
        case 2: AMC_FOLD(2, true); break;
        case 3: AMC_FOLD(3, true); break;
        case 4: AMC_FOLD(4, true); break;
        default: return fail(AMC_ERR_STATE, "fold_log: ratio sums ride on the K <= 4 fold only");
        }
        if (ratio_rows) *ratio_rows = grid;
    } else {
        switch (h->K) {
        case 1: AMC_FOLD(1, false); break;
        case 2: AMC_FOLD(2, false); break;
        case 3: AMC_FOLD(3, false); break;
        case 4: AMC_FOLD(4, false); break;
        default: {
            // more than four moves: ceil(K / 4) passes of the four-move form, one per group of moves (fold_log_kernel<.., GROUP>)
            const bool bytes = log_form(h) == AMC_LOG_BYTES;
            const int n_groups = (h->K + 3) / 4;
#define AMC_FOLD_GROUP(KS, GROUP, BYTES)                                                                              \
    hipLaunchKernelGGL((amc::fold_log_kernel<KS, false, uint32_t, false, GROUP, BYTES>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, \
                       h->d_log, h->log_fill, acc_g, tot_g, no_hi, no_hi, h->M, h->M_pad, g, h->t_counted, ratio_dst, RATIO_STRIDE)
#define AMC_FOLD_GROUPS(BYTES)                                                                                        \
    for (int g = 0; g < n_groups; ++g) {                                                                              \
        uint32_t* const acc_g = h->d_acc + 4 * (size_t)g * (size_t)h->M_pad;                                          \
        uint32_t* const tot_g = h->d_tot + 4 * (size_t)g * (size_t)h->M_pad;                                          \
        if (g + 1 < n_groups) AMC_FOLD_GROUP(4, 1, BYTES);                                                            \
        else switch (h->K - 4 * g) {                                                                                  \
            case 1: AMC_FOLD_GROUP(1, 2, BYTES); break;                                                               \
            case 2: AMC_FOLD_GROUP(2, 2, BYTES); break;                                                               \
            case 3: AMC_FOLD_GROUP(3, 2, BYTES); break;                                                               \
            default: AMC_FOLD_GROUP(4, 2, BYTES); break;                                                              \
        }                                                                                                             \
    }
            if (bytes) { AMC_FOLD_GROUPS(true) } else { AMC_FOLD_GROUPS(false) }
#undef AMC_FOLD_GROUPS
#undef AMC_FOLD_GROUP
            break;
        }
        }
    }
#undef AMC_FOLD
#undef AMC_FOLD_N
#undef AMC_FOLD_W
    AMC_HIP(hipGetLastError());
    h->log_fill = 0;
    return AMC_OK;
}

// Allocates the per-chain counter arrays, zeroed: two u16 planes per counter (narrow) or u32 arrays.
static hipError_t alloc_counters(amc_handle* h, bool narrow)
{
    const size_t n = (size_t)h->K * (size_t)h->M_pad;
    const size_t nt = (size_t)(h->K - 1) * (size_t)h->M_pad;      // K - 1 rows: the last move's total_calls is the step count
    h->narrow = narrow;                                            // minus the others (fold_log_kernel)
    h->use_high = false;
    auto zeroed = [&](void** p, size_t bytes) {
        if (bytes == 0) return hipSuccess;
        const hipError_t e = hipMalloc(p, bytes);
        return e != hipSuccess ? e : hipMemsetAsync(*p, 0, bytes, h->stream);
    };
    hipError_t e;
    if (!narrow) {
        if ((e = zeroed((void**)&h->d_acc, n * sizeof(uint32_t))) != hipSuccess) return e;
        return zeroed((void**)&h->d_tot, nt * sizeof(uint32_t));
    }
    if ((e = zeroed((void**)&h->d_acc16, n * sizeof(uint16_t))) != hipSuccess) return e;
    if ((e = zeroed((void**)&h->d_acc_hi, n * sizeof(uint16_t))) != hipSuccess) return e;
    if ((e = zeroed((void**)&h->d_tot16, nt * sizeof(uint16_t))) != hipSuccess) return e;
    return zeroed((void**)&h->d_tot_hi, nt * sizeof(uint16_t));
}

// u16 planes suit a handle with K <= 4 (the register-resident fold); AMC_WIDE_COUNTERS=1 keeps plain u32 arrays (A/B, tests)
static bool narrow_counters_allowed(const amc_handle* h)
{
    static const bool forced_wide = [] { const char* e = getenv("AMC_WIDE_COUNTERS"); return e && *e && *e != '0'; }();
    return h->counters && h->K <= 4 && !forced_wide;
}

// Makes room for at least one more row of the step log (a full log is folded first); *rows = how many fit.
int log_room(amc_handle* h, int* rows)
{
    if (h->log_fill == h->log_depth) {
        const int rc = fold_log(h);
        if (rc != AMC_OK) return rc;
    }
    *rows = h->log_depth - h->log_fill;
    return AMC_OK;
}

// The form of a launch that also leaves the callback sums (amc::RED_FORM_*): the one with sum e alone compiled in when the
// callbacks read nothing else (amc_set_reduce_columns; harmonic potential, Float64 state: sum x^2 is the same sum), else the one
// that forms whatever SweepArgs.red_cols names.
int red_form(const amc_handle* h)
{
    const int e_alone = (h->potential == AMC_POTENTIAL_HARMONIC && !h->f32) ? (amc::RED_WANT_E | amc::RED_WANT_XX) : amc::RED_WANT_E;
    return (h->red_cols & ~e_alone) == 0 ? amc::RED_FORM_E : amc::RED_FORM_COLS;
}

template <int POT, bool MULTI, int LOG, int FORM>
int launch_sweep_reduce_mlf(amc_handle* h, const amc::SweepArgs& a, int grid)
{
    if (a.n_steps == 1) {
        if (h->beta_arr)
            hipLaunchKernelGGL((amc::sweep_kernel<POT, MULTI, LOG, true, true, FORM>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, a);
        else
            hipLaunchKernelGGL((amc::sweep_kernel<POT, MULTI, LOG, false, true, FORM>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, a);
    } else {
        if (h->beta_arr)
            hipLaunchKernelGGL((amc::sweep_kernel<POT, MULTI, LOG, true, false, FORM>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, a);
        else
            hipLaunchKernelGGL((amc::sweep_kernel<POT, MULTI, LOG, false, false, FORM>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, a);
    }
    AMC_HIP(hipGetLastError());
    return AMC_OK;
}

template <int POT, bool MULTI, int LOG>
int launch_sweep_reduce_ml(amc_handle* h, const amc::SweepArgs& a, int grid)
{
    return red_form(h) == amc::RED_FORM_E ? launch_sweep_reduce_mlf<POT, MULTI, LOG, amc::RED_FORM_E>(h, a, grid)
                                          : launch_sweep_reduce_mlf<POT, MULTI, LOG, amc::RED_FORM_COLS>(h, a, grid);
}

template <int POT>
int launch_sweep_reduce(amc_handle* h, const amc::SweepArgs& a, int grid)
{
    if (h->K > 1 && log_form(h) == AMC_LOG_PACKED) return launch_sweep_reduce_ml<POT, true, AMC_LOG_PACKED>(h, a, grid);
    if (h->K > 1) return launch_sweep_reduce_ml<POT, true, AMC_LOG_BYTES>(h, a, grid);
    if (h->counters) return launch_sweep_reduce_ml<POT, false, AMC_LOG_PACKED>(h, a, grid);
    return launch_sweep_reduce_ml<POT, false, AMC_LOG_NONE>(h, a, grid);
}

template <int POT>
int launch_sweep(amc_handle* h, const amc::SweepArgs& a, int grid)
{
    return a.n_steps == 1 ? launch_sweep_s<POT, true>(h, a, grid) : launch_sweep_s<POT, false>(h, a, grid);
}

// Pool-wide accepted total (K == 1): sum of the per-block slots the sweep kernel maintains.
static int sum_acc_slots(amc_handle* h, unsigned long long* out)
{
    std::vector<unsigned long long> slots((size_t)h->n_slots);
    AMC_HIP(hipMemcpyAsync(slots.data(), h->d_acc_slots, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));
    unsigned long long t = 0;
    for (unsigned long long v : slots) t += v;
    *out = t;
    return AMC_OK;
}

int nl_capacity(int n_learn) { return n_learn <= 1 ? 1 : n_learn <= 2 ? 2 : n_learn <= 4 ? 4 : 8; }


static const char* tf(bool b) { return b ? "true" : "false"; }

// sweep_kernel<POT_CUSTOM, MULTI, LOG, BETA, SINGLE, REDUCE> with the flags launch_sweep_s / launch_sweep_reduce pick
static int launch_sweep_custom(amc_handle* h, amc::SweepArgs& a, int grid, bool reduce)
{
    const bool multi = h->K > 1;
    const std::string inst = "amc::sweep_kernel<" + std::to_string(h->potential) + "," + tf(multi) + "," + std::to_string(log_form(h)) + "," + tf(h->beta_arr) + "," +
                             tf(a.n_steps == 1) + "," + std::to_string(reduce ? red_form(h) : (int)amc::RED_FORM_NONE) + ">";
    void* params[] = {&a};
    return rtc_launch(h, inst, grid, params);
}

// Wait for everything queued on the stream.  The runtime's blocking wait parks the thread on an interrupt after a
// short spin and wakes it tens of microseconds after the device is done -- as long as a whole sweep; a host that steps
// the engine (callbacks, short timed regions) sees that latency on every hand-over.  So: poll the stream for up to 5 ms
// (a query is a read of the queue's completion signal), then fall back to the blocking wait.
hipError_t wait_stream(hipStream_t stream)
{
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(stream);
        if (e != hipErrorNotReady) return e;
        if ((spins & 63) == 63) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 5000000ll) break;
        }
    }
    return hipStreamSynchronize(stream);
}

static hipError_t wait_event(hipEvent_t ev)
{
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int spins = 0;; ++spins) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if ((spins & 63) == 63) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 5000000ll) break;
        }
    }
    return hipEventSynchronize(ev);
}


extern "C" {

const char* amc_last_error(void) { return g_last_error.c_str(); }

int amc_version(void) { return AMC_VERSION_MAJOR * 1000 + AMC_VERSION_MINOR; }

int amc_device_count(int* count)
{
    if (!count) return fail(AMC_ERR_BAD_ARG, "amc_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(AMC_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return AMC_OK;
}

struct ClassExprs { const char *sample, *logq, *dlogq, *perform, *invert; };
struct ProposalExprs { const char *sample, *logq, *dlogq, *perform, *invert; int n_params; const char* const* dlogq_more;   // dlogq_more: partials 1 .. n_params - 1
                       int n_classes; const ClassExprs* more_classes; const int* class_of_move; };   // pools that mix policies / actions: classes 1 .. n_classes - 1

// The expressions of a script-defined proposal, checked as text (validate_potential_expr: one line of ordinary expression text
// that mentions what it must).
static int validate_proposal_exprs(const ProposalExprs* proposal)
{
    int rc_p = validate_potential_expr(proposal->sample, "sample_action expression", "z");
    if (rc_p == AMC_OK) rc_p = validate_potential_expr(proposal->logq, "log_proposal_density expression", "delta");
    if (rc_p == AMC_OK && proposal->dlogq) {
        rc_p = validate_potential_expr(proposal->dlogq, "d log_proposal_density / d sigma expression", proposal->n_params > 1 ? "" : "sigma");
        if (rc_p != AMC_OK && proposal->n_params == 1 &&         // theta0 is another name of sigma
            validate_potential_expr(proposal->dlogq, "d log_proposal_density / d sigma expression", "theta0") == AMC_OK)
            rc_p = AMC_OK;
    }
    for (int pidx = 1; rc_p == AMC_OK && proposal->dlogq && pidx < proposal->n_params; ++pidx)
        rc_p = validate_potential_expr(proposal->dlogq_more[pidx - 1], "d log_proposal_density / d theta expression", "");
    if (rc_p == AMC_OK && (proposal->perform != nullptr) != (proposal->invert != nullptr))
        rc_p = fail(AMC_ERR_BAD_ARG, "amc_create_action_model: perform_expr and invert_expr come together (No invert_action! is defined)");
    if (rc_p == AMC_OK && proposal->perform) rc_p = validate_potential_expr(proposal->perform, "perform_action expression", "delta");
    if (rc_p == AMC_OK && proposal->invert) rc_p = validate_potential_expr(proposal->invert, "invert_action expression", "delta");
    for (int c = 1; rc_p == AMC_OK && c < proposal->n_classes; ++c) {
        const ClassExprs& ce = proposal->more_classes[c - 1];
        rc_p = validate_potential_expr(ce.sample, "sample_action expression", "z");
        if (rc_p == AMC_OK) rc_p = validate_potential_expr(ce.logq, "log_proposal_density expression", "delta");
        if (rc_p == AMC_OK && ce.dlogq) rc_p = validate_potential_expr(ce.dlogq, "d log_proposal_density / d sigma expression", "");
        if (rc_p == AMC_OK && ce.perform) rc_p = validate_potential_expr(ce.perform, "perform_action expression", "delta");
        if (rc_p == AMC_OK && ce.invert) rc_p = validate_potential_expr(ce.invert, "invert_action expression", "delta");
    }
    return rc_p;
}

// What the run-time compiler is given for a script-defined model: the expressions in one string, section marks between them
// (amc_rtc.hip rtc_compile takes it apart again).  A derivative section that is absent means: differentiate logq (amc_dual.h).
static std::string encode_model_expr(bool f32, const char* potential_expr, const char* reward_expr, const char* scale_expr, const ProposalExprs* proposal)
{
    std::string e;
    if (f32) e = "\x02";
    if (potential_expr) e += potential_expr;
    if (potential_expr && reward_expr) e += std::string("\x01") + reward_expr;
    if (potential_expr && scale_expr) e += std::string("\x03") + scale_expr;
    if (potential_expr && proposal) {
        e += std::string("\x04") + proposal->sample + std::string("\x05") + proposal->logq;
        if (proposal->dlogq) {
            e += std::string("\x06") + proposal->dlogq;
            for (int pidx = 1; pidx < proposal->n_params; ++pidx) e += std::string("\x0b") + proposal->dlogq_more[pidx - 1];
        }
        if (proposal->perform) e += std::string("\x07") + proposal->perform + std::string("\x08") + proposal->invert;
        if (proposal->n_params > 1) e += std::string("\x0e") + std::to_string(proposal->n_params);
        if (proposal->n_classes > 1) {
            // [ '\x0f' n_classes { '\x10' sample '\x11' logq '\x12' dlogq '\x13' perform '\x14' invert } per class 1 .. ]: empty = not given
            e += std::string("\x0f") + std::to_string(proposal->n_classes);
            for (int c = 1; c < proposal->n_classes; ++c) {
                const ClassExprs& ce = proposal->more_classes[c - 1];
                e += std::string("\x10") + ce.sample + std::string("\x11") + ce.logq + std::string("\x12") + (ce.dlogq ? ce.dlogq : "") +
                     std::string("\x13") + (ce.perform ? ce.perform : "") + std::string("\x14") + (ce.invert ? ce.invert : "");
            }
        }
    }
    return e;
}

static int create_impl(const amc_config* cfg, const char* potential_expr, amc_handle** out, const char* reward_expr = nullptr,
                       const char* scale_expr = nullptr, const ProposalExprs* proposal = nullptr)
{
    if (!cfg || !out) return fail(AMC_ERR_BAD_ARG, "amc_create: NULL argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(amc_config) && cfg->struct_size != AMC_CONFIG_SIZE_V0_1)
        return fail(AMC_ERR_BAD_ARG, "amc_create: struct_size %u != %zu (ABI mismatch)", cfg->struct_size, sizeof(amc_config));
    const int state_dtype = cfg->struct_size == sizeof(amc_config) ? cfg->state_dtype : (int)AMC_DTYPE_F64;
    if (state_dtype != AMC_DTYPE_F64 && state_dtype != AMC_DTYPE_F32)
        return fail(AMC_ERR_BAD_ARG, "amc_create: unknown state_dtype %d", state_dtype);
    if (cfg->struct_size == sizeof(amc_config) && cfg->reserved != 0)
        return fail(AMC_ERR_BAD_ARG, "amc_create: reserved must be 0");
    if (cfg->n_chains < 1) return fail(AMC_ERR_BAD_ARG, "amc_create: n_chains must be >= 1");
    if (cfg->chain_offset < 0 || (cfg->chain_offset & 1))
        return fail(AMC_ERR_BAD_ARG, "amc_create: chain_offset must be even and >= 0 (shards split on chain pairs)");
    if (cfg->n_chains_global < cfg->chain_offset + cfg->n_chains)
        return fail(AMC_ERR_BAD_ARG, "amc_create: n_chains_global < chain_offset + n_chains");
    if (cfg->n_moves < 1 || cfg->n_moves > AMC_MAX_MOVES)
        return fail(AMC_ERR_BAD_ARG, "amc_create: n_moves must be in [1, %d]", AMC_MAX_MOVES);
    if (cfg->sweepstep < 1) return fail(AMC_ERR_BAD_ARG, "amc_create: sweepstep must be >= 1");
    if (cfg->potential == AMC_POTENTIAL_CUSTOM) {
        if (!potential_expr)
            return fail(AMC_ERR_BAD_ARG, "amc_create: AMC_POTENTIAL_CUSTOM needs its expression: use amc_create_custom");
        const int rc_expr = validate_potential_expr(potential_expr);
        if (rc_expr != AMC_OK) return rc_expr;
        if (reward_expr) {
            const int rc_rew = validate_potential_expr(reward_expr, "custom reward", "delta");
            if (rc_rew != AMC_OK) return rc_rew;
        }
        if (scale_expr) {
            const int rc_sc = validate_potential_expr(scale_expr, "proposal-width scale", "x");
            if (rc_sc != AMC_OK) return rc_sc;
        }
        if (proposal) {
            const int rc_p = validate_proposal_exprs(proposal);
            if (rc_p != AMC_OK) return rc_p;
        }
    } else if (potential_expr) {
        return fail(AMC_ERR_BAD_ARG, "amc_create_custom: cfg->potential must be AMC_POTENTIAL_CUSTOM");
    } else if (cfg->potential != AMC_POTENTIAL_HARMONIC && cfg->potential != AMC_POTENTIAL_DOUBLE_WELL) {
        return fail(AMC_ERR_BAD_ARG, "amc_create: unknown potential id %d", cfg->potential);
    }
    if (!cfg->sigma || !cfg->weight) return fail(AMC_ERR_BAD_ARG, "amc_create: sigma/weight is NULL");
    double wsum = 0.0;
    for (int k = 0; k < cfg->n_moves; ++k) {
        if (!(cfg->sigma[k] >= 1e-100) || !(cfg->sigma[k] <= 1e100))
            return fail(AMC_ERR_BAD_ARG, "amc_create: sigma[%d] must lie in [1e-100, 1e100]", k);
        if (!(cfg->weight[k] >= 0.0) || !std::isfinite(cfg->weight[k]))
            return fail(AMC_ERR_BAD_ARG, "amc_create: weight[%d] must be finite and >= 0", k);
        wsum += cfg->weight[k];
    }
    // Categorical(weights) requires a probability vector (isprobvec: sum ~ 1, rtol sqrt(eps))
    if (std::fabs(wsum - 1.0) > 1.4901161193847656e-08)
        return fail(AMC_ERR_BAD_ARG, "amc_create: weights must sum to 1 (got %.17g)", wsum);
    if (cfg->n_moves > 1 && !cfg->per_chain_counters) {
        // K > 1 needs per-chain (accepted, total) pairs for callback_acceptance's mean of ratios
    }

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev < 1)
        return fail(AMC_ERR_NO_DEVICE, "amc_create: no HIP device (%s); this engine has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "count = 0");
    if (cfg->device < 0 || cfg->device >= n_dev)
        return fail(AMC_ERR_BAD_ARG, "amc_create: device %d out of range [0, %d)", cfg->device, n_dev);
    AMC_HIP(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    AMC_HIP(hipGetDeviceProperties(&prop, cfg->device));

    amc_handle* h = new (std::nothrow) amc_handle();
    if (!h) return fail(AMC_ERR_OOM, "amc_create: host allocation failed");
    h->device = cfg->device;
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        // "gfx950:sramecc+:xnack-" -> "gfx950": run-time compiled kernels target the device they will run on; the offline
        // kernels of this library were built for AMC_BUILD_ARCH and cannot run anywhere else
        std::string arch(prop.gcnArchName);
        const size_t colon = arch.find(':');
        if (colon != std::string::npos) arch.erase(colon);
        if (!arch.empty()) h->arch = arch;
        if (h->arch != AMC_BUILD_ARCH) {
            delete h;
            return fail(AMC_ERR_NO_DEVICE, "amc_create: device %d is %s, this libamc.so was built for %s (make ARCH=%s)", cfg->device,
                        arch.c_str(), AMC_BUILD_ARCH, arch.c_str());
        }
    }
    // 8 resident blocks per CU; the single-step launch of the K = 1 pool-wide-counter sweep (no step log) measures 5 %
    // faster with 6 (29.4 vs 31.2 us at 1e7 chains; its fused launches and all other forms are fastest at 8)
    // (round 5: the K > 1 single-step launch holds 7 blocks per CU -- 70 VGPRs -- and ONE round of them is 4 % faster than 8 on
    // 7 slots, 31.1 against 32.4 us; env AMC_BLOCKS_PER_CU_SINGLE)
    h->blocks_per_cu = 8;
    h->blocks_per_cu_single = (cfg->n_moves == 1 && !cfg->per_chain_counters) ? 6 : (cfg->n_moves > 1 ? 7 : 8);
    if (const char* env = std::getenv("AMC_BLOCKS_PER_CU")) {   // tuning knob, 1..64
        const int v = std::atoi(env);
        if (v >= 1 && v <= 64) h->blocks_per_cu = h->blocks_per_cu_single = h->blocks_per_cu_pg = v;
    }
    if (const char* env = std::getenv("AMC_BLOCKS_PER_CU_SINGLE")) {   // tuning knob, 1..64
        const int v = std::atoi(env);
        if (v >= 1 && v <= 64) h->blocks_per_cu_single = v;
    }
    if (const char* env = std::getenv("AMC_BLOCKS_PER_CU_REDUCE")) {   // tuning knob, 1..64
        const int v = std::atoi(env);
        if (v >= 1 && v <= 64) h->blocks_per_cu_red = v;
    }
    if (const char* env = std::getenv("AMC_EXACT_ACCEPT")) h->exact_accept = std::atoi(env) != 0;
    if (const char* env = std::getenv("AMC_WIDE_RED_ROWS")) h->wide_red_rows = std::atoi(env) != 0;
    if (const char* env = std::getenv("AMC_SHARD_ROUTE_ON_ONE_RANK")) h->shard_route_one_rank = std::atoi(env) != 0;
    if (const char* env = std::getenv("AMC_NO_DEFERRED_UPDATE")) h->no_deferred_update = std::atoi(env) != 0;
    if (const char* env = std::getenv("AMC_CLASS_PER_MOVE")) h->class_per_move_forced = std::atoi(env) != 0;
    if (const char* env = std::getenv("AMC_NO_COLUMN_SKIP")) h->no_column_skip = std::atoi(env) != 0;
    if (const char* env = std::getenv("AMC_NP_SMALL_LAUNCHES")) h->np_small_launches = std::atoi(env) != 0;
    h->M = cfg->n_chains;
    // padding: unclamped 16-B tail loads stay in bounds; rows of every per-chain array start on a 256-byte boundary
    // (M_pad is a multiple of 256): a wave's 128-byte step-log store then covers exactly one aligned line
    h->M_pad = ((cfg->n_chains + AMC_PAD_DOUBLES + 255) / 256) * 256;
    h->offset = cfg->chain_offset;
    h->M_global = cfg->n_chains_global;
    h->potential = cfg->potential;
    h->f32 = state_dtype == AMC_DTYPE_F32;
    h->use_rtc = h->f32 || cfg->potential == AMC_POTENTIAL_CUSTOM;
    h->pot_expr = encode_model_expr(h->f32, potential_expr, reward_expr, scale_expr, proposal);
    if (potential_expr && scale_expr) h->scaled_policy = true;
    if (potential_expr && proposal) {
        if (proposal->n_classes > 1) {
            h->n_classes = proposal->n_classes;
            for (int k = 0; k < cfg->n_moves; ++k) h->class_of_move[k] = proposal->class_of_move[k];
        }
        h->script_policy = true;
        h->script_dlogq = proposal->dlogq != nullptr;
        h->n_params = proposal->n_params;
    }
    h->K = cfg->n_moves;
    h->sweepstep = cfg->sweepstep;
    h->counters = cfg->per_chain_counters != 0 || cfg->n_moves > 1;
    h->beta = cfg->beta;
    h->seed = cfg->seed;

    int rc = AMC_OK;
    auto bail = [&](int code) { amc_destroy(h); return code; };
#define AMC_TRY(call)                                                                   \
    do {                                                                                \
        hipError_t e2_ = (call);                                                        \
        if (e2_ != hipSuccess) {                                                        \
            rc = fail(e2_ == hipErrorOutOfMemory ? AMC_ERR_OOM : AMC_ERR_HIP,           \
                      "%s failed: %s", #call, hipGetErrorString(e2_));                  \
            return bail(rc);                                                            \
        }                                                                               \
    } while (0)

    if (cfg->stream) {
        h->stream = (hipStream_t)cfg->stream;
    } else {
        AMC_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    AMC_TRY(hipMalloc(&h->d_x, (size_t)h->M_pad * sizeof(double)));
    AMC_TRY(hipMemsetAsync(h->d_x, 0, (size_t)h->M_pad * sizeof(double), h->stream));
    if (h->counters) {
        AMC_TRY(alloc_counters(h, narrow_counters_allowed(h)));
        // Half a byte (K <= 8) or one byte per chain and MH step; folding costs a read-modify-write of every counter, so a
        // deeper log amortises it over more steps: 128 rows where rows of one byte per chain fit in 2 GiB (0.64 / 1.28 GB at
        // 1e7 chains), never below 16.
        {
            const int64_t fit = (int64_t)(2147483648ll / h->M_pad);
            h->log_depth = (int)(fit > 128 ? 128 : (fit < 16 ? 16 : fit));
        }
        if (const char* env = std::getenv("AMC_LOG_DEPTH")) {     // tuning knob, 1..255
            const int v = std::atoi(env);
            if (v >= 1 && v <= 255) h->log_depth = v;
        }
        // K <= AMC_PACKED_LOG_MOVES: two chains per byte (store_log_pair)
        const size_t row_bytes = log_form(h) == AMC_LOG_PACKED ? (size_t)h->M_pad / 2 : (size_t)h->M_pad;
        AMC_TRY(hipMalloc(&h->d_log, (size_t)h->log_depth * row_bytes));
        AMC_TRY(hipMemsetAsync(h->d_log, 0, (size_t)h->log_depth * row_bytes, h->stream));
    }
    AMC_TRY(hipMalloc(&h->d_ptab, (size_t)amc::PT_ROWS * AMC_MAX_MOVES * sizeof(double)));
    AMC_TRY(hipMemsetAsync(h->d_ptab, 0, (size_t)amc::PT_ROWS * AMC_MAX_MOVES * sizeof(double), h->stream));
    AMC_TRY(hipMalloc(&h->d_pick, AMC_PICK_CELLS));
    AMC_TRY(hipMemsetAsync(h->d_pick, 0, AMC_PICK_CELLS, h->stream));
    AMC_TRY(hipMalloc(&h->d_totals, 2 * AMC_MAX_MOVES * sizeof(unsigned long long)));
    AMC_TRY(hipMemsetAsync(h->d_totals, 0, 2 * AMC_MAX_MOVES * sizeof(unsigned long long), h->stream));
    // room for two rounds of the estimator kernels' resident blocks (pg_plan: up to 2 x 5 per CU) beside the sweeps' 8 per CU
    const int slots_per_cu = std::max(std::max(std::max(h->blocks_per_cu, h->blocks_per_cu_single), h->blocks_per_cu_red), 10);
    h->n_slots = h->n_cu * slots_per_cu;
    AMC_TRY(hipMalloc(&h->d_acc_slots, (size_t)h->n_slots * sizeof(unsigned long long)));
    AMC_TRY(hipMemsetAsync(h->d_acc_slots, 0, (size_t)h->n_slots * sizeof(unsigned long long), h->stream));
    h->red_blocks = grid_for(h, h->M, slots_per_cu);
    AMC_TRY(hipMalloc(&h->d_partials, (size_t)(h->red_blocks + amc::PG_GROUP) * PG_MAX_COLS * amc::XS_ROW_R * sizeof(amc::xs_word)));   // whole groups
    for (int i = 0; i < RED_TICKETS; ++i) {
        RedTicket& t = h->red[i];
        AMC_TRY(hipHostMalloc((void**)&t.h_rows, (size_t)h->n_slots * RED_HOST_STRIDE * sizeof(amc::xs_word), 0));
        AMC_TRY(hipHostMalloc((void**)&t.h_ratio, (size_t)h->n_slots * RATIO_STRIDE * amc::XS_ROW_Q * sizeof(amc::xs_word), 0));
        AMC_TRY(hipMalloc(&t.d_ratio_acc, (size_t)AMC_MAX_MOVES * 3 * sizeof(unsigned long long)));
        AMC_TRY(hipHostMalloc((void**)&t.h_ratio_acc, (size_t)AMC_MAX_MOVES * 3 * sizeof(unsigned long long), 0));
        AMC_TRY(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming));
    }
    AMC_TRY(hipMalloc(&h->d_out, (size_t)PG_MAX_COLS * amc::xs::XS_WORDS * sizeof(double)));
    AMC_TRY(hipHostMalloc((void**)&h->h_pg_out, (size_t)AMC_MAX_LEARN * PG_NP_MAX_COLS * amc::xs::XS_WORDS * sizeof(double), 0));
    AMC_TRY(hipMalloc(&h->d_gd_acc, (size_t)AMC_MAX_MOVES * AMC_GD_STRIDE_MAX * sizeof(double)));
    AMC_TRY(hipMemsetAsync(h->d_gd_acc, 0, (size_t)AMC_MAX_MOVES * AMC_GD_STRIDE_MAX * sizeof(double), h->stream));
    AMC_TRY(hipMalloc(&h->d_status, sizeof(int)));
    AMC_TRY(hipMemsetAsync(h->d_status, 0, sizeof(int), h->stream));
    {
        const size_t groups = (size_t)(h->n_slots + amc::PG_GROUP - 1) / amc::PG_GROUP + 1;
        AMC_TRY(hipMalloc(&h->d_pg_tickets, (groups + 1) * sizeof(uint32_t)));
        AMC_TRY(hipMemsetAsync(h->d_pg_tickets, 0, (groups + 1) * sizeof(uint32_t), h->stream));
        AMC_TRY(hipMalloc(&h->d_pg_groups, (size_t)2 * amc::PG_PARITY_WORDS * sizeof(amc::xs_word)));   // group rows, by the parity of the estimator step
        AMC_TRY(hipMalloc(&h->d_theta_ring, (size_t)2 * AMC_MAX_LEARN * sizeof(double)));
        AMC_TRY(hipMemsetAsync(h->d_theta_ring, 0, (size_t)2 * AMC_MAX_LEARN * sizeof(double), h->stream));
        AMC_TRY(hipMalloc(&h->d_pg_tail, sizeof(amc::PgTail)));
    }
    AMC_TRY(hipEventCreate(&h->ev0));
    AMC_TRY(hipEventCreate(&h->ev1));
    AMC_TRY(hipEventCreateWithFlags(&h->ev_params, hipEventDisableTiming));
    AMC_TRY(hipHostMalloc((void**)&h->h_params, (size_t)AMC_MAX_NP * AMC_MAX_MOVES * sizeof(double), 0));
#undef AMC_TRY
    rc = push_params(h, cfg->sigma, cfg->weight);
    if (rc != AMC_OK) return bail(rc);
    if (h->n_classes > 1) {            // the moves' classes: a row of the parameter table (amc_kernels.h PT_CLASS)
        double cls[AMC_MAX_MOVES];
        for (int k = 0; k < AMC_MAX_MOVES; ++k) cls[k] = (double)h->class_of_move[k];
        if (hipMemcpy(h->d_ptab + amc::PT_CLASS * AMC_MAX_MOVES, cls, sizeof(cls), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(AMC_ERR_HIP, "amc_create_mixed_model: copying the class table failed"));
    }
    if (h->use_rtc) {
        // compile the smallest kernel now so that a malformed expression fails HERE, with the compiler's message
        hipFunction_t fn = nullptr;
        rc = rtc_function(h, "amc::energy_kernel<" + std::to_string(h->potential) + ">", &fn);
        if (rc != AMC_OK) return bail(rc);
    }
    if (h->f32) {
        hipError_t e64 = hipMalloc(&h->d_x64, (size_t)h->M_pad * sizeof(double));
        if (e64 != hipSuccess) return bail(fail(AMC_ERR_OOM, "amc_create: %s", hipGetErrorString(e64)));
    }
    *out = h;
    return AMC_OK;
}

int amc_create(const amc_config* cfg, amc_handle** out) { return create_impl(cfg, nullptr, out); }

int amc_create_custom(const amc_config* cfg, const char* potential_expr, amc_handle** out)
{
    if (!potential_expr) return fail(AMC_ERR_BAD_ARG, "amc_create_custom: potential_expr is NULL");
    return create_impl(cfg, potential_expr, out);
}

// The creators of run-time compiled models: the potential's text (the script's, or the built-in's own expression where the script
// names cfg->potential -- compiled at run time it is the built-in bit for bit; with Float32 state the double well subtracts a Float32
// one, as Julia's `(x^2 - 1)^2` does) and the caller's config as an AMC_POTENTIAL_CUSTOM one (a 0.1 caller's struct is shorter).
static int as_custom_model(const amc_config* cfg, const char* potential_expr, const char* who, const char** pot, amc_config* c2)
{
    *pot = potential_expr;
    if (!*pot) {
        const bool f32 = cfg->struct_size == sizeof(amc_config) && cfg->state_dtype == AMC_DTYPE_F32;
        if (cfg->potential == AMC_POTENTIAL_HARMONIC) *pot = "x*x";
        else if (cfg->potential == AMC_POTENTIAL_DOUBLE_WELL) *pot = f32 ? "(x*x - 1.0f)*(x*x - 1.0f)" : "(x*x - 1.0)*(x*x - 1.0)";
        else return fail(AMC_ERR_BAD_ARG, "%s: potential_expr is NULL and cfg->potential names no built-in", who);
    }
    std::memset(c2, 0, sizeof(*c2));
    std::memcpy(c2, cfg, cfg->struct_size < sizeof(*c2) ? (cfg->struct_size >= 4 ? cfg->struct_size : 4) : sizeof(*c2));
    c2->potential = AMC_POTENTIAL_CUSTOM;
    return AMC_OK;
}

int amc_create_model(const amc_config* cfg, const char* potential_expr, const char* reward_expr, amc_handle** out)
{
    if (!cfg) return fail(AMC_ERR_BAD_ARG, "amc_create_model: NULL argument");
    // a built-in potential with a custom reward: the built-in's own expression, compiled at run time (bit-identical)
    const char* pot = nullptr;
    amc_config c2;
    { const int rc = as_custom_model(cfg, potential_expr, "amc_create_model", &pot, &c2); if (rc != AMC_OK) return rc; }
    return create_impl(&c2, pot, out, reward_expr);
}

int amc_create_policy_model(const amc_config* cfg, const char* potential_expr, const char* reward_expr, const char* scale_expr,
                            amc_handle** out)
{
    if (!cfg) return fail(AMC_ERR_BAD_ARG, "amc_create_policy_model: NULL argument");
    if (!scale_expr) return amc_create_model(cfg, potential_expr, reward_expr, out);
    const char* pot = nullptr;
    amc_config c2;
    { const int rc = as_custom_model(cfg, potential_expr, "amc_create_policy_model", &pot, &c2); if (rc != AMC_OK) return rc; }
    return create_impl(&c2, pot, out, reward_expr, scale_expr);
}

int amc_create_action_model(const amc_config* cfg, const char* potential_expr, const char* reward_expr, const char* sample_expr,
                            const char* logq_expr, const char* dlogq_expr, const char* perform_expr, const char* invert_expr,
                            amc_handle** out);

int amc_create_proposal_model(const amc_config* cfg, const char* potential_expr, const char* reward_expr, const char* sample_expr,
                              const char* logq_expr, const char* dlogq_expr, amc_handle** out)
{
    return amc_create_action_model(cfg, potential_expr, reward_expr, sample_expr, logq_expr, dlogq_expr, nullptr, nullptr, out);
}

int amc_create_vector_policy_model(const amc_config* cfg, int n_params, const char* potential_expr, const char* reward_expr,
                            const char* sample_expr, const char* logq_expr, const char* const* dlogq_exprs, const char* perform_expr,
                            const char* invert_expr, amc_handle** out)
{
    if (!cfg) return fail(AMC_ERR_BAD_ARG, "amc_create_vector_policy_model: NULL argument");
    if (n_params < 1 || n_params > AMC_MAX_NP)
        return fail(AMC_ERR_BAD_ARG, "amc_create_vector_policy_model: n_params must be in [1, %d]", AMC_MAX_NP);
    if (!sample_expr || !logq_expr)
        return fail(AMC_ERR_BAD_ARG, "amc_create_vector_policy_model: sample_expr and logq_expr are both required (No sample_action! / log_proposal_density is defined)");
    if (dlogq_exprs)
        for (int p = 0; p < n_params; ++p)
            if (!dlogq_exprs[p]) return fail(AMC_ERR_BAD_ARG, "amc_create_vector_policy_model: dlogq_exprs[%d] is NULL (one expression per parameter, or none at all)", p);
    const char* pot = nullptr;
    amc_config c2;
    { const int rc = as_custom_model(cfg, potential_expr, "amc_create_vector_policy_model", &pot, &c2); if (rc != AMC_OK) return rc; }
    const ProposalExprs prop = {sample_expr, logq_expr, dlogq_exprs ? dlogq_exprs[0] : nullptr, perform_expr, invert_expr, n_params,
                                dlogq_exprs ? dlogq_exprs + 1 : nullptr, 1, nullptr, nullptr};
    return create_impl(&c2, pot, out, reward_expr, nullptr, &prop);
}

int amc_create_action_model(const amc_config* cfg, const char* potential_expr, const char* reward_expr, const char* sample_expr,
                            const char* logq_expr, const char* dlogq_expr, const char* perform_expr, const char* invert_expr,
                            amc_handle** out)
{
    if (!cfg) return fail(AMC_ERR_BAD_ARG, "amc_create_proposal_model: NULL argument");
    if (!sample_expr || !logq_expr)
        return fail(AMC_ERR_BAD_ARG, "amc_create_proposal_model: sample_expr and logq_expr are both required (No sample_action! / log_proposal_density is defined)");
    const char* pot = nullptr;
    amc_config c2;
    { const int rc = as_custom_model(cfg, potential_expr, "amc_create_proposal_model", &pot, &c2); if (rc != AMC_OK) return rc; }
    const ProposalExprs prop = {sample_expr, logq_expr, dlogq_expr, perform_expr, invert_expr, 1, nullptr, 1, nullptr, nullptr};
    return create_impl(&c2, pot, out, reward_expr, nullptr, &prop);
}

int amc_create_mixed_model(const amc_config* cfg, int n_classes, const int* class_of_move, const char* potential_expr,
                           const char* reward_expr, const char* const* sample_exprs, const char* const* logq_exprs,
                           const char* const* dlogq_exprs, const char* const* perform_exprs, const char* const* invert_exprs,
                           amc_handle** out)
{
    if (!cfg || !class_of_move || !sample_exprs || !logq_exprs) return fail(AMC_ERR_BAD_ARG, "amc_create_mixed_model: NULL argument");
    if (n_classes < 1 || n_classes > AMC_MAX_CLASSES)
        return fail(AMC_ERR_BAD_ARG, "amc_create_mixed_model: n_classes must be in [1, %d]", AMC_MAX_CLASSES);
    if (cfg->n_moves < 1 || cfg->n_moves > AMC_MAX_MOVES) return fail(AMC_ERR_BAD_ARG, "amc_create_mixed_model: n_moves must be in [1, %d]", AMC_MAX_MOVES);
    for (int k = 0; k < cfg->n_moves; ++k)
        if (class_of_move[k] < 0 || class_of_move[k] >= n_classes)
            return fail(AMC_ERR_BAD_ARG, "amc_create_mixed_model: class_of_move[%d] = %d is no class", k, class_of_move[k]);
    for (int c = 0; c < n_classes; ++c) {
        if (!sample_exprs[c] || !logq_exprs[c])
            return fail(AMC_ERR_BAD_ARG, "amc_create_mixed_model: class %d has no sample / logq expression (No sample_action! / log_proposal_density is defined)", c);
        const bool p = perform_exprs && perform_exprs[c], i = invert_exprs && invert_exprs[c];
        if (p != i) return fail(AMC_ERR_BAD_ARG, "amc_create_mixed_model: class %d: perform_expr and invert_expr come together (No invert_action! is defined)", c);
    }
    const char* pot = nullptr;
    amc_config c2;
    { const int rc = as_custom_model(cfg, potential_expr, "amc_create_mixed_model", &pot, &c2); if (rc != AMC_OK) return rc; }
    ClassExprs more[AMC_MAX_CLASSES];
    for (int c = 1; c < n_classes; ++c)
        more[c - 1] = ClassExprs{sample_exprs[c], logq_exprs[c], dlogq_exprs ? dlogq_exprs[c] : nullptr, perform_exprs ? perform_exprs[c] : nullptr,
                                 invert_exprs ? invert_exprs[c] : nullptr};
    const ProposalExprs prop = {sample_exprs[0], logq_exprs[0], dlogq_exprs ? dlogq_exprs[0] : nullptr, perform_exprs ? perform_exprs[0] : nullptr,
                                invert_exprs ? invert_exprs[0] : nullptr, 1, nullptr, n_classes, more, class_of_move};
    return create_impl(&c2, pot, out, reward_expr, nullptr, &prop);
}




int amc_model_check(int n_params, int n_classes, const char* potential_expr, const char* reward_expr, const char* const* sample_exprs,
                    const char* const* logq_exprs, const char* const* dlogq_exprs, const char* const* perform_exprs,
                    const char* const* invert_exprs, char* log, int log_capacity)
{
    if (log && log_capacity > 0) log[0] = 0;
    if (!sample_exprs || !logq_exprs) return fail(AMC_ERR_BAD_ARG, "amc_model_check: NULL argument");
    if (n_classes < 1 || n_classes > AMC_MAX_CLASSES) return fail(AMC_ERR_BAD_ARG, "amc_model_check: n_classes must be in [1, %d]", AMC_MAX_CLASSES);
    if (n_params < 1 || n_params > AMC_MAX_NP || (n_params > 1 && n_classes > 1))
        return fail(AMC_ERR_BAD_ARG, "amc_model_check: n_params must be in [1, %d], and 1 for a pool of several classes", AMC_MAX_NP);
    for (int c = 0; c < n_classes; ++c) {
        if (!sample_exprs[c] || !logq_exprs[c]) return fail(AMC_ERR_BAD_ARG, "amc_model_check: class %d has no sample / logq expression", c);
        const bool p = perform_exprs && perform_exprs[c], i = invert_exprs && invert_exprs[c];
        if (p != i) return fail(AMC_ERR_BAD_ARG, "amc_model_check: class %d: perform_expr and invert_expr come together (No invert_action! is defined)", c);
    }
    // several parameters: dlogq_exprs holds the P partials of the one class (all or none); several classes: one entry per class, NULL entries allowed
    if (n_params > 1 && dlogq_exprs)
        for (int q = 0; q < n_params; ++q)
            if (!dlogq_exprs[q]) return fail(AMC_ERR_BAD_ARG, "amc_model_check: dlogq_exprs[%d] is NULL (one expression per parameter, or none at all)", q);
    const char* pot = potential_expr ? potential_expr : "x*x";
    { const int rc = validate_potential_expr(pot); if (rc != AMC_OK) return rc; }
    if (reward_expr) { const int rc = validate_potential_expr(reward_expr, "custom reward", "delta"); if (rc != AMC_OK) return rc; }
    ClassExprs more[AMC_MAX_CLASSES];
    for (int c = 1; c < n_classes; ++c)
        more[c - 1] = ClassExprs{sample_exprs[c], logq_exprs[c], dlogq_exprs ? dlogq_exprs[c] : nullptr, perform_exprs ? perform_exprs[c] : nullptr,
                                 invert_exprs ? invert_exprs[c] : nullptr};
    const int com[1] = {0};
    const ProposalExprs prop = {sample_exprs[0], logq_exprs[0], dlogq_exprs ? dlogq_exprs[0] : nullptr, perform_exprs ? perform_exprs[0] : nullptr,
                                invert_exprs ? invert_exprs[0] : nullptr, n_params, (dlogq_exprs && n_params > 1) ? dlogq_exprs + 1 : nullptr,
                                n_classes, more, com};
    { const int rc = validate_proposal_exprs(&prop); if (rc != AMC_OK) return rc; }
    // (developer knob: AMC_MODEL_CHECK_F32=1 builds the form for Float32 state)
    const char* f32_env = std::getenv("AMC_MODEL_CHECK_F32");
    const std::string expr = encode_model_expr(f32_env && f32_env[0] == '1', pot, reward_expr, nullptr, &prop);
    // the estimator kernel is the one that uses every expression (sample, logq, its derivative, perform / invert, reward)
    const RtcCode* code = nullptr;
    std::string text;
    // (developer knob: AMC_MODEL_CHECK_INST names another instantiation to build -- with AMC_RTC_CACHE_DIR the code object lands in a
    // file that llvm-objdump reads: tools/rtc_isa.py)
    const char* inst_env = std::getenv("AMC_MODEL_CHECK_INST");
    const int rc = rtc_compile(expr, inst_env && *inst_env ? inst_env : "amc::pg_estimate_kernel<2,1,false,0,0,false>", AMC_BUILD_ARCH, &code, &text);
    if (log && log_capacity > 0) {
        std::strncpy(log, text.c_str(), (size_t)log_capacity - 1);
        log[log_capacity - 1] = 0;
    }
    return rc;
}

int amc_destroy(amc_handle* h)
{
    if (!h) return AMC_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    comm_release(h);
    for (hipModule_t m : h->rtc_mods) (void)hipModuleUnload(m);
    (void)hipFree(h->d_gd_acc);
    (void)hipFree(h->d_status);
    (void)hipFree(h->d_pg_tickets);
    (void)hipFree(h->d_pg_groups);
    (void)hipFree(h->d_theta_ring);
    (void)hipFree(h->d_pg_tail);
    (void)hipFree(h->d_x);
    (void)hipFree(h->d_x64);
    (void)hipFree(h->d_beta);
    (void)hipFree(h->d_acc);
    (void)hipFree(h->d_tot);
    (void)hipFree(h->d_acc16);
    (void)hipFree(h->d_tot16);
    (void)hipFree(h->d_acc_hi);
    (void)hipFree(h->d_tot_hi);
    (void)hipFree(h->d_acc_base);
    (void)hipFree(h->d_tot_base);
    (void)hipFree(h->d_log);
    (void)hipFree(h->d_ptab);
    (void)hipFree(h->d_pick);
    (void)hipFree(h->d_totals);
    (void)hipFree(h->d_acc_slots);
    (void)hipFree(h->d_partials);
    for (int i = 0; i < RED_TICKETS; ++i) {
        RedTicket& t = h->red[i];
        if (t.h_rows) (void)hipHostFree(t.h_rows);
        if (t.h_ratio) (void)hipHostFree(t.h_ratio);
        (void)hipFree(t.d_ratio_acc);
        if (t.h_ratio_acc) (void)hipHostFree(t.h_ratio_acc);
        if (t.ev) (void)hipEventDestroy(t.ev);
    }
    (void)hipFree(h->d_out);
    if (h->h_pg_out) (void)hipHostFree(h->h_pg_out);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_params) (void)hipEventDestroy(h->ev_params);
    (void)hipFree(h->d_hist);
    if (h->h_params) (void)hipHostFree(h->h_params);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return AMC_OK;
}

// Float32 state: d_x64 (doubles) -> dst (floats, rounded to nearest) and back, on the stream.
static int narrow_from_x64(amc_handle* h, double* dst_as_float)
{
    const double* in = h->d_x64;
    int64_t n = h->M;
    void* params[] = {&in, &n, &dst_as_float};
    return rtc_launch(h, "amc::narrow_state_kernel", grid_for(h, h->M), params);
}

static int widen_to_x64(amc_handle* h)
{
    const double* in = h->d_x;
    int64_t n = h->M;
    double* out = h->d_x64;
    void* params[] = {&in, &n, &out};
    return rtc_launch(h, "amc::widen_state_kernel", grid_for(h, h->M), params);
}

// The positions as doubles on the device: d_x itself, or (Float32 state) the widened copy.
static int positions_f64(amc_handle* h, const double** out)
{
    *out = h->d_x;
    if (!h->f32) return AMC_OK;
    *out = h->d_x64;
    return widen_to_x64(h);
}

int amc_upload_state(amc_handle* h, const double* x, const double* beta_or_null)
{
    if (!h || !x) return fail(AMC_ERR_BAD_ARG, "amc_upload_state: NULL argument");
    AMC_HIP(hipSetDevice(h->device));
    if (h->f32) {
        AMC_HIP(hipMemcpyAsync(h->d_x64, x, (size_t)h->M * sizeof(double), hipMemcpyHostToDevice, h->stream));
        const int rc = narrow_from_x64(h, h->d_x);
        if (rc != AMC_OK) return rc;
    } else {
        AMC_HIP(hipMemcpyAsync(h->d_x, x, (size_t)h->M * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    if (beta_or_null) {
        if (!h->d_beta) {
            AMC_HIP(hipMalloc(&h->d_beta, (size_t)h->M_pad * sizeof(double)));
            AMC_HIP(hipMemsetAsync(h->d_beta, 0, (size_t)h->M_pad * sizeof(double), h->stream));
        }
        if (h->f32) {
            AMC_HIP(hipMemcpyAsync(h->d_x64, beta_or_null, (size_t)h->M * sizeof(double), hipMemcpyHostToDevice, h->stream));
            const int rc = narrow_from_x64(h, h->d_beta);
            if (rc != AMC_OK) return rc;
        } else {
            AMC_HIP(hipMemcpyAsync(h->d_beta, beta_or_null, (size_t)h->M * sizeof(double), hipMemcpyHostToDevice, h->stream));
        }
        h->beta_arr = true;
    }
    AMC_HIP(hipStreamSynchronize(h->stream));   // caller's buffers are only valid during the call
    return AMC_OK;
}

int amc_init_uniform(amc_handle* h, double lo, double hi)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_init_uniform: NULL handle");
    AMC_HIP(hipSetDevice(h->device));
    const int grid = grid_for(h, (h->M + 1) / 2);
    // Float32 state: System(Float32(lo + (hi - lo) u), beta) -- the Float64 ensemble, rounded
    hipLaunchKernelGGL(amc::init_uniform_kernel, dim3(grid), dim3(AMC_BLOCK), 0, h->stream, h->f32 ? h->d_x64 : h->d_x, h->M,
                       (uint64_t)h->offset >> 1, (uint32_t)h->seed, (uint32_t)(h->seed >> 32), lo, hi);
    AMC_HIP(hipGetLastError());
    if (h->f32) return narrow_from_x64(h, h->d_x);
    return AMC_OK;
}

int amc_download_state(amc_handle* h, double* x, double* e)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_download_state: NULL handle");
    if (!x && !e) return AMC_OK;
    AMC_HIP(hipSetDevice(h->device));
    double* dst = x;
    std::vector<double> tmp;
    if (!dst) { tmp.resize((size_t)h->M); dst = tmp.data(); }
    const double* d_pos = nullptr;
    { const int rc = positions_f64(h, &d_pos); if (rc != AMC_OK) return rc; }
    AMC_HIP(hipMemcpyAsync(dst, d_pos, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));
    if (e && h->use_rtc) {
        // the host cannot evaluate the user's expression (or must not: Float32 arithmetic): e = potential(x) by the
        // run-time compiled kernel
        double* d_e = nullptr;
        AMC_HIP(hipMalloc(&d_e, (size_t)h->M * sizeof(double)));
        const double* d_x = h->d_x;
        int64_t m = h->M;
        void* params[] = {&d_x, &m, &d_e};
        int rc = rtc_launch(h, "amc::energy_kernel<" + std::to_string(h->potential) + ">", h->red_blocks, params);
        hipError_t he = hipSuccess;
        if (rc == AMC_OK) he = hipMemcpyAsync(e, d_e, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (rc == AMC_OK && he == hipSuccess) he = hipStreamSynchronize(h->stream);
        (void)hipFree(d_e);
        if (rc != AMC_OK) return rc;
        if (he != hipSuccess) return fail(AMC_ERR_HIP, "amc_download_state: %s", hipGetErrorString(he));
        return AMC_OK;
    }
    if (e) {
        // e == potential(x) exactly (particle_1d.jl:33): the same two IEEE multiplies on the host
        for (int64_t c = 0; c < h->M; ++c) {
            const double xc = dst[c];
            if (h->potential == AMC_POTENTIAL_DOUBLE_WELL) {
                volatile double q = xc * xc;   // volatile: no host-side fma contraction of x*x - 1
                const double r = q - 1.0;
                e[c] = r * r;
            } else {
                e[c] = xc * xc;
            }
        }
    }
    return AMC_OK;
}

int amc_download_counters(amc_handle* h, int64_t* accepted, int64_t* total)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_download_counters: NULL handle");
    if (!h->counters)
        return fail(AMC_ERR_STATE, "amc_download_counters: handle was created with per_chain_counters = 0");
    AMC_HIP(hipSetDevice(h->device));
    { const int rc = fold_log(h); if (rc != AMC_OK) return rc; }
    std::vector<uint32_t> buf((size_t)h->M);
    // one row of counters, whatever their width on the device, as int64
    auto fetch_row = [&](const uint32_t* wide, const uint16_t* narrow, const uint16_t* high, int k, int64_t* out) -> int {
        if (h->narrow) {
            uint16_t* b16 = reinterpret_cast<uint16_t*>(buf.data());
            AMC_HIP(hipMemcpyAsync(b16, narrow + (size_t)k * h->M_pad, (size_t)h->M * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
            AMC_HIP(hipStreamSynchronize(h->stream));
            for (int64_t c = 0; c < h->M; ++c) out[c] = b16[(size_t)c];
            if (h->use_high) {
                AMC_HIP(hipMemcpyAsync(b16, high + (size_t)k * h->M_pad, (size_t)h->M * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
                AMC_HIP(hipStreamSynchronize(h->stream));
                for (int64_t c = 0; c < h->M; ++c) out[c] |= (int64_t)b16[(size_t)c] << 16;
            }
        } else {
            AMC_HIP(hipMemcpyAsync(buf.data(), wide + (size_t)k * h->M_pad, (size_t)h->M * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
            AMC_HIP(hipStreamSynchronize(h->stream));
            for (int64_t c = 0; c < h->M; ++c) out[c] = buf[(size_t)c];
        }
        return AMC_OK;
    };
    // what the arrays have been carried into (counter_rebase): 64-bit bases, added in
    std::vector<unsigned long long> bbuf(h->d_acc_base ? (size_t)h->M : 0);
    auto add_base = [&](const unsigned long long* base, int k, int64_t* out) -> int {
        if (!base) return AMC_OK;
        AMC_HIP(hipMemcpyAsync(bbuf.data(), base + (size_t)k * h->M_pad, (size_t)h->M * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        AMC_HIP(hipStreamSynchronize(h->stream));
        for (int64_t c = 0; c < h->M; ++c) out[c] += (int64_t)bbuf[(size_t)c];
        return AMC_OK;
    };
    for (int k = 0; k < h->K; ++k) {
        if (accepted) {
            int rc = fetch_row(h->d_acc, h->d_acc16, h->d_acc_hi, k, accepted + (int64_t)k * h->M);
            if (rc == AMC_OK) rc = add_base(h->d_acc_base, k, accepted + (int64_t)k * h->M);
            if (rc != AMC_OK) return rc;
        }
        if (total) {
            if (k + 1 < h->K) {
                int rc = fetch_row(h->d_tot, h->d_tot16, h->d_tot_hi, k, total + (int64_t)k * h->M);
                if (rc == AMC_OK) rc = add_base(h->d_tot_base, k, total + (int64_t)k * h->M);
                if (rc != AMC_OK) return rc;
            } else {
                // the last move: every chain has taken the same number of steps, its total_calls is what the other moves left
                for (int64_t c = 0; c < h->M; ++c) {
                    int64_t others = 0;
                    for (int j = 0; j + 1 < h->K; ++j) others += total[(int64_t)j * h->M + c];
                    total[(int64_t)k * h->M + c] = (int64_t)(h->t_base + h->t_counted) - others;
                }
            }
        }
    }
    return AMC_OK;
}

// pool totals of the counter ARRAYS (K > 1; without their 64-bit bases): host[k] accepted, host[AMC_MAX_MOVES + k] total of move k < K - 1
static int array_totals(amc_handle* h, unsigned long long (&host)[2 * AMC_MAX_MOVES])
{
    { const int rc = fold_log(h); if (rc != AMC_OK) return rc; }
    if (h->K > 1) {
        AMC_HIP(hipMemsetAsync(h->d_totals, 0, 2 * AMC_MAX_MOVES * sizeof(unsigned long long), h->stream));
        if (h->narrow)
            hipLaunchKernelGGL(amc::counter_totals_kernel<uint16_t>, dim3(grid_for(h, (h->M + 3) / 4)), dim3(AMC_BLOCK), 0, h->stream,
                               h->d_acc16, h->d_tot16, h->use_high ? h->d_acc_hi : nullptr, h->use_high ? h->d_tot_hi : nullptr, h->M,
                               h->M_pad, h->K, h->d_totals, h->d_totals + AMC_MAX_MOVES);
        else
            hipLaunchKernelGGL(amc::counter_totals_kernel<uint32_t>, dim3(grid_for(h, (h->M + 3) / 4)), dim3(AMC_BLOCK), 0, h->stream,
                               h->d_acc, h->d_tot, (const uint16_t*)nullptr, (const uint16_t*)nullptr, h->M, h->M_pad, h->K, h->d_totals,
                               h->d_totals + AMC_MAX_MOVES);
        AMC_HIP(hipGetLastError());
    }
    AMC_HIP(hipMemcpyAsync(host, h->d_totals, sizeof(host), hipMemcpyDeviceToHost, h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));
    return AMC_OK;
}

int amc_counter_totals(amc_handle* h, int64_t* accepted, int64_t* total)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_counter_totals: NULL handle");
    AMC_HIP(hipSetDevice(h->device));
    unsigned long long host[2 * AMC_MAX_MOVES];
    { const int rc = array_totals(h, host); if (rc != AMC_OK) return rc; }
    if (h->K == 1) {
        unsigned long long acc = 0;
        const int rc = sum_acc_slots(h, &acc);         // (the pool-wide slots are 64-bit and never carried)
        if (rc != AMC_OK) return rc;
        host[0] = acc;
    }
    unsigned long long others = 0;
    for (int k = 0; k < h->K; ++k) {
        if (accepted) accepted[k] = (int64_t)(host[k] + (h->K > 1 ? h->base_acc_total[k] : 0ull));
        // the last move's total: all counted steps of all chains minus the other moves' (its per-chain array does not exist)
        const unsigned long long tk = (k + 1 < h->K) ? host[AMC_MAX_MOVES + k] + h->base_tot_total[k]
                                                     : (h->t_base + h->t_counted) * (uint64_t)h->M - others;
        others += tk;
        if (total) total[k] = (int64_t)tk;
    }
    return AMC_OK;
}

// Carries the 32-bit counter arrays into their 64-bit bases and restarts them at zero (see counter_rebase_kernel): pending log
// rows are folded first, the pool totals of what is carried are kept on the host (amc_counter_totals), and a handle with u16
// planes goes on with u32 arrays -- the same bytes per counter, and the pass that forms the acceptance ratios from arrays plus
// bases (reduce_kernel) reads those.
static int counter_rebase(amc_handle* h)
{
    unsigned long long host[2 * AMC_MAX_MOVES];
    { const int rc = array_totals(h, host); if (rc != AMC_OK) return rc; }         // folds the log
    const size_t n = (size_t)h->K * (size_t)h->M_pad, nt = (size_t)(h->K - 1) * (size_t)h->M_pad;
    if (!h->d_acc_base) {
        AMC_HIP(hipMalloc(&h->d_acc_base, n * sizeof(unsigned long long)));
        AMC_HIP(hipMemsetAsync(h->d_acc_base, 0, n * sizeof(unsigned long long), h->stream));
        if (nt) {
            AMC_HIP(hipMalloc(&h->d_tot_base, nt * sizeof(unsigned long long)));
            AMC_HIP(hipMemsetAsync(h->d_tot_base, 0, nt * sizeof(unsigned long long), h->stream));
        }
    }
    const int grid = grid_for(h, (int64_t)n);
    if (h->narrow) {
        hipLaunchKernelGGL(amc::counter_rebase_kernel<uint16_t>, dim3(grid), dim3(AMC_BLOCK), 0, h->stream, h->d_acc16,
                           h->use_high ? h->d_acc_hi : nullptr, (int64_t)n, h->d_acc_base);
        if (nt) hipLaunchKernelGGL(amc::counter_rebase_kernel<uint16_t>, dim3(grid), dim3(AMC_BLOCK), 0, h->stream, h->d_tot16,
                                   h->use_high ? h->d_tot_hi : nullptr, (int64_t)nt, h->d_tot_base);
    } else {
        hipLaunchKernelGGL(amc::counter_rebase_kernel<uint32_t>, dim3(grid), dim3(AMC_BLOCK), 0, h->stream, h->d_acc, (uint16_t*)nullptr,
                           (int64_t)n, h->d_acc_base);
        if (nt) hipLaunchKernelGGL(amc::counter_rebase_kernel<uint32_t>, dim3(grid), dim3(AMC_BLOCK), 0, h->stream, h->d_tot,
                                   (uint16_t*)nullptr, (int64_t)nt, h->d_tot_base);
    }
    AMC_HIP(hipGetLastError());
    if (h->narrow) {                 // u32 arrays from here on
        AMC_HIP(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_acc16); (void)hipFree(h->d_tot16); (void)hipFree(h->d_acc_hi); (void)hipFree(h->d_tot_hi);
        h->d_acc16 = h->d_tot16 = h->d_acc_hi = h->d_tot_hi = nullptr;
        const hipError_t e = alloc_counters(h, false);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? AMC_ERR_OOM : AMC_ERR_HIP, "counter_rebase: %s", hipGetErrorString(e));
    }
    if (h->K > 1)
        for (int k = 0; k < h->K; ++k) {
            h->base_acc_total[k] += host[k];
            if (k + 1 < h->K) h->base_tot_total[k] += host[AMC_MAX_MOVES + k];
        }
    h->t_base += h->t_counted;
    h->t_counted = 0;
    h->use_high = false;
    return AMC_OK;
}

// Move.accepted_calls / total_calls are Int (Int64) in the reference (src/metropolis.jl:145-146); the per-chain arrays on
// the device count in 32 bits.  No chain's counter can exceed the number of counted steps, so before the launch that would take
// that number past 2^32 - 1 the arrays are carried into 64-bit bases (counter_rebase) and the count goes on -- round 5; until
// round 4 that call was refused.  `steps`: what the next LAUNCH counts (at most 2^20).  The pool-wide counter of a K = 1 handle
// without per-chain counters is 64-bit anyway.
// Handles with u16 planes bring the high planes into play here, before the call that would count past 65 535 steps (rows
// still waiting in the log are then folded by the carrying form as well: it starts from high halves that are zero).
extern "C++" int counter_room(amc_handle* h, const char* who, uint64_t steps)
{
    (void)who;
    if (!h->counters) return AMC_OK;
    if (h->t_counted + steps > 0xFFFFFFFFull) {
        const int rc = counter_rebase(h);
        if (rc != AMC_OK) return rc;
    }
    if (h->narrow && h->t_counted + steps > 0xFFFFull) h->use_high = true;
    return AMC_OK;
}

// The form of the rows a launch of `grid` blocks leaves its callback sums in (amc::red_finish): the compact 64-byte row while a
// lane adds at most RED_COMPACT_TRIPS summands per column -- one per trip --, the wide one beyond.
extern "C++" int red_row_stride(const amc_handle* h, int grid)
{
    const int64_t pairs = (h->M + 1) / 2, lanes = (int64_t)grid * AMC_BLOCK;
    return (!h->wide_red_rows && (pairs + lanes - 1) / lanes <= amc::RED_COMPACT_TRIPS) ? (int)amc::RED_COMPACT_WORDS : RED_HOST_STRIDE;
}

extern "C++" amc::SweepArgs make_sweep_args(const amc_handle* h, int32_t n_steps)
{
    amc::SweepArgs a;
    a.x = h->d_x;
    a.beta_arr = h->beta_arr ? h->d_beta : nullptr;
    a.log = h->d_log;
    a.log_pos = h->log_fill;
    a.ptab = h->d_ptab;
    a.pick_tab = h->d_pick;
    a.acc_total = h->d_acc_slots;
    a.n_chains = h->M;
    a.m_stride = h->M_pad;
    a.pair0 = (uint64_t)h->offset >> 1;
    a.t0 = h->t;
    a.n_steps = n_steps;
    a.n_moves = h->K;
    a.key0 = (uint32_t)h->seed;
    a.key1 = (uint32_t)(h->seed >> 32);
    a.beta = h->beta;
    a.red_partials = h->red[(h->red_head + h->red_count) % RED_TICKETS].h_rows;   // the ticket a REDUCE launch would fill
    a.red_stride = RED_HOST_STRIDE;
    a.red_cols = h->red_cols;
    a.exact_accept = h->exact_accept ? 1 : 0;
    a.n_slots = h->n_slots;
    return a;
}

// the grid of the sweep launch that also forms the callback sums (sweep_impl)
static int reduce_sweep_grid(const amc_handle* h)
{
    return grid_for(h, (h->M + 1) / 2, h->blocks_per_cu_pg ? h->blocks_per_cu_pg : h->blocks_per_cu_red);
}

// n_sweeps x sweepstep MH steps in launches of at most 2^20 steps; when fuse_reduce is set (streamed form
// only) the LAST launch also leaves the callback partial sums of the final state in h_partials[grid][8].
extern "C++" int sweep_impl(amc_handle* h, int64_t n_sweeps, bool fuse_reduce, int* grid_out)
{
    AMC_HIP(hipSetDevice(h->device));
    { const int rc = pg_resolve(h); if (rc != AMC_OK) return rc; }      // the sweep kernels read sigma from the parameter table
    int64_t remaining = n_sweeps * (int64_t)h->sweepstep;
    // one grid for the whole call (the caller of a fused reduction sums `grid` rows)
    // (a call whose last launch also forms the callback sums: that form holds 5 blocks per CU -- 89 VGPRs -- and runs one round of
    // them, 49 -> 45 us per callback at K = 2 and 1e7 chains; plain sweeps are indifferent between 5 and 8)
    const int grid = fuse_reduce ? reduce_sweep_grid(h) : grid_for(h, (h->M + 1) / 2, remaining == 1 ? h->blocks_per_cu_single : 0);
    if (fuse_reduce && grid > h->n_slots) return fail(AMC_ERR_STATE, "sweep_impl: a grid of %d blocks has no rows to leave its callback sums in (%d)", grid, h->n_slots);
    if (grid_out) *grid_out = grid;
    while (remaining > 0) {
        int32_t chunk = remaining > (1 << 20) ? (1 << 20) : (int32_t)remaining;
        if (h->d_log) {      // per-chain counters: one log row per MH step; a full log is folded before it is reused
            int room = 0;
            const int rc = log_room(h, &room);
            if (rc != AMC_OK) return rc;
            if (chunk > room) chunk = room;
        }
        { const int rc = counter_room(h, "amc_sweep", (uint64_t)chunk); if (rc != AMC_OK) return rc; }      // (may carry the counters: arrays restart at zero)
        amc::SweepArgs a = make_sweep_args(h, chunk);
        a.red_stride = red_row_stride(h, grid);
        const bool last = remaining == chunk;
        int rc;
        if (h->use_rtc)
            rc = launch_sweep_custom(h, a, grid, fuse_reduce && last);
        else if (fuse_reduce && last)
            rc = (h->potential == AMC_POTENTIAL_DOUBLE_WELL) ? launch_sweep_reduce<amc::POT_DOUBLE_WELL>(h, a, grid)
                                                             : launch_sweep_reduce<amc::POT_HARMONIC>(h, a, grid);
        else
            rc = (h->potential == AMC_POTENTIAL_DOUBLE_WELL) ? launch_sweep<amc::POT_DOUBLE_WELL>(h, a, grid)
                                                             : launch_sweep<amc::POT_HARMONIC>(h, a, grid);
        if (rc != AMC_OK) return rc;
        h->t += (uint64_t)chunk;
        h->t_counted += (uint64_t)chunk;
        if (h->d_log) h->log_fill += chunk;
        remaining -= chunk;
    }
    return AMC_OK;
}

int amc_sweep(amc_handle* h, int64_t n_sweeps)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_sweep: NULL handle");
    if (n_sweeps < 0) return fail(AMC_ERR_BAD_ARG, "amc_sweep: n_sweeps < 0");
    if (n_sweeps == 0) return AMC_OK;
    return sweep_impl(h, n_sweeps, false, nullptr);
}

int amc_sweep_launches(amc_handle* h, int64_t n_launches)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_sweep_launches: NULL handle");
    if (n_launches < 0) return fail(AMC_ERR_BAD_ARG, "amc_sweep_launches: n_launches < 0");
    for (int64_t i = 0; i < n_launches; ++i) {
        const int rc = sweep_impl(h, 1, false, nullptr);
        if (rc != AMC_OK) return rc;
    }
    return AMC_OK;
}

int amc_upload_counters(amc_handle* h, const int64_t* accepted, const int64_t* total)
{
    if (!h || !accepted) return fail(AMC_ERR_BAD_ARG, "amc_upload_counters: NULL argument");
    if (!h->counters)
        return fail(AMC_ERR_STATE, "amc_upload_counters: handle was created with per_chain_counters = 0 "
                                   "(use amc_set_counter_totals)");
    if (h->K > 1 && !total) return fail(AMC_ERR_BAD_ARG, "amc_upload_counters: total is required when K > 1");
    // Every chain takes the same number of MH steps (mc_sweep!, metropolis.jl:205-210), so sum_k total_calls_ck is ONE number
    // for all chains: the count of steps taken.  The device keeps that number and K - 1 of the K total arrays.
    const int64_t LIMIT = (int64_t)1 << 52;          // counts are divided as Float64s (callback_acceptance): exact below 2^53
    uint64_t steps = h->t_base + h->t_counted;
    if (total) {
        for (int64_t c = 0; c < h->M; ++c) {
            int64_t sum = 0;
            for (int k = 0; k < h->K; ++k) {
                const int64_t v = total[(int64_t)k * h->M + c];
                if (v < 0 || v > LIMIT) return fail(AMC_ERR_BAD_ARG, "amc_upload_counters: counter out of range [0, 2^52]");
                sum += v;
            }
            if (c == 0) steps = (uint64_t)sum;
            else if ((uint64_t)sum != steps)
                return fail(AMC_ERR_BAD_ARG, "amc_upload_counters: the total_calls of a chain must add up to the same step count on "
                                             "every chain (chain 0: %llu, chain %lld: %lld)", (unsigned long long)steps, (long long)c, (long long)sum);
        }
        if (steps > (uint64_t)LIMIT) return fail(AMC_ERR_BAD_ARG, "amc_upload_counters: step count out of range [0, 2^52]");
    }
    int64_t acc_max = 0;
    for (int64_t i = 0; i < (int64_t)h->K * h->M; ++i) {
        if (accepted[i] < 0 || accepted[i] > LIMIT) return fail(AMC_ERR_BAD_ARG, "amc_upload_counters: counter out of range [0, 2^52]");
        acc_max = std::max(acc_max, accepted[i]);
    }
    AMC_HIP(hipSetDevice(h->device));
    if (steps > 0xFFFFFFFFull || (uint64_t)acc_max > 0xFFFFFFFFull || h->d_acc_base) {
        // counts beyond 32 bits (or a handle that has carried before): everything goes into the 64-bit bases, the arrays restart
        // at zero (counter_rebase does the allocating and the switch to u32 arrays; what it carries is overwritten next)
        h->log_fill = 0;
        { const int rc = counter_rebase(h); if (rc != AMC_OK) return rc; }
        std::vector<unsigned long long> b((size_t)h->M);
        for (int k = 0; k < h->K; ++k) {
            h->base_acc_total[k] = h->base_tot_total[k] = 0ull;
            for (int pass = 0; pass < 2; ++pass) {
                const int64_t* src = pass == 0 ? accepted : total;
                if (!src || (pass == 1 && k + 1 == h->K)) continue;
                unsigned long long sum = 0;
                for (int64_t c = 0; c < h->M; ++c) { b[(size_t)c] = (unsigned long long)src[(int64_t)k * h->M + c]; sum += b[(size_t)c]; }
                (pass == 0 ? h->base_acc_total[k] : h->base_tot_total[k]) = sum;
                unsigned long long* dst = (pass == 0 ? h->d_acc_base : h->d_tot_base) + (size_t)k * h->M_pad;
                AMC_HIP(hipMemcpyAsync(dst, b.data(), (size_t)h->M * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
                AMC_HIP(hipStreamSynchronize(h->stream));
            }
        }
        if (h->K == 1) {
            const unsigned long long acc_sum = h->base_acc_total[0];
            AMC_HIP(hipMemsetAsync(h->d_acc_slots, 0, (size_t)h->n_slots * sizeof(unsigned long long), h->stream));
            AMC_HIP(hipMemcpyAsync(h->d_acc_slots, &acc_sum, sizeof(acc_sum), hipMemcpyHostToDevice, h->stream));
            AMC_HIP(hipStreamSynchronize(h->stream));
        }
        h->t_base = steps;
        h->t_counted = 0;
        return AMC_OK;
    }
    // (the handle's own bookkeeping -- log_fill, use_high, t_counted -- changes only once every plane has been copied: a copy
    // that fails leaves the handle counting as before)
    std::vector<uint32_t> buf((size_t)h->M);
    unsigned long long acc_sum = 0;
    for (int k = 0; k < h->K; ++k) {
        for (int pass = 0; pass < 2; ++pass) {
            const int64_t* src = pass == 0 ? accepted : total;
            if (!src || (pass == 1 && k + 1 == h->K)) continue;             // the last move's totals have no array
            uint16_t* b16 = reinterpret_cast<uint16_t*>(buf.data());
            for (int64_t c = 0; c < h->M; ++c) {
                const int64_t v = src[(int64_t)k * h->M + c];
                if (h->narrow) b16[(size_t)c] = (uint16_t)(v & 0xFFFF); else buf[(size_t)c] = (uint32_t)v;
                if (pass == 0) acc_sum += (unsigned long long)v;
            }
            if (h->narrow) {
                uint16_t* dst = (pass == 0 ? h->d_acc16 : h->d_tot16) + (size_t)k * h->M_pad;
                AMC_HIP(hipMemcpyAsync(dst, b16, (size_t)h->M * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
                AMC_HIP(hipStreamSynchronize(h->stream));
                for (int64_t c = 0; c < h->M; ++c) b16[(size_t)c] = (uint16_t)(src[(int64_t)k * h->M + c] >> 16);
                dst = (pass == 0 ? h->d_acc_hi : h->d_tot_hi) + (size_t)k * h->M_pad;
                AMC_HIP(hipMemcpyAsync(dst, b16, (size_t)h->M * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
            } else {
                uint32_t* dst = (pass == 0 ? h->d_acc : h->d_tot) + (size_t)k * h->M_pad;
                AMC_HIP(hipMemcpyAsync(dst, buf.data(), (size_t)h->M * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
            }
            AMC_HIP(hipStreamSynchronize(h->stream));
        }
    }
    if (h->K == 1) {
        AMC_HIP(hipMemsetAsync(h->d_acc_slots, 0, (size_t)h->n_slots * sizeof(unsigned long long), h->stream));
        AMC_HIP(hipMemcpyAsync(h->d_acc_slots, &acc_sum, sizeof(acc_sum), hipMemcpyHostToDevice, h->stream));
        AMC_HIP(hipStreamSynchronize(h->stream));
    }
    h->log_fill = 0;            // every counter is replaced: steps still waiting in the log are dropped with the old values
    // u16 planes: the high halves take part from now on unless no counter can have reached 2^16 (see counter_room); both
    // planes are always written, so that halves which do not take part yet are zero when they do
    if (h->narrow) h->use_high = steps > 0xFFFFull || (uint64_t)acc_max > steps;
    h->t_counted = steps;
    return AMC_OK;
}

int amc_set_counter_totals(amc_handle* h, const int64_t* accepted, uint64_t steps_counted)
{
    if (!h || !accepted) return fail(AMC_ERR_BAD_ARG, "amc_set_counter_totals: NULL argument");
    if (h->K != 1 || h->counters)
        return fail(AMC_ERR_STATE, "amc_set_counter_totals: only for K = 1 handles without per-chain counters");
    if (accepted[0] < 0) return fail(AMC_ERR_BAD_ARG, "amc_set_counter_totals: negative count");
    AMC_HIP(hipSetDevice(h->device));
    const unsigned long long acc = (unsigned long long)accepted[0];
    AMC_HIP(hipMemsetAsync(h->d_acc_slots, 0, (size_t)h->n_slots * sizeof(unsigned long long), h->stream));
    AMC_HIP(hipMemcpyAsync(h->d_acc_slots, &acc, sizeof(acc), hipMemcpyHostToDevice, h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));
    h->t_counted = steps_counted;
    return AMC_OK;
}

// Every block ends with one 64-bit atomic per non-empty bin on the SAME few hundred addresses, and those serialise (~13 ns
// each per address): a full grid of 2048 blocks spends 27 us there.  Two blocks per CU keep enough loads in flight and the
// flush short (1e7 chains, 200 bins: 53.1 us with 2048 blocks, 32.2 with 1024, 23.4 with 512, 27.0 with 256, 43.5 with 128).
static int hist_grid(const amc_handle* h)
{
    const int g = 2 * h->n_cu;
    return g < h->red_blocks ? g : h->red_blocks;
}

int amc_histogram(amc_handle* h, double lo, double hi, int n_bins, uint64_t* counts)
{
    if (!h || !counts) return fail(AMC_ERR_BAD_ARG, "amc_histogram: NULL argument");
    if (n_bins < 1 || n_bins > 8192 || !(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi))
        return fail(AMC_ERR_BAD_ARG, "amc_histogram: need 1 <= n_bins <= 8192 and finite lo < hi");
    AMC_HIP(hipSetDevice(h->device));
    unsigned long long* d_counts = nullptr;
    const size_t bytes = (size_t)(n_bins + 3) * sizeof(unsigned long long);
    AMC_HIP(hipMalloc(&d_counts, bytes));
    AMC_HIP(hipMemsetAsync(d_counts, 0, bytes, h->stream));
    const double inv_w = (double)n_bins / (hi - lo);
    const double* d_pos = nullptr;
    { const int rc = positions_f64(h, &d_pos); if (rc != AMC_OK) { (void)hipFree(d_counts); return rc; } }
    hipLaunchKernelGGL(amc::histogram_kernel, dim3(hist_grid(h)), dim3(AMC_BLOCK), (size_t)(n_bins + 3) * sizeof(unsigned int),
                       h->stream, d_pos, h->M, lo, hi, inv_w, n_bins, d_counts);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(counts, d_counts, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_counts);
    if (e != hipSuccess) return fail(AMC_ERR_HIP, "amc_histogram: %s", hipGetErrorString(e));
    return AMC_OK;
}

int amc_histogram_accumulate(amc_handle* h, double lo, double hi, int n_bins)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_histogram_accumulate: NULL handle");
    if (n_bins < 1 || n_bins > 8192 || !(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi))
        return fail(AMC_ERR_BAD_ARG, "amc_histogram_accumulate: need 1 <= n_bins <= 8192 and finite lo < hi");
    AMC_HIP(hipSetDevice(h->device));
    if (h->d_hist && (n_bins != h->hist_bins || lo != h->hist_lo || hi != h->hist_hi))
        return fail(AMC_ERR_STATE, "amc_histogram_accumulate: the running histogram has other bins (fetch it with reset first)");
    if (!h->d_hist) {
        const size_t bytes = (size_t)(n_bins + 3) * sizeof(unsigned long long);
        AMC_HIP(hipMalloc(&h->d_hist, bytes));
        AMC_HIP(hipMemsetAsync(h->d_hist, 0, bytes, h->stream));
        h->hist_bins = n_bins; h->hist_lo = lo; h->hist_hi = hi;
    }
    const double inv_w = (double)n_bins / (hi - lo);
    const double* d_pos = nullptr;
    { const int rc = positions_f64(h, &d_pos); if (rc != AMC_OK) return rc; }
    hipLaunchKernelGGL(amc::histogram_kernel, dim3(hist_grid(h)), dim3(AMC_BLOCK), (size_t)(n_bins + 3) * sizeof(unsigned int),
                       h->stream, d_pos, h->M, lo, hi, inv_w, n_bins, h->d_hist);
    AMC_HIP(hipGetLastError());
    return AMC_OK;
}

int amc_histogram_fetch(amc_handle* h, uint64_t* counts, int n_bins, int reset)
{
    if (!h || !counts) return fail(AMC_ERR_BAD_ARG, "amc_histogram_fetch: NULL argument");
    if (!h->d_hist) return fail(AMC_ERR_STATE, "amc_histogram_fetch: nothing has been accumulated");
    if (n_bins != h->hist_bins) return fail(AMC_ERR_BAD_ARG, "amc_histogram_fetch: the running histogram has %d bins", h->hist_bins);
    AMC_HIP(hipSetDevice(h->device));
    const size_t bytes = (size_t)(n_bins + 3) * sizeof(unsigned long long);
    AMC_HIP(hipMemcpyAsync(counts, h->d_hist, bytes, hipMemcpyDeviceToHost, h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));
    if (reset) {
        (void)hipFree(h->d_hist);
        h->d_hist = nullptr;
        h->hist_bins = 0;
    }
    return AMC_OK;
}

int amc_download_strided(amc_handle* h, int64_t first, int64_t stride, int64_t count, double* x)
{
    if (!h || !x) return fail(AMC_ERR_BAD_ARG, "amc_download_strided: NULL argument");
    // (no product that could overflow: count - 1 <= (M - 1 - first) / stride)
    if (first < 0 || stride < 1 || count < 0 || (count > 0 && (first >= h->M || count - 1 > (h->M - 1 - first) / stride)))
        return fail(AMC_ERR_BAD_ARG, "amc_download_strided: range [first + i*stride] leaves the local shard");
    if (count == 0) return AMC_OK;
    AMC_HIP(hipSetDevice(h->device));
    double* d_out = nullptr;
    AMC_HIP(hipMalloc(&d_out, (size_t)count * sizeof(double)));
    const double* d_pos = nullptr;
    { const int rc = positions_f64(h, &d_pos); if (rc != AMC_OK) { (void)hipFree(d_out); return rc; } }
    hipLaunchKernelGGL(amc::gather_strided_kernel, dim3(grid_for(h, count)), dim3(AMC_BLOCK), 0, h->stream, d_pos, first,
                       stride, count, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(x, d_out, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(AMC_ERR_HIP, "amc_download_strided: %s", hipGetErrorString(e));
    return AMC_OK;
}

int amc_get_estimator_step(amc_handle* h, uint64_t* t)
{
    if (!h || !t) return fail(AMC_ERR_BAD_ARG, "amc_get_estimator_step: NULL argument");
    *t = h->t_est;
    return AMC_OK;
}

int amc_set_estimator_step(amc_handle* h, uint64_t t)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_set_estimator_step: NULL handle");
    if (t >> 48) return fail(AMC_ERR_BAD_ARG, "amc_set_estimator_step: call index must fit 48 bits");
    h->t_est = t;
    return AMC_OK;
}

int amc_get_step(amc_handle* h, uint64_t* t)
{
    if (!h || !t) return fail(AMC_ERR_BAD_ARG, "amc_get_step: NULL argument");
    *t = h->t;
    return AMC_OK;
}

int amc_set_step(amc_handle* h, uint64_t t)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_set_step: NULL handle");
    if (t >> 48) return fail(AMC_ERR_BAD_ARG, "amc_set_step: step index must fit 48 bits");
    h->t = t;
    return AMC_OK;
}

// The ticket a new reduction fills (tickets complete in the order they were begun), or nullptr when RED_TICKETS are in flight.
extern "C++" RedTicket* red_next(amc_handle* h) { return h->red_count == RED_TICKETS ? nullptr : &h->red[(h->red_head + h->red_count) % RED_TICKETS]; }

static int red_commit(amc_handle* h, RedTicket* t, int rows)
{
    AMC_HIP(hipEventRecord(t->ev, h->stream));
    t->pending = true;
    t->rows = rows;
    t->row_stride = red_row_stride(h, rows);       // (rows = the grid of the launch that wrote them)
    t->cols = h->red_cols;
    t->t_counted = h->t_base + h->t_counted;
    h->red_count += 1;
    return AMC_OK;
}

int amc_reduce_begin(amc_handle* h)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_reduce_begin: NULL handle");
    RedTicket* t = red_next(h);
    if (!t) return fail(AMC_ERR_STATE, "amc_reduce_begin: %d reductions are already in flight (call amc_reduce_end)", RED_TICKETS);
    AMC_HIP(hipSetDevice(h->device));
    int ratio_mode = (h->K > 1) ? 2 : (h->counters ? 1 : 0);
    t->ratio_rows = 0;
    t->ratio_acc = false;
    if (ratio_mode != 0 && h->K <= 4 && !h->d_acc_base) {
        // per-chain counters, few moves: the fold of the step log forms the acceptance-ratio sums while the counters
        // are in its registers (rows in h_ratio); the pass below then reads x only
        const int rc = fold_log(h, true, &t->ratio_rows, t->h_ratio);
        if (rc != AMC_OK) return rc;
        ratio_mode = 0;
    } else if (ratio_mode != 0) {
        const int rc = fold_log(h);
        if (rc != AMC_OK) return rc;
        AMC_HIP(hipMemsetAsync(t->d_ratio_acc, 0, (size_t)AMC_MAX_MOVES * 3 * sizeof(unsigned long long), h->stream));
        t->ratio_acc = true;
    }
    // The blocks store their rows straight into pinned, device-mapped host memory and the HOST adds them up in
    // amc_reduce_end (integers: amc_xsum.h) -- no final-pass launches (~5 us each even when empty) and no D2H copy in
    // stream order (which would hold the next sweep back for a copy-engine round trip).
    amc::xs_word* rows = t->h_rows;
    const int stride = red_row_stride(h, h->red_blocks);
    int cols = h->red_cols;
    const unsigned long long* slots = (ratio_mode == 0 && t->ratio_rows == 0) ? h->d_acc_slots : nullptr;
    unsigned long long* racc = t->d_ratio_acc;
    if (h->use_rtc) {
        const double* d_x = h->d_x;
        const uint32_t *d_acc = h->d_acc, *d_tot = h->d_tot;
        int64_t m = h->M, m_pad = h->M_pad;
        int k = h->K, mode = ratio_mode, st = stride, n_slots = h->n_slots;
        uint64_t t_counted = h->t_counted;
        const unsigned long long *acc_base = h->d_acc_base, *tot_base = h->d_tot_base;
        uint64_t t_base = h->t_base;
        void* params[] = {&d_x, &d_acc, &d_tot, &m, &m_pad, &k, &mode, &t_counted, &rows, &st, &slots, &n_slots, &racc, &cols, &acc_base, &tot_base, &t_base};
        const int rc = rtc_launch(h, "amc::reduce_kernel<" + std::to_string(h->potential) + ">", h->red_blocks, params);
        if (rc != AMC_OK) return rc;
    } else if (h->potential == AMC_POTENTIAL_DOUBLE_WELL)
        hipLaunchKernelGGL(amc::reduce_kernel<amc::POT_DOUBLE_WELL>, dim3(h->red_blocks), dim3(AMC_BLOCK), 0, h->stream,
                           h->d_x, h->d_acc, h->d_tot, h->M, h->M_pad, h->K, ratio_mode, h->t_counted, rows, stride, slots,
                           h->n_slots, racc, cols, h->d_acc_base, h->d_tot_base, h->t_base);
    else
        hipLaunchKernelGGL(amc::reduce_kernel<amc::POT_HARMONIC>, dim3(h->red_blocks), dim3(AMC_BLOCK), 0, h->stream,
                           h->d_x, h->d_acc, h->d_tot, h->M, h->M_pad, h->K, ratio_mode, h->t_counted, rows, stride, slots,
                           h->n_slots, racc, cols, h->d_acc_base, h->d_tot_base, h->t_base);
    AMC_HIP(hipGetLastError());
    if (t->ratio_acc)
        AMC_HIP(hipMemcpyAsync(t->h_ratio_acc, t->d_ratio_acc, (size_t)h->K * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                               h->stream));
    return red_commit(h, t, h->red_blocks);
}

// Second half of a reduction whose sums over x were formed by the launch that has just been queued (rows in the next ticket's
// h_rows[grid][RED_HOST_STRIDE], make_sweep_args): with per-chain counters the fold of the step log (pending rows incl. that
// launch's) forms the ratio sums -- no pass re-reads x or the counters.
extern "C++" int finish_fused_reduce(amc_handle* h, int grid)
{
    RedTicket* t = red_next(h);
    if (!t) return fail(AMC_ERR_STATE, "finish_fused_reduce: no free reduction ticket");
    t->ratio_rows = 0;
    t->ratio_acc = false;
    if (h->counters) {
        const int rc2 = fold_log(h, true, &t->ratio_rows, t->h_ratio);
        if (rc2 != AMC_OK) return rc2;
    }
    return red_commit(h, t, grid);
}

// A launch that forms the callback sums adds ONE summand per trip and column (a chain pair's sum) into each lane's accumulators,
// and those hold XS_LANE_CAP of them (amc_xsum.h); a launch of `grid` blocks makes ceil(pairs / (grid 256)) trips per lane.
// Beyond that (ensembles of more than 2e9 chains) the sums are formed by the pass of their own, which flushes as it goes.
extern "C++" bool reduce_fits_in_grid(const amc_handle* h, int grid)
{
    const int64_t pairs = (h->M + 1) / 2, lanes = (int64_t)grid * AMC_BLOCK;
    return (pairs + lanes - 1) / lanes <= amc::xs::XS_LANE_CAP - 2;
}
int amc_sweep_reduce_begin(amc_handle* h, int64_t n_sweeps)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_sweep_reduce_begin: NULL handle");
    if (n_sweeps < 1) return fail(AMC_ERR_BAD_ARG, "amc_sweep_reduce_begin: n_sweeps must be >= 1");
    if (!red_next(h))
        return fail(AMC_ERR_STATE, "amc_sweep_reduce_begin: %d reductions are already in flight (call amc_reduce_end)", RED_TICKETS);
    // the ratio sums need the counters of every move (K > 4), or arrays plus their 64-bit bases (a handle that has counted past
    // 2^32 steps): sweep, then the reduction pass
    if (h->K > 4 || h->d_acc_base || !reduce_fits_in_grid(h, reduce_sweep_grid(h))) {
        const int rc = sweep_impl(h, n_sweeps, false, nullptr);
        return rc != AMC_OK ? rc : amc_reduce_begin(h);
    }
    int grid = 0;
    const int rc = sweep_impl(h, n_sweeps, true, &grid);     // the last launch wrote the sums over x to the ticket's rows
    if (rc != AMC_OK) return rc;
    return finish_fused_reduce(h, grid);
}

// Finishes the OLDEST reduction in flight: its columns as records (amc_xsum.h): AMC_RED_HEADER + K of them.
static int reduce_end_records(amc_handle* h, const char* who, double* recs, uint64_t* steps_counted)
{
    if (h->red_count == 0) return fail(AMC_ERR_STATE, "%s: no reduction in flight (call amc_reduce_begin)", who);
    AMC_HIP(hipSetDevice(h->device));
    RedTicket* t = &h->red[h->red_head];
    AMC_HIP(wait_event(t->ev));                  // waits for that reduction only, not for work queued after it
    t->pending = false;
    h->red_head = (h->red_head + 1) % RED_TICKETS;
    h->red_count -= 1;
    namespace xs = amc::xs;
    xs::PartR col[amc::RED_COLS];
    for (int c = 0; c < amc::RED_COLS; ++c) col[c] = xs::part_r_empty();
    double slot_total = 0.0;
    const bool compact = t->row_stride == amc::RED_COMPACT_WORDS;
    const bool with_slot = h->K == 1 && !h->counters;       // the row's last word is written by those launches only
    for (int r = 0; r < t->rows; ++r) {
        const amc::xs_word* row = t->h_rows + (size_t)r * t->row_stride;
        for (int c = 0; c < amc::RED_COLS; ++c)
            xs::part_r_merge(col[c], compact ? amc::xs_load_compact_row(row, c) : amc::xs_load_r_row(row + c * amc::XS_ROW_R));
        if (with_slot) {
            double v;
            std::memcpy(&v, row + (compact ? (int)amc::RED_COMPACT_SLOT : (int)amc::RED_ROW_SLOT), sizeof(double));
            slot_total += v;                                // integers: exact in any order
        }
    }
    // a sum nobody asked for (amc_set_reduce_columns) was not formed: its record stays empty
    static const int want[amc::RED_COLS] = {amc::RED_WANT_E, amc::RED_WANT_X, amc::RED_WANT_XX};
    for (int c = 0; c < amc::RED_COLS; ++c) {
        if (t->cols & want[c]) xs::rec_from_r(recs + (size_t)c * xs::XS_WORDS, col[c]);
        else xs::rec_clear(recs + (size_t)c * xs::XS_WORDS);
    }
    // the rows of a reduction cover the handle's chains
    xs::rec_from_plain(recs + (size_t)AMC_RED_COUNT * xs::XS_WORDS, (double)h->M);
    for (int k = 0; k < h->K; ++k) {
        xs::PartQ q = xs::PartQ{xs::i128{0, 0}, 0u};
        int e = xs::XS_E_RATIO;
        if (t->ratio_rows > 0) {
            for (int r = 0; r < t->ratio_rows; ++r) {
                const xs::PartQ b = amc::xs_load_q_row(t->h_ratio + ((size_t)r * RATIO_STRIDE + k) * amc::XS_ROW_Q);
                q.k = xs::i128_add(q.k, b.k);
                q.flags |= b.flags;
            }
        } else if (t->ratio_acc) {
            const unsigned long long* a = t->h_ratio_acc + 3 * k;
            if (a[2] != 0) q.flags |= xs::XS_F_NAN;
            // low 32-bit halves and high parts were added separately: k = hi 2^32 + lo
            q.k = xs::i128_add(xs::i128_shl(xs::i128_of((long long)a[1]), 32), xs::i128{a[0], 0});
        } else {
            // K == 1 without per-chain counters: total_calls is the same on every chain, so sum_c accepted_c / total is
            // (sum_c accepted_c) / total up to rounding: the record is the pool-wide accepted TOTAL
            // (an integer, quantum 2^0); whoever rounds it divides by steps_counted
            q.k = xs::i128_of((long long)slot_total);
            e = 0;
        }
        xs::rec_from_q(recs + (size_t)(AMC_RED_HEADER + k) * xs::XS_WORDS, q, e);
    }
    if (steps_counted) *steps_counted = t->t_counted;
    return AMC_OK;
}

int amc_reduce_end_exact(amc_handle* h, double* records, uint64_t* steps_counted)
{
    if (!h || !records) return fail(AMC_ERR_BAD_ARG, "amc_reduce_end_exact: NULL argument");
    return reduce_end_records(h, "amc_reduce_end_exact", records, steps_counted);
}

int amc_reduce_end(amc_handle* h, double* out)
{
    if (!h || !out) return fail(AMC_ERR_BAD_ARG, "amc_reduce_end: NULL argument");
    double recs[(AMC_RED_HEADER + AMC_MAX_MOVES) * amc::xs::XS_WORDS];
    uint64_t steps = 0;
    const int rc = reduce_end_records(h, "amc_reduce_end", recs, &steps);
    if (rc != AMC_OK) return rc;
    for (int i = 0; i < AMC_RED_HEADER + h->K; ++i)
        out[i] = recs[(size_t)i * amc::xs::XS_WORDS] == (double)amc::xs::XS_EMPTY ? std::nan("") : amc::xs::rec_round(recs + (size_t)i * amc::xs::XS_WORDS);
    if (h->K == 1 && !h->counters) out[AMC_RED_SUM_RATIO0] = out[AMC_RED_SUM_RATIO0] / (double)steps;
    return AMC_OK;
}

int amc_reduce(amc_handle* h, double* out)
{
    if (!h || !out) return fail(AMC_ERR_BAD_ARG, "amc_reduce: NULL argument");
    if (h->red_count != 0) return fail(AMC_ERR_STATE, "amc_reduce: a reduction is in flight (call amc_reduce_end first)");
    const int rc = amc_reduce_begin(h);
    return rc != AMC_OK ? rc : amc_reduce_end(h, out);
}

int amc_set_reduce_columns(amc_handle* h, int columns)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_set_reduce_columns: NULL handle");
    if (columns < 0 || columns > AMC_REDUCE_ALL) return fail(AMC_ERR_BAD_ARG, "amc_set_reduce_columns: columns must be a combination of AMC_REDUCE_E / _X / _XX");
    h->red_cols = columns;
    return AMC_OK;
}

// Host-side arithmetic on records (no device involved): into[i] += from[i]; out[i] = the Float64 of records[i].
int amc_xsum_merge(double* into, const double* from, int n_records)
{
    if (!into || !from || n_records < 0) return fail(AMC_ERR_BAD_ARG, "amc_xsum_merge: bad argument");
    for (int i = 0; i < n_records; ++i) amc::xs::rec_merge(into + (size_t)i * amc::xs::XS_WORDS, from + (size_t)i * amc::xs::XS_WORDS);
    return AMC_OK;
}

int amc_xsum_round(const double* records, int n_records, double* out)
{
    if (!records || !out || n_records < 0) return fail(AMC_ERR_BAD_ARG, "amc_xsum_round: bad argument");
    for (int i = 0; i < n_records; ++i) out[i] = amc::xs::rec_round(records + (size_t)i * amc::xs::XS_WORDS);
    return AMC_OK;
}

// row of the parameter table that holds parameter p of every move
static int theta_row(int p) { return p == 0 ? (int)amc::PT_SIGMA : (int)amc::PT_THETA1 + p - 1; }

int amc_set_parameters(amc_handle* h, int k, const double* p, int n)
{
    if (!h || !p) return fail(AMC_ERR_BAD_ARG, "amc_set_parameters: NULL argument");
    if (k < 0 || k >= h->K) return fail(AMC_ERR_BAD_ARG, "amc_set_parameters: move index %d out of range", k);
    if (n != h->n_params)
        return fail(AMC_ERR_BAD_ARG, h->n_params == 1 ? "amc_set_parameters: StandardGaussian has exactly 1 parameter (sigma)"
                                                     : "amc_set_parameters: this handle's policy has %d parameters", h->n_params);
    if (h->n_params == 1) {
        if (!(p[0] >= 1e-100) || !(p[0] <= 1e100))
            return fail(AMC_ERR_BAD_ARG, "amc_set_parameters: sigma must lie in [1e-100, 1e100] (got %.17g)", p[0]);
    } else {
        for (int i = 0; i < n; ++i)
            if (!(p[i] - p[i] == 0.0)) return fail(AMC_ERR_BAD_ARG, "amc_set_parameters: parameter %d is not finite", i);
    }
    AMC_HIP(hipSetDevice(h->device));
    { const int rc = pg_resolve(h); if (rc != AMC_OK) return rc; }
    for (int i = 0; i < n; ++i)
        AMC_HIP(hipMemcpyAsync(h->d_ptab + theta_row(i) * AMC_MAX_MOVES + k, p + i, sizeof(double), hipMemcpyHostToDevice, h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));
    if (h->n_params == 1) {            // what derives from sigma (the script kernels of a policy with several parameters read none of it)
        hipLaunchKernelGGL(amc::prepare_params_kernel, dim3(1), dim3(64), 0, h->stream, h->d_ptab, h->K);
        AMC_HIP(hipGetLastError());
    }
    return AMC_OK;
}

int amc_get_parameters(amc_handle* h, int k, double* p, int n)
{
    if (!h || !p) return fail(AMC_ERR_BAD_ARG, "amc_get_parameters: NULL argument");
    if (k < 0 || k >= h->K) return fail(AMC_ERR_BAD_ARG, "amc_get_parameters: move index %d out of range", k);
    if (n != h->n_params)
        return fail(AMC_ERR_BAD_ARG, h->n_params == 1 ? "amc_get_parameters: StandardGaussian has exactly 1 parameter (sigma)"
                                                     : "amc_get_parameters: this handle's policy has %d parameters", h->n_params);
    AMC_HIP(hipSetDevice(h->device));
    { const int rc = pg_resolve(h); if (rc != AMC_OK) return rc; }
    for (int i = 0; i < n; ++i)
        AMC_HIP(hipMemcpyAsync(p + i, h->d_ptab + theta_row(i) * AMC_MAX_MOVES + k, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));
    return AMC_OK;
}

int amc_n_params(amc_handle* h, int* n_params, int* gd_stride)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_n_params: NULL handle");
    if (n_params) *n_params = h->n_params;
    if (gd_stride) *gd_stride = amc::pg_gd_stride(h->n_params);
    return AMC_OK;
}

int amc_parameters_begin(amc_handle* h)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_parameters_begin: NULL handle");
    if (h->params_pending) return fail(AMC_ERR_STATE, "amc_parameters_begin: a read is already in flight (call amc_parameters_end)");
    AMC_HIP(hipSetDevice(h->device));
    { const int rc = pg_resolve(h); if (rc != AMC_OK) return rc; }
    AMC_HIP(hipMemcpyAsync(h->h_params, h->d_ptab + amc::PT_SIGMA * AMC_MAX_MOVES, (size_t)h->K * sizeof(double), hipMemcpyDeviceToHost,
                           h->stream));
    if (h->n_params > 1)         // parameters 1 .. P - 1: consecutive rows of the table
        AMC_HIP(hipMemcpyAsync(h->h_params + AMC_MAX_MOVES, h->d_ptab + amc::PT_THETA1 * AMC_MAX_MOVES,
                               (size_t)(h->n_params - 1) * AMC_MAX_MOVES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    AMC_HIP(hipEventRecord(h->ev_params, h->stream));
    h->params_pending = true;
    return AMC_OK;
}

static int parameters_end_impl(amc_handle* h, const char* who, double* out, int per_move)
{
    if (!h || !out) return fail(AMC_ERR_BAD_ARG, "%s: NULL argument", who);
    if (!h->params_pending) return fail(AMC_ERR_STATE, "%s: no read in flight (call amc_parameters_begin)", who);
    AMC_HIP(hipSetDevice(h->device));
    AMC_HIP(wait_event(h->ev_params));           // waits for that copy only, not for work queued after it
    h->params_pending = false;
    for (int k = 0; k < h->K; ++k)
        for (int p = 0; p < per_move; ++p) out[(size_t)k * per_move + p] = h->h_params[(size_t)p * AMC_MAX_MOVES + k];
    return AMC_OK;
}

int amc_parameters_end(amc_handle* h, double* sigma) { return parameters_end_impl(h, "amc_parameters_end", sigma, 1); }

int amc_parameters_end_all(amc_handle* h, double* parameters, int n)
{
    if (h && n != h->K * h->n_params)
        return fail(AMC_ERR_BAD_ARG, "amc_parameters_end_all: this handle has %d moves of %d parameters", h->K, h->n_params);
    return parameters_end_impl(h, "amc_parameters_end_all", parameters, h ? h->n_params : 1);
}

int amc_sync(amc_handle* h)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_sync: NULL handle");
    AMC_HIP(hipSetDevice(h->device));
    AMC_HIP(wait_stream(h->stream));
    return AMC_OK;
}

int amc_get_stream(amc_handle* h, void** stream)
{
    if (!h || !stream) return fail(AMC_ERR_BAD_ARG, "amc_get_stream: NULL argument");
    *stream = (void*)h->stream;
    return AMC_OK;
}

int amc_timing_begin(amc_handle* h)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_timing_begin: NULL handle");
    AMC_HIP(hipSetDevice(h->device));
    AMC_HIP(hipEventRecord(h->ev0, h->stream));
    return AMC_OK;
}

int amc_timing_mark(amc_handle* h)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_timing_mark: NULL handle");
    AMC_HIP(hipSetDevice(h->device));
    AMC_HIP(hipEventRecord(h->ev1, h->stream));
    h->ev1_marked = true;
    return AMC_OK;
}

int amc_timing_end(amc_handle* h, double* elapsed_ms)
{
    if (!h || !elapsed_ms) return fail(AMC_ERR_BAD_ARG, "amc_timing_end: NULL argument");
    AMC_HIP(hipSetDevice(h->device));
    if (!h->ev1_marked) AMC_HIP(hipEventRecord(h->ev1, h->stream));
    h->ev1_marked = false;
    AMC_HIP(wait_stream(h->stream));           // the end event has completed once the stream has drained up to it
    AMC_HIP(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    AMC_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *elapsed_ms = (double)ms;
    return AMC_OK;
}



}  // extern "C"
