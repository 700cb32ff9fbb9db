// amc_internal.h -- what the host-side translation units of libamc.so share: the handle, the error convention, the RCCL and
// hiprtc surfaces resolved with dlopen.  Nothing here is part of the C ABI (include/amc.h); the functions declared here have
// hidden visibility.
//   amc_api.hip       handles, state, sweeps, callback reductions
//   amc_pg.hip        the estimator's host side (amc_pg_*, amc_pgmc_steps*)
//   amc_rtc.hip       kernels compiled at run time for script-defined models (hiprtc, code-object cache)
//   amc_comm.hip      the engine's own RCCL communicator (amc_comm_*, amc_allreduce_*)
//   amc_selftest.hip  parity-test hooks
//   amc_pg_fused.hip  kernel instantiations built with other code-generation options
#pragma once

#include "../../include/amc.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "amc_kernels.h"

#ifndef AMC_BUILD_ARCH
#define AMC_BUILD_ARCH "gfx950"      // the Makefile passes the arch the offline kernels were compiled for
#endif

#define AMC_INTERNAL __attribute__((visibility("hidden")))

// Every C entry returns 0 or a negative amc_status and leaves its message in a thread-local string (amc_last_error()).
AMC_INTERNAL int fail(int code, const char* fmt, ...);

#define AMC_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? AMC_ERR_OOM : AMC_ERR_HIP, "%s failed: %s",    \
                        #call, hipGetErrorString(e_));                                             \
    } while (0)

// Minimal RCCL surface, resolved with dlopen so the library has no link-time RCCL
// dependency and shares the instance a host process may already have loaded.
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, const void*, int) = nullptr;   // id passed by pointer (see shim)
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;       // optional: what the communicator says about itself (amc_comm_info)
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
};


static const int RED_HOST_STRIDE = amc::RED_ROW_WORDS;   // 64-bit words per row of the callback sums' block rows (red_finish)
static const int RATIO_STRIDE = 4;      // columns per row of the fold's acceptance-ratio partials (K <= 4): XS_ROW_Q words per move
static const int PG_MAX_COLS = AMC_MAX_LEARN * 4;   // GradientData columns of one estimator launch
static const int PG_NP_MAX_COLS = 1 + 2 * AMC_MAX_NP + AMC_MAX_NP * (AMC_MAX_NP + 1) / 2;   // ... of one move with AMC_MAX_NP parameters (< PG_MAX_COLS)
static const int RED_TICKETS = 2;       // reductions that may be in flight per handle (amc_reduce_begin .. amc_reduce_end)

// One reduction in flight: where its block rows land and what amc_reduce_end needs to finish it.
struct RedTicket {
    bool pending = false;
    int rows = 0;                    // block rows of the sums over x in h_rows
    int ratio_rows = 0;              // rows of h_ratio that belong to it (0: none)
    bool ratio_acc = false;          // the per-move ratio totals come from h_ratio_acc (K > 4)
    uint64_t t_counted = 0;
    int row_stride = RED_HOST_STRIDE;    // words per row of h_rows: the wide form, or amc::RED_COMPACT_WORDS (red_finish)
    int cols = amc::RED_WANT_ALL;        // the sums that were formed (amc_set_reduce_columns at the time)
    hipEvent_t ev = nullptr;
    amc::xs_word* h_rows = nullptr;      // pinned [n_slots][RED_HOST_STRIDE]
    amc::xs_word* h_ratio = nullptr;     // pinned [n_slots][RATIO_STRIDE]
    unsigned long long* d_ratio_acc = nullptr;   // [AMC_MAX_MOVES][3] (reduce_kernel, K > 4)
    unsigned long long* h_ratio_acc = nullptr;   // pinned copy
};

struct amc_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t M = 0, M_pad = 0, offset = 0, M_global = 0;
    int potential = 0, K = 1, sweepstep = 1;
    bool counters = false;      // per-chain counters kept
    bool beta_arr = false;
    double beta = 1.0;
    uint64_t seed = 0;
    uint64_t t = 0;             // MH steps done (Philox step index)
    uint64_t t_counted = 0;     // MH steps counted in acc/tot since creation
    uint64_t t_est = 0;         // estimator calls done
    double* d_x = nullptr;
    double* d_beta = nullptr;
    uint32_t* d_acc = nullptr;
    uint32_t* d_tot = nullptr;
    // K <= 4 handles (narrow == true) keep the counters as two u16 planes instead: low halves here, high halves in *_hi;
    // the high planes stay all zero, and untouched by the folds, until counter_room() sets use_high before the call that
    // would count step 65 536 (fold_log_kernel<.., HIGH>).  Exactly one of the two forms is allocated.
    uint16_t* d_acc16 = nullptr;
    uint16_t* d_tot16 = nullptr;
    uint16_t* d_acc_hi = nullptr;
    uint16_t* d_tot_hi = nullptr;
    bool narrow = false;
    bool use_high = false;
    // Counts beyond 32 bits (counter_rebase): what the arrays above have been carried into, nullptr until the first carry --
    // [K][M_pad] / [K - 1][M_pad] 64-bit integers --, the steps counted with them, and their pool totals (host side)
    unsigned long long* d_acc_base = nullptr;
    unsigned long long* d_tot_base = nullptr;
    uint64_t t_base = 0;
    unsigned long long base_acc_total[AMC_MAX_MOVES] = {0}, base_tot_total[AMC_MAX_MOVES] = {0};
    uint8_t* d_log = nullptr;   // [log_depth][M_pad / 2 or M_pad] step log: (move << 1) | accepted per chain and MH step (log_form)
    int log_depth = 32;         // rows of the step log: 2 GiB worth, between 16 and 128 (env AMC_LOG_DEPTH, 1..255: the fold counts rows in bytes)
    int log_fill = 0;           // rows written since the last fold into d_acc / d_tot
    double* d_ptab = nullptr;
    uint8_t* d_pick = nullptr;  // [AMC_PICK_CELLS] move pick by the 12 leading bits of the pick uniform (K > 1)
    unsigned long long* d_totals = nullptr;   // [2*K]: accepted, total (K > 1, filled on demand)
    unsigned long long* d_acc_slots = nullptr; // [max grid]: per-block accepted counts (K == 1)
    int n_slots = 0;
    amc::xs_word* d_partials = nullptr;   // [groups][nl * 4][PG_GROUP][words per column]: block rows of the estimator's fold
    RedTicket red[RED_TICKETS];      // reductions in flight, oldest first from red_head
    int red_head = 0, red_count = 0;
    double* d_out = nullptr;    // records of the estimator's fold: [comm ranks][PG_MAX_COLS][XS_WORDS]
    int d_out_ranks = 1;
    double* h_pg_out = nullptr; // pinned: records of amc_pg_estimate
    int red_blocks = 0;
    int red_cols = amc::RED_WANT_ALL;   // the callback sums a reduction forms (amc_set_reduce_columns)
    bool wide_red_rows = false;         // env AMC_WIDE_RED_ROWS=1 (read at amc_create; tests): the wide row form whatever the launch
    bool shard_route_one_rank = false;  // env AMC_SHARD_ROUTE_ON_ONE_RANK=1 (measurement, tests): a communicator of one rank takes the route of several
    bool no_deferred_update = false;    // env AMC_NO_DEFERRED_UPDATE=1 (read at amc_create; tests, A/B): every fused time step takes its own learning step
    int n_cu = 256;
    int blocks_per_cu = 8;      // grid cap = n_cu * blocks_per_cu blocks of 256, grid-stride beyond
    int blocks_per_cu_single = 8;   // ... of single-step sweep launches (6 for the K = 1 pool-wide-counter form)
    int blocks_per_cu_red = 5;      // ... of the sweep launch that also forms the callback sums (env AMC_BLOCKS_PER_CU_REDUCE)
    int blocks_per_cu_pg = 0;       // ... of the estimator kernels when AMC_BLOCKS_PER_CU is given; 0: what a CU HOLDS of the kernel form at hand
                                    // (hipOccupancyMaxActiveBlocksPerMultiprocessor: 5 for the built-in forms, 4 for most hiprtc ones), see pg_plan
    int occ_query = 0;              // out-slot of a launch_pg call made with grid < 0 (a query, nothing is launched)
    std::map<int, int> pg_resident; // resident blocks per CU of the estimator kernel forms, by (nl, sweep, reduce)
    std::map<int, std::string> class_form_errors;  // pools of several classes: the several-move estimator forms (nl, sweep, reduce) that do NOT build, with
                                                   // the compiler's last words about each (amc_pg.hip class_general_route, amc_pg_route)
    std::string class_form_error;   // ... of the form the last class_general_route call asked about ("" when it builds)
    bool no_column_skip = false;    // env AMC_NO_COLUMN_SKIP=1 (A/B, tests): fused script-defined steps sum every GradientData column whatever the optimiser reads
    bool np_small_launches = false;    // A/B knob (AMC_NP_SMALL_LAUNCHES=1): several parameters, several learnable moves: records + the small accumulate / update launches, as before round 6
    bool class_per_move_forced = false;     // env AMC_CLASS_PER_MOVE=1 (read at amc_create; A/B, tests): class pools take one estimator launch per learnable move
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long* d_hist = nullptr;   // running histogram of amc_histogram_accumulate: [hist_bins + 3]
    int hist_bins = 0;
    double hist_lo = 0.0, hist_hi = 0.0;
    hipEvent_t ev_params = nullptr;   // behind the copy queued by amc_parameters_begin
    double* h_params = nullptr;       // pinned [AMC_MAX_NP][AMC_MAX_MOVES]: its destination (row p: parameter p of every move)
    bool params_pending = false;
    bool ev1_marked = false;    // amc_timing_mark recorded the end event already
    void* comm = nullptr;
    int comm_rank = 0, comm_ranks = 1;   // this shard's slot in record gathers (amc_comm_init's arguments)
    int comm_capacity = 0;               // doubles d_comm / h_comm hold
    double* d_comm = nullptr;
    double* h_comm = nullptr;            // pinned staging of amc_allreduce_sum's values (the caller's buffer is pageable)
    hipStream_t comm_stream = nullptr;   // amc_allreduce_sum's own stream: host-side sums must not wait for the queued sweeps
    hipEvent_t ev_comm_main = nullptr;   // behind the last collective queued on the engine's stream (the estimator's all-reduce)
    bool comm_main_pending = false;      // ... which comm_stream has not been ordered behind yet
    double* d_gd_acc = nullptr;   // [AMC_MAX_MOVES][5] running GradientData per move (device-resident estimator); n_params > 1:
                                  // [AMC_MAX_MOVES][AMC_GD_STRIDE_MAX], fields as in amc::pg_np_unpack
    int* d_status = nullptr;      // [1] sticky flag: a learning step was rejected
    uint32_t* d_pg_tickets = nullptr;   // [1 + groups] arrival counters of the estimator kernel's in-kernel final reduce
    amc::xs_word* d_pg_groups = nullptr;   // [nl * 4][PG_GROUP][words per column]: group rows
    double* d_theta_ring = nullptr;     // [2][AMC_MAX_LEARN]: sigma of the learnable moves as the fused launches of even / odd estimator steps used it
    // A learning step a fused time step left to the next launch's prologue (amc::pg_apply_pending): what it needs to be taken --
    // by that launch, or by pg_resolve_kernel when anything else wants the parameter table first
    struct {
        bool active = false;
        int source = 0;                 // amc::PG_PENDING_GROUPS / _RECORDS
        int groups = 0;                 // groups of PG_GROUP blocks the launch wrote
        int n_learn = 0;
        uint64_t t_est = 0;             // the estimator step of that launch (its parity names the ring slot and the group rows)
    } pend;
    bool pend_consumed = false;         // the last estimator launch took the pending step in its prologue (pg_launch)
    uint64_t gd_nonzero = 0;            // moves whose gradients_data on the device may be non-zero (estimator steps since their last update)
    amc::PgTail* d_pg_tail = nullptr;   // the estimator kernel's per-configuration record (see amc::PgTail)
    amc::PgTail pg_tail_host;           // ... and what it holds now (rewritten only when it changes)
    bool pg_tail_valid = false;
    Rccl rccl;
    bool exact_accept = false;    // env AMC_EXACT_ACCEPT=1: no accept filter (every decision in the reference's arithmetic)
    std::string arch = AMC_BUILD_ARCH;   // the device's ISA name (gcnArchName up to its first ':'): what hiprtc compiles for
    std::string pot_expr;         // AMC_POTENTIAL_CUSTOM: the C expression of potential(x); '\x02' in front: Float32 state
    bool f32 = false;             // state_dtype == AMC_DTYPE_F32: d_x / d_beta hold floats
    bool scaled_policy = false;   // the proposal width is sigma * scale(x) (amc_create_policy_model)
    bool script_policy = false;   // sample_action! / log_proposal_density are script-defined expressions (amc_create_proposal_model)
    bool script_dlogq = false;    // ... and so is d logq / d sigma: the estimator is available
    int n_params = 1;             // parameters of the moves' policy (amc_create_policy_model; 1: sigma)
    int n_classes = 1;            // policy / action classes of the pool (amc_create_mixed_model)
    int class_of_move[AMC_MAX_MOVES] = {0};
    bool use_rtc = false;         // custom potential or Float32 state: every kernel that touches x is compiled at run time
    double* d_x64 = nullptr;      // f32 only: [M_pad] doubles, staging for uploads / downloads / host-side readers
    std::map<std::string, hipFunction_t> rtc_fn;   // kernel instantiation -> function of a module loaded on `device`
    std::vector<hipModule_t> rtc_mods;
};

// ---- shared between the translation units ----------------------------------------------------------------------------------------
AMC_INTERNAL int pg_resolve(amc_handle* h);      // takes a pending learning step now (amc_pg.hip)
// launch plumbing of amc_api.hip that the estimator's host code (amc_pg.hip) shares
AMC_INTERNAL int grid_for(const amc_handle* h, int64_t n_items, int blocks_per_cu = 0);
AMC_INTERNAL int log_form(const amc_handle* h);
AMC_INTERNAL int log_room(amc_handle* h, int* rows);
AMC_INTERNAL int red_form(const amc_handle* h);
AMC_INTERNAL int nl_capacity(int n_learn);
AMC_INTERNAL hipError_t wait_stream(hipStream_t stream);
AMC_INTERNAL int counter_room(amc_handle* h, const char* who, uint64_t steps);
AMC_INTERNAL int red_row_stride(const amc_handle* h, int grid);
AMC_INTERNAL amc::SweepArgs make_sweep_args(const amc_handle* h, int32_t n_steps);
AMC_INTERNAL int sweep_impl(amc_handle* h, int64_t n_sweeps, bool fuse_reduce, int* grid_out);
AMC_INTERNAL RedTicket* red_next(amc_handle* h);
AMC_INTERNAL int finish_fused_reduce(amc_handle* h, int grid);
AMC_INTERNAL bool reduce_fits_in_grid(const amc_handle* h, int grid);
AMC_INTERNAL void comm_release(amc_handle* h);   // drops the handle's communicator and its buffers (amc_comm.hip)
// kernels compiled at run time (amc_rtc.hip)
struct RtcCode { std::vector<char> code; std::string lowered; };
AMC_INTERNAL int validate_potential_expr(const char* expr, const char* what = "custom potential", const char* var = "x");
AMC_INTERNAL int rtc_compile(const std::string& expr_in, const std::string& inst, const std::string& arch, const RtcCode** out, std::string* log_out);
AMC_INTERNAL int rtc_function(amc_handle* h, const std::string& inst, hipFunction_t* fn);
AMC_INTERNAL int rtc_launch(amc_handle* h, const std::string& inst, int grid, void** params);
