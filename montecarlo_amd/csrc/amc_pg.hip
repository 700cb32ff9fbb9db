// amc_pg.hip -- the host side of the policy-gradient estimator (amc_pg_*, amc_pgmc_steps*): launch plans, the estimator kernel's
// forms (built-in potentials here; script-defined models through amc_rtc.hip), records across shards, the device-resident
// gradients_data += / learning step, and the learning step a fused launch leaves to its successor.
// Reference: src/PolicyGuided/estimator.jl:111-134, update.jl:50-57, learning.jl:32-164; run! order src/simulation.jl:185-190.
#define AMC_KERNEL_LINKAGE static      // this object's own copies of the small kernels it launches (accumulate / update / resolve / tail record)
#include "amc_internal.h"

// defined in amc_pg_fused.hip (compiled with other code-generation options, see there): not instantiated here
namespace amc {
#define AMC_PG_FUSED(POT, NL, BETA)                                                                                     \
    extern template __global__ void pg_estimate_kernel<POT, NL, BETA, 2, RED_FORM_NONE>(const PgArgs, const SweepArgs);   \
    extern template __global__ void pg_estimate_kernel<POT, NL, BETA, 2, RED_FORM_COLS>(const PgArgs, const SweepArgs);   \
    extern template __global__ void pg_estimate_kernel<POT, NL, BETA, 2, RED_FORM_E>(const PgArgs, const SweepArgs);      \
    extern template __global__ void pg_estimate_kernel<POT, NL, BETA, 2, RED_FORM_NONE, true>(const PgArgs, const SweepArgs)
AMC_PG_FUSED(POT_HARMONIC, 1, false);
AMC_PG_FUSED(POT_HARMONIC, 1, true);
AMC_PG_FUSED(POT_HARMONIC, 2, false);
AMC_PG_FUSED(POT_HARMONIC, 2, true);
AMC_PG_FUSED(POT_DOUBLE_WELL, 1, false);
AMC_PG_FUSED(POT_DOUBLE_WELL, 1, true);
AMC_PG_FUSED(POT_DOUBLE_WELL, 2, false);
AMC_PG_FUSED(POT_DOUBLE_WELL, 2, true);
#undef AMC_PG_FUSED
}  // namespace amc

namespace {

const char* tf(bool b) { return b ? "true" : "false"; }

template <int POT, int NL, int SWEEP, int REDUCE = amc::RED_FORM_NONE, bool MID = false>
int launch_pg_nls(amc_handle* h, const amc::PgArgs& a, const amc::SweepArgs& sw, int grid)
{
    if (grid < 0) {            // a query: how many blocks of this form a CU holds
        int nb = 0;
        if (h->beta_arr) AMC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, amc::pg_estimate_kernel<POT, NL, true, SWEEP, REDUCE, MID>, AMC_BLOCK, 0));
        else AMC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, amc::pg_estimate_kernel<POT, NL, false, SWEEP, REDUCE, MID>, AMC_BLOCK, 0));
        h->occ_query = nb;
        return AMC_OK;
    }
    if (h->beta_arr)
        hipLaunchKernelGGL((amc::pg_estimate_kernel<POT, NL, true, SWEEP, REDUCE, MID>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, a, sw);
    else
        hipLaunchKernelGGL((amc::pg_estimate_kernel<POT, NL, false, SWEEP, REDUCE, MID>), dim3(grid), dim3(AMC_BLOCK), 0, h->stream, a, sw);
    AMC_HIP(hipGetLastError());
    return AMC_OK;
}

// sweep: 0 = estimator only; 1 / 2 / 3 = preceded by one make_step!(::Metropolis) in the same launch (K == 1 with the step
// log / K > 1 / K == 1 with the pool-wide counter; offered for up to 2 learnable moves, see `fused` in pgmc_steps_impl)
// reduce (sweep != 0): the launch also leaves the callback sums of the state it stores (pg_estimate_kernel<.., REDUCE>)
// mid (sweep == 0): a lane's GradientData accumulators do not take the whole launch -- the kernel form that flushes as it goes
template <int POT>
int launch_pg(amc_handle* h, const amc::PgArgs& a, const amc::SweepArgs& sw, int grid, int nl_cap, int sweep, bool reduce, bool mid)
{
    if (mid && reduce) return fail(AMC_ERR_STATE, "launch_pg: the callback sums ride on launches that need no flush mid-launch (see pg_plan)");
    constexpr int NONE = amc::RED_FORM_NONE, COLS = amc::RED_FORM_COLS, E = amc::RED_FORM_E;
    if (mid) {
        if (sweep == 1) return nl_cap == 1 ? launch_pg_nls<POT, 1, 1, NONE, true>(h, a, sw, grid) : launch_pg_nls<POT, 2, 1, NONE, true>(h, a, sw, grid);
        if (sweep == 2) return nl_cap == 1 ? launch_pg_nls<POT, 1, 2, NONE, true>(h, a, sw, grid) : launch_pg_nls<POT, 2, 2, NONE, true>(h, a, sw, grid);
        if (sweep == 3) return nl_cap == 1 ? launch_pg_nls<POT, 1, 3, NONE, true>(h, a, sw, grid) : launch_pg_nls<POT, 2, 3, NONE, true>(h, a, sw, grid);
        switch (nl_cap) {
        case 1: return launch_pg_nls<POT, 1, 0, NONE, true>(h, a, sw, grid);
        case 2: return launch_pg_nls<POT, 2, 0, NONE, true>(h, a, sw, grid);
        case 4: return launch_pg_nls<POT, 4, 0, NONE, true>(h, a, sw, grid);
        default: return launch_pg_nls<POT, 8, 0, NONE, true>(h, a, sw, grid);
        }
    }
    if (reduce && red_form(h) == E) {
        if (sweep == 1) return nl_cap == 1 ? launch_pg_nls<POT, 1, 1, E>(h, a, sw, grid) : launch_pg_nls<POT, 2, 1, E>(h, a, sw, grid);
        if (sweep == 2) return nl_cap == 1 ? launch_pg_nls<POT, 1, 2, E>(h, a, sw, grid) : launch_pg_nls<POT, 2, 2, E>(h, a, sw, grid);
        if (sweep == 3) return nl_cap == 1 ? launch_pg_nls<POT, 1, 3, E>(h, a, sw, grid) : launch_pg_nls<POT, 2, 3, E>(h, a, sw, grid);
        return fail(AMC_ERR_STATE, "launch_pg: the callback sums ride on the fused time step only");
    }
    if (reduce) {
        if (sweep == 1) return nl_cap == 1 ? launch_pg_nls<POT, 1, 1, COLS>(h, a, sw, grid) : launch_pg_nls<POT, 2, 1, COLS>(h, a, sw, grid);
        if (sweep == 2) return nl_cap == 1 ? launch_pg_nls<POT, 1, 2, COLS>(h, a, sw, grid) : launch_pg_nls<POT, 2, 2, COLS>(h, a, sw, grid);
        if (sweep == 3) return nl_cap == 1 ? launch_pg_nls<POT, 1, 3, COLS>(h, a, sw, grid) : launch_pg_nls<POT, 2, 3, COLS>(h, a, sw, grid);
        return fail(AMC_ERR_STATE, "launch_pg: the callback sums ride on the fused time step only");
    }
    if (sweep == 1) return nl_cap == 1 ? launch_pg_nls<POT, 1, 1>(h, a, sw, grid) : launch_pg_nls<POT, 2, 1>(h, a, sw, grid);
    if (sweep == 2) return nl_cap == 1 ? launch_pg_nls<POT, 1, 2>(h, a, sw, grid) : launch_pg_nls<POT, 2, 2>(h, a, sw, grid);
    if (sweep == 3) return nl_cap == 1 ? launch_pg_nls<POT, 1, 3>(h, a, sw, grid) : launch_pg_nls<POT, 2, 3>(h, a, sw, grid);
    switch (nl_cap) {
    case 1: return launch_pg_nls<POT, 1, 0>(h, a, sw, grid);
    case 2: return launch_pg_nls<POT, 2, 0>(h, a, sw, grid);
    case 4: return launch_pg_nls<POT, 4, 0>(h, a, sw, grid);
    default: return launch_pg_nls<POT, 8, 0>(h, a, sw, grid);
    }
}

int launch_pg_custom(amc_handle* h, amc::PgArgs& a, amc::SweepArgs& sw, int grid, int nl_cap, int sweep, bool reduce, bool mid)
{
    if (mid && reduce) return fail(AMC_ERR_STATE, "launch_pg_custom: the callback sums ride on launches that need no flush mid-launch");
    const std::string inst = "amc::pg_estimate_kernel<" + std::to_string(h->potential) + "," + std::to_string(nl_cap) + "," + tf(h->beta_arr) + "," +
                             std::to_string(sweep) + "," + std::to_string(reduce ? red_form(h) : (int)amc::RED_FORM_NONE) + "," + tf(mid) + ">";
    if (grid < 0) {            // a query: how many blocks of this form a CU holds
        hipFunction_t fn = nullptr;
        { const int rc = rtc_function(h, inst, &fn); if (rc != AMC_OK) return rc; }
        int nb = 0;
        AMC_HIP(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, AMC_BLOCK, 0));
        h->occ_query = nb;
        return AMC_OK;
    }
    void* params[] = {&a, &sw};
    return rtc_launch(h, inst, grid, params);
}

}  // namespace

extern "C" {

// Takes a learning step that a fused time step left pending (amc::pg_apply_pending) NOW: one small launch that brings the
// parameter table up to date.  Everything that reads or writes the moves' parameters, gradients_data or the status flag -- other
// than the next fused launch, which takes the step in its prologue -- calls this first.  (The device's tail record still describes
// the pending step's configuration: a launch that changes it resolves before it rewrites.)
extern "C++" int pg_resolve(amc_handle* h)      // (declared in amc_internal.h: amc_api.hip and amc_comm.hip call it too)
{
    if (!h->pend.active) return AMC_OK;
    AMC_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(amc::pg_resolve_kernel, dim3(1), dim3(AMC_BLOCK), 0, h->stream, (const amc::PgTail*)h->d_pg_tail, h->pend.source,
                       (int)(h->pend.t_est & 1ull), h->pend.groups, h->pend.n_learn);
    AMC_HIP(hipGetLastError());
    h->pend.active = false;
    return AMC_OK;
}
// the kernel forms that can take a pending step in their prologue, and leave one (amc_kernels.h CAN_DEFER): the built-in
// potentials' Gaussian policy (kind-Q sums), at most two learnable moves
static bool pg_form_defers(const amc_handle* h, int n_learn)
{
    return !h->no_deferred_update && !h->use_rtc && h->n_params == 1 && h->n_classes == 1 && n_learn >= 1 && n_learn <= 2;
}

// The estimator's grid over this shard.
// The estimator's grid over this shard and the kernel form it needs.
// ONE round of resident blocks: every block's prologue and way out (table staging, the row's stores, the ticket's round trip) are
// latency the CU cannot hide behind other blocks' arithmetic at the end of a round, so 1.6 rounds (8 blocks per CU on 5 slots) paid
// them 1.6 times -- 43.1 -> 40.6 us for the estimator launch, 66.0 -> 65.5 for the fused time step at 1e7 chains; whole rounds are
// good, fractions bad (NOTES_r04.md).  What a CU holds of the kernel form at hand is asked of the runtime (5 blocks for the built-in
// forms, 4 for the form that also leaves the callback sums and for most hiprtc forms).
// An estimator launch puts 2 q_batch summands per trip into each of a lane's GradientData accumulators, which hold XS_GD_LANE_CAP
// (kind Q: the built-in potentials) or XS_LANE_CAP (kind R: hiprtc forms) of them between two flushes (amc_xsum.h).  Launches whose
// lanes stay within that run the kernel form without flush code in its sampling loop (pg_estimate_kernel, MIDFLUSH) -- with two
// rounds of blocks if one does not fit and two do --; the others empty full accumulators into integers of the lane on the way.
struct PgPlan { int grid; bool mid; };
static int pg_plan(amc_handle* h, int nl, int sweep, bool reduce, int q_batch, PgPlan* plan)
{
    const int64_t pairs = (h->M + 1) / 2;
    int limit = h->red_blocks;
    if (limit > amc::PG_GROUP * amc::PG_GROUP) limit = amc::PG_GROUP * amc::PG_GROUP;      // two levels of PG_GROUP in the kernel's tail
    const int64_t cap = h->use_rtc ? amc::xs::XS_LANE_CAP : amc::xs::XS_GD_LANE_CAP;
    auto fits = [&](int grid) { const int64_t lanes = (int64_t)grid * AMC_BLOCK; return 2 * (int64_t)q_batch * ((pairs + lanes - 1) / lanes) <= cap; };
    // what a CU holds of the kernel form at hand (AMC_BLOCKS_PER_CU given: that many), asked once per form -- the flushing
    // form (mid) is an instantiation of its own, with its own register count
    auto resident = [&](bool mid, int* per_cu) -> int {
        if (h->blocks_per_cu_pg) { *per_cu = h->blocks_per_cu_pg; return AMC_OK; }
        const bool red = reduce && !mid;             // (the callback sums ride on launches that need no flush: launch_pg)
        const int key = (nl << 8) | (sweep << 4) | (red ? red_form(h) : 0) | (mid ? 4 : 0);      // (the callback sums' form may change: amc_set_reduce_columns)
        auto it = h->pg_resident.find(key);
        if (it == h->pg_resident.end()) {
            amc::PgArgs a0;
            std::memset(&a0, 0, sizeof(a0));
            amc::SweepArgs s0;
            std::memset(&s0, 0, sizeof(s0));
            h->occ_query = 0;
            const int rc = h->use_rtc                                    ? launch_pg_custom(h, a0, s0, -1, nl, sweep, red, mid)
                           : (h->potential == AMC_POTENTIAL_DOUBLE_WELL) ? launch_pg<amc::POT_DOUBLE_WELL>(h, a0, s0, -1, nl, sweep, red, mid)
                                                                         : launch_pg<amc::POT_HARMONIC>(h, a0, s0, -1, nl, sweep, red, mid);
            if (rc != AMC_OK) return rc;
            if (std::getenv("AMC_DEBUG_PLAN")) std::fprintf(stderr, "[amc] estimator form nl=%d sweep=%d reduce=%d mid=%d: %d resident blocks per CU\n", nl, sweep, (int)red, (int)mid, h->occ_query);
            it = h->pg_resident.emplace(key, h->occ_query > 0 ? h->occ_query : 5).first;
        }
        *per_cu = it->second;
        return AMC_OK;
    };
    auto grid_of = [&](int per_cu) { const int g = grid_for(h, pairs, per_cu); return g > limit ? limit : g; };
    int per_cu = 0;
    { const int rc = resident(false, &per_cu); if (rc != AMC_OK) return rc; }
    plan->grid = grid_of(per_cu);
    plan->mid = !fits(plan->grid);
    if (plan->mid && h->blocks_per_cu_pg == 0) {
        const int two = grid_of(2 * per_cu);
        if (fits(two)) { plan->grid = two; plan->mid = false; return AMC_OK; }
        // the flushing form, on two rounds of ITS resident blocks (3 % faster than on one: q_batch 4, 143.6 -> 138.8 us)
        { const int rc = resident(true, &per_cu); if (rc != AMC_OK) return rc; }
        plan->grid = grid_of(2 * per_cu);
    }
    return AMC_OK;
}

// Handles whose estimator takes one launch per learnable move: policies with several parameters (the move's columns fill a row
// of the kernel's tail).  Pools of several policy / action classes (one parameter per move) take the route every other pool takes
// -- every learnable move in ONE launch, the time step fused, as the reference's make_step!(::PolicyGradientEstimator) loops over
// all learnable moves whatever their policy types (src/PolicyGuided/estimator.jl:111-134, src/metropolis.jl:140-162) -- wherever
// the kernel form of the call BUILDS: hipcc 7.2 meets a back-end error ("illegal VGPR to SGPR copy", a fatal error of the compiler)
// on the unrolled loop over moves with a class switch inside for SOME pools and move counts.  The compiler runs in a child process
// (amc_rtc.hip build_in_child), so that is an AMC_ERR_COMPILE like any other: class_general_route asks for the form once, keeps
// the answer per form, and a form that does not build sends its calls down the per-move route (PgArgs.l_base), which needs only
// the NL = 1 form.  Same samples (draw ids by the move's index in the call), same integer sums: same bits either way.
static bool per_move_launches(const amc_handle* h) { return h->n_params > 1; }
// ... of which a policy with several parameters (one class) has the single-launch forms too when ONE move learns: the launch's
// tail is generic in P (pg_tail_np), so sweep + estimator + gradients_data += + learning step are one launch as for P = 1
static bool np_single_launch(const amc_handle* h, int n_learn) { return h->n_params > 1 && h->n_classes == 1 && n_learn == 1; }
// ... and with SEVERAL learnable moves on one shard, a chain of such launches, one per move, each with its own tail (pg_accumulate_impl)
static bool np_move_chain(const amc_handle* h, int n_learn)
{
    return h->n_params > 1 && h->n_classes == 1 && !h->comm && n_learn > 1 && !h->np_small_launches;
}

// A pool of several classes: does this call go down the general route?  *general = the kernel form the call needs builds.  The
// compiler is asked at most once per form and process (rtc_compile keeps what it learnt, failures included, and so does the
// code-object cache directory for later processes); errors other than the compiler's are the call's errors.
static int class_general_route(amc_handle* h, int n_learn, bool with_sweep, bool reduce, int q_batch, bool* general)
{
    *general = true;
    if (h->n_classes <= 1 || h->n_params > 1) return AMC_OK;
    if (h->class_per_move_forced) { *general = false; return AMC_OK; }
    if (n_learn < 1 || n_learn > AMC_MAX_LEARN || q_batch < 1 || q_batch > AMC_MAX_QBATCH) return AMC_OK;      // (the call's own validation speaks)
    const int nl = nl_capacity(n_learn);
    const int sweep = with_sweep ? (h->K > 1 ? 2 : (h->d_log ? 1 : 3)) : 0;
    const bool red = reduce && with_sweep;
    const int key = (nl << 8) | (sweep << 4) | (red ? red_form(h) : 0);
    h->class_form_error.clear();
    auto it = h->class_form_errors.find(key);
    if (it != h->class_form_errors.end()) { h->class_form_error = it->second; *general = false; return AMC_OK; }
    PgPlan plan;
    const int rc = pg_plan(h, nl, sweep, red, q_batch, &plan);      // (cheap once the form is loaded: two map lookups)
    if (rc == AMC_OK) return AMC_OK;
    if (rc != AMC_ERR_COMPILE) return rc;
    h->class_form_error = amc_last_error();
    h->class_form_errors[key] = h->class_form_error;
    if (std::getenv("AMC_DEBUG_PLAN"))
        std::fprintf(stderr, "[amc] class pool: estimator form nl=%d sweep=%d reduce=%d does not build, one launch per learnable move instead: %s\n", nl, sweep, (int)red, h->class_form_error.c_str());
    *general = false;
    return AMC_OK;
}

// Validates, launches K3 over this shard.  Shared by the host- and device-resident estimator paths.
// tail: 1 = the totals of (j, grad j, grad logq, g) per learnable move as records in h->d_out (this shard's slot), 2 = instead
// gradients_data += gd, 3 = + learning step (opt must be given); see PgArgs.
// with_sweep: the launch first does one make_step!(::Metropolis) (caller decided: `fused` in pgmc_steps_impl).
// reduce (with_sweep only): the launch also leaves the callback sums of the state it stores in the next reduction ticket's rows.
// l_base, advance (policies with several parameters: one launch per learnable move): the move's index in the estimator call,
// and whether this launch is the call's last (the estimator's step counter then advances).
// call_ids, call_n (a policy with several parameters, one launch per learnable move with the launch's own tail): the estimator CALL's
// learnable moves -- the tail's record (ids, optimisers) then describes the call, the launch names its move by l_base, and the record
// stays what it is from launch to launch (a rewrite is a 5 us launch of its own).
static int pg_launch(amc_handle* h, const char* who, int n_learn, const int* learn_ids, int q_batch, int* nl_out,
                     int tail = 1, const amc::PgOpts* opt = nullptr, bool with_sweep = false, bool reduce = false,
                     int* grid_out = nullptr, int l_base = 0, bool advance = true, const int* call_ids = nullptr, int call_n = 0)
{
    if (n_learn > 0 && !learn_ids) return fail(AMC_ERR_BAD_ARG, "%s: learn_ids is NULL", who);
    if (n_learn < 0 || n_learn > AMC_MAX_LEARN) return fail(AMC_ERR_BAD_ARG, "%s: n_learn must be in [0, %d]", who, AMC_MAX_LEARN);
    if (q_batch < 1 || q_batch > AMC_MAX_QBATCH || (int64_t)q_batch * n_learn >= 4096)
        return fail(AMC_ERR_BAD_ARG, "%s: q_batch must be in [1, %d] and q_batch*n_learn < 4096", who, AMC_MAX_QBATCH);
    for (int l = 0; l < n_learn; ++l)
        if (learn_ids[l] < 0 || learn_ids[l] >= h->K)
            return fail(AMC_ERR_BAD_ARG, "%s: learn_ids[%d] = %d out of range", who, l, learn_ids[l]);
    *nl_out = 0;
    // (a script-defined proposal that came without d logq / d theta expressions gets them by forward-mode differentiation of its
    // logq in the kernel -- amc_dual.h; the reference's withgrad_log_proposal_density! does the same with ForwardDiff, gradients.jl:28-33)
    if (n_learn == 0) { h->t_est += 1; return AMC_OK; }
    if (per_move_launches(h) && (n_learn != 1 || ((tail != 1 || with_sweep) && !np_single_launch(h, n_learn))))
        return fail(AMC_ERR_STATE, "%s: a policy with several parameters takes one learnable move per launch", who);
    AMC_HIP(hipSetDevice(h->device));
    amc::PgArgs a;
    a.x = h->d_x;
    a.beta_arr = h->beta_arr ? h->d_beta : nullptr;
    a.ptab = h->d_ptab;
    a.partials = h->d_partials;          // [groups][nl * 4][PG_GROUP][words per column] (amc::PgKind)
    a.n_chains = h->M;
    a.pair0 = (uint64_t)h->offset >> 1;
    a.t_est = h->t_est;
    a.q_batch = q_batch;
    a.n_learn = n_learn;
    for (int l = 0; l < AMC_MAX_LEARN; ++l) a.learn_ids[l] = l < n_learn ? learn_ids[l] : 0;
    a.key0 = (uint32_t)h->seed;
    a.key1 = (uint32_t)(h->seed >> 32);
    a.beta = h->beta;
    a.tail_mode = tail;
    // a script-defined form whose own tail takes the learning step: the column groups the moves' optimisers never read are not summed
    // (amc_estimator.h; gradients_data is consumed and reset in that tail, so nothing of them could be seen afterwards)
    if (opt && (tail == 3 || tail == (int)amc::PG_TAIL_GROUPS) && h->use_rtc && !h->no_column_skip)
        a.tail_mode |= amc::pg_skip_bits(opt->kind + (call_ids ? l_base : 0), n_learn);
    a.l_base = l_base;
    {
        amc::PgTail tl;
        std::memset(&tl, 0, sizeof(tl));          // padding included: the record is compared bytewise below
        tl.tickets = h->d_pg_tickets;
        tl.group_sums = h->d_pg_groups;
        tl.out = h->d_out;
        tl.gd_acc = h->d_gd_acc;
        tl.ptab_rw = h->d_ptab;
        tl.status = h->d_status;
        tl.n_samples = (double)h->M * (double)q_batch;
        tl.n_samples_global = (double)h->M_global * (double)q_batch;
        tl.theta_ring = h->d_theta_ring;
        tl.n_moves = h->K;
        tl.rank = h->comm ? h->comm_rank : 0;
        tl.n_ranks = h->comm ? h->comm_ranks : 1;
        // (a launch that only leaves records -- one per learnable move of a several-parameter policy or a pool of classes -- reads
        // neither the ids nor the optimisers from the record: kept out, or every such launch would rewrite it, 4.7 us each)
        if (call_ids)
            for (int l = 0; l < AMC_MAX_LEARN; ++l) tl.learn_ids[l] = l < call_n ? call_ids[l] : 0;
        else if (tail != 1 || !(per_move_launches(h) || h->n_classes > 1))
            for (int l = 0; l < AMC_MAX_LEARN; ++l) tl.learn_ids[l] = a.learn_ids[l];
        if (opt) tl.opt = *opt;
        // a learning step the previous fused launch left pending: this launch takes it in its prologue if it is the very next
        // estimator step, of a kernel form that can, under the same record (learnable moves, optimisers, sample count) --
        // anything else takes it now (pg_resolve: before the record is rewritten)
        h->pend_consumed = false;
        if (h->pend.active) {
            const bool same = h->pg_tail_valid && std::memcmp(&tl, &h->pg_tail_host, sizeof(tl)) == 0;
            if (same && pg_form_defers(h, n_learn) && h->t_est == h->pend.t_est + 1 && n_learn == h->pend.n_learn) {
                a.tail_mode |= (h->pend.source << 8) | (h->pend.groups << 16);
                h->pend.active = false;
                h->pend_consumed = true;
            } else {
                const int rcr = pg_resolve(h);
                if (rcr != AMC_OK) return rcr;
            }
        }
        if (!h->pg_tail_valid || std::memcmp(&tl, &h->pg_tail_host, sizeof(tl)) != 0) {
            // stream-ordered, the record travels as a kernel argument: launches already queued read the old one
            hipLaunchKernelGGL(amc::pg_tail_store_kernel, dim3(1), dim3(64), 0, h->stream, tl, h->d_pg_tail);
            AMC_HIP(hipGetLastError());
            h->pg_tail_host = tl;
            h->pg_tail_valid = true;
        }
        a.tail = h->d_pg_tail;
    }
    const int nl = nl_capacity(n_learn);
    int sweep = 0;
    if (with_sweep) {
        { const int rcc = counter_room(h, who, 1); if (rcc != AMC_OK) return rcc; }
        if (h->d_log) { int room = 0; const int rcf = log_room(h, &room); if (rcf != AMC_OK) return rcf; }
        sweep = h->K > 1 ? 2 : (h->d_log ? 1 : 3);
    }
    const bool red = reduce && with_sweep;
    PgPlan plan;
    { const int rcp = pg_plan(h, nl, sweep, red, q_batch, &plan); if (rcp != AMC_OK) return rcp; }
    const int grid = plan.grid;
    const bool mid = plan.mid;
    if (grid_out) *grid_out = grid;
    amc::SweepArgs sw = make_sweep_args(h, 1);
    sw.red_stride = red_row_stride(h, grid);
    const int rc = h->use_rtc                                    ? launch_pg_custom(h, a, sw, grid, nl, sweep, red, mid)
                   : (h->potential == AMC_POTENTIAL_DOUBLE_WELL) ? launch_pg<amc::POT_DOUBLE_WELL>(h, a, sw, grid, nl, sweep, red, mid)
                                                                 : launch_pg<amc::POT_HARMONIC>(h, a, sw, grid, nl, sweep, red, mid);
    if (rc != AMC_OK) return rc;
    if (with_sweep) {
        h->t += 1;
        h->t_counted += 1;
        if (h->d_log) h->log_fill += 1;
    }
    // tail 1: the launch itself left the columns' totals as records in d_out[ranks][n_learn * 4][XS_WORDS] (in-kernel final reduction)
    if (advance) h->t_est += 1;
    *nl_out = nl;
    return AMC_OK;
}

// The estimator's fold over this shard as records: n_learn * 4 of them (j, grad j, grad logq, g per learnable move).
// The arguments of an estimator call, checked BEFORE its first launch: a call may go in several (one per learnable move, or four moves
// at a time), and must not stop half way over an argument -- samples drawn, sums accumulated, the step counter not advanced.
static int pg_check_call(const amc_handle* h, const char* who, int n_learn, const int* learn_ids, int q_batch)
{
    if (n_learn < 0 || n_learn > AMC_MAX_LEARN) return fail(AMC_ERR_BAD_ARG, "%s: n_learn must be in [0, %d]", who, AMC_MAX_LEARN);
    if (n_learn > 0 && !learn_ids) return fail(AMC_ERR_BAD_ARG, "%s: learn_ids is NULL", who);
    for (int l = 0; l < n_learn; ++l)
        if (learn_ids[l] < 0 || learn_ids[l] >= h->K) return fail(AMC_ERR_BAD_ARG, "%s: learn_ids[%d] = %d out of range", who, l, learn_ids[l]);
    if (q_batch < 1 || q_batch > AMC_MAX_QBATCH || (int64_t)q_batch * n_learn >= 4096)
        return fail(AMC_ERR_BAD_ARG, "%s: q_batch must be in [1, %d] and q_batch*n_learn < 4096", who, AMC_MAX_QBATCH);
    return AMC_OK;
}

static int pg_estimate_records(amc_handle* h, const char* who, int n_learn, const int* learn_ids, int q_batch, const double** recs)
{
    int nl = 0;
    bool general = true;
    { const int rcc = pg_check_call(h, who, n_learn, learn_ids, q_batch); if (rcc != AMC_OK) return rcc; }
    { const int rcg = class_general_route(h, n_learn, false, false, q_batch, &general); if (rcg != AMC_OK) return rcg; }
    if (per_move_launches(h) || !general) {
        // one launch per learnable move, its 1 + 2P + P(P+1)/2 records behind those of the moves before it
        if (n_learn < 0 || n_learn > AMC_MAX_LEARN) return fail(AMC_ERR_BAD_ARG, "%s: n_learn must be in [0, %d]", who, AMC_MAX_LEARN);
        if (n_learn == 0) return pg_launch(h, who, 0, learn_ids, q_batch, &nl);
        const int nc = amc::pg_n_columns(h->n_params);
        const int slot = h->comm ? h->comm_rank : 0;
        const size_t n = (size_t)nc * amc::xs::XS_WORDS;
        for (int l = 0; l < n_learn; ++l) {
            const int rc = pg_launch(h, who, 1, learn_ids + l, q_batch, &nl, 1, nullptr, false, false, nullptr, l, l + 1 == n_learn);
            if (rc != AMC_OK) return rc;
            AMC_HIP(hipMemcpyAsync(h->h_pg_out + (size_t)l * n, h->d_out + (size_t)slot * n, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        }
        AMC_HIP(wait_stream(h->stream));
        *recs = h->h_pg_out;
        return AMC_OK;
    }
    const int rc = pg_launch(h, who, n_learn, learn_ids, q_batch, &nl);
    if (rc != AMC_OK || n_learn == 0) return rc;
    const int slot = h->comm ? h->comm_rank : 0;
    const size_t n = (size_t)n_learn * 4 * amc::xs::XS_WORDS;
    AMC_HIP(hipMemcpyAsync(h->h_pg_out, h->d_out + (size_t)slot * n, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    AMC_HIP(wait_stream(h->stream));
    *recs = h->h_pg_out;
    return AMC_OK;
}

int amc_pg_estimate(amc_handle* h, int n_learn, const int* learn_ids, int q_batch, double* out)
{
    if (!h || !out) return fail(AMC_ERR_BAD_ARG, "amc_pg_estimate: NULL argument");
    const double* recs = nullptr;
    const int rc = pg_estimate_records(h, "amc_pg_estimate", n_learn, learn_ids, q_batch, &recs);
    if (rc != AMC_OK || n_learn == 0) return rc;
    if (h->n_params > 1) {             // [j, grad j [P], grad logq [P], g [P][P], n] per move
        const int np = h->n_params, nc = amc::pg_n_columns(np), stride = amc::pg_gd_stride(np);
        for (int l = 0; l < n_learn; ++l) {
            double vals[PG_NP_MAX_COLS];
            for (int c = 0; c < nc; ++c) vals[c] = amc::xs::rec_round(recs + (size_t)(l * nc + c) * amc::xs::XS_WORDS);
            amc::pg_np_unpack(vals, np, out + (size_t)l * stride);
            out[(size_t)l * stride + stride - 1] = (double)h->M * (double)q_batch;
        }
        return AMC_OK;
    }
    for (int l = 0; l < n_learn; ++l) {
        for (int i = 0; i < 4; ++i) out[l * AMC_GD_STRIDE + i] = amc::xs::rec_round(recs + (size_t)(l * 4 + i) * amc::xs::XS_WORDS);
        out[l * AMC_GD_STRIDE + AMC_GD_N] = (double)h->M * (double)q_batch;
    }
    return AMC_OK;
}

int amc_pg_estimate_exact(amc_handle* h, int n_learn, const int* learn_ids, int q_batch, double* records)
{
    if (!h || !records) return fail(AMC_ERR_BAD_ARG, "amc_pg_estimate_exact: NULL argument");
    const double* recs = nullptr;
    const int rc = pg_estimate_records(h, "amc_pg_estimate_exact", n_learn, learn_ids, q_batch, &recs);
    if (rc != AMC_OK || n_learn == 0) return rc;
    if (h->n_params > 1) {
        const int np = h->n_params, nc = amc::pg_n_columns(np), stride = amc::pg_gd_stride(np);
        const size_t W = amc::xs::XS_WORDS;
        for (int l = 0; l < n_learn; ++l) {
            double* dst = records + (size_t)l * stride * W;
            const double* src = recs + (size_t)l * nc * W;
            std::memcpy(dst, src, (size_t)(1 + 2 * np) * W * sizeof(double));
            int at = 1 + 2 * np;
            for (int a = 0; a < np; ++a)
                for (int b = a; b < np; ++b) {
                    std::memcpy(dst + (size_t)(1 + 2 * np + a * np + b) * W, src + (size_t)at * W, W * sizeof(double));
                    std::memcpy(dst + (size_t)(1 + 2 * np + b * np + a) * W, src + (size_t)at * W, W * sizeof(double));
                    ++at;
                }
            amc::xs::rec_from_plain(dst + (size_t)(stride - 1) * W, (double)h->M * (double)q_batch);
        }
        return AMC_OK;
    }
    for (int l = 0; l < n_learn; ++l) {
        for (int i = 0; i < 4; ++i)
            std::memcpy(records + (size_t)(l * AMC_GD_STRIDE + i) * amc::xs::XS_WORDS, recs + (size_t)(l * 4 + i) * amc::xs::XS_WORDS,
                        amc::xs::XS_WORDS * sizeof(double));
        amc::xs::rec_from_plain(records + (size_t)(l * AMC_GD_STRIDE + AMC_GD_N) * amc::xs::XS_WORDS, (double)h->M * (double)q_batch);
    }
    return AMC_OK;
}

static amc::PgIds make_ids(int n_learn, const int* learn_ids)
{
    amc::PgIds ids;
    for (int l = 0; l < AMC_MAX_LEARN; ++l) ids.v[l] = l < n_learn ? learn_ids[l] : 0;
    return ids;
}

// The gather of the shards' records d_out[ranks][n_cols][XS_WORDS] (each shard filled its slot, zeroed the others): one in-place
// all-reduce(sum) on the engine's stream.
static int pg_allreduce_records(amc_handle* h, int n_cols)
{
    const size_t n_words = (size_t)h->comm_ranks * (size_t)n_cols * amc::xs::XS_WORDS;
    const int e = h->rccl.AllReduce(h->d_out, h->d_out, n_words, /*ncclFloat64*/ 8, /*ncclSum*/ 0, h->comm, h->stream);
    if (e != 0) return fail(AMC_ERR_COMM, "ncclAllReduce failed: %s", h->rccl.GetErrorString ? h->rccl.GetErrorString(e) : "?");
    // one communicator, two streams: a collective amc_allreduce_sum queues on comm_stream later must start after this one
    AMC_HIP(hipEventRecord(h->ev_comm_main, h->stream));
    h->comm_main_pending = true;
    return AMC_OK;
}

// make_step!(::PolicyGradientUpdate) for a policy with several parameters: one tiny launch per learnable move
static int pg_update_np(amc_handle* h, int n_learn, const int* learn_ids, const amc::PgOpts& opt)
{
    for (int l = 0; l < n_learn; ++l) {
        hipLaunchKernelGGL(amc::pg_update_np_kernel, dim3(1), dim3(64), 0, h->stream, h->d_ptab, h->d_gd_acc, h->n_params, learn_ids[l],
                           opt.kind[l], opt.h0[l], opt.h1[l], h->d_status);
        AMC_HIP(hipGetLastError());
    }
    return AMC_OK;
}

// make_step!(::PolicyGradientEstimator) on the device, optionally followed by make_step!(::PolicyGradientUpdate)
// (opt != nullptr).  Single shard: ONE launch (the estimator kernel's last block folds, accumulates and, if asked,
// takes the learning step).  Shards connected by amc_comm_init: estimator launch, in-place all-reduce, then the
// small accumulate (and update) kernels.
// may_defer: the caller's next launch is a fused time step of the same configuration that does NOT also form the callback sums
// (amc_pgmc_steps knows its own loop), so the learning step may be left to that launch's prologue -- see below
static int pg_accumulate_impl(amc_handle* h, int n_learn, const int* learn_ids, int q_batch, const amc::PgOpts* opt,
                              bool with_sweep = false, bool reduce = false, int* grid_out = nullptr, bool may_defer = false)
{
    int nl = 0;
    bool general = true;
    { const int rcc = pg_check_call(h, "amc_pg_accumulate", n_learn, learn_ids, q_batch); if (rcc != AMC_OK) return rcc; }
    { const int rcg = class_general_route(h, n_learn, with_sweep, reduce, q_batch, &general); if (rcg != AMC_OK) return rcg; }
    if (np_move_chain(h, n_learn)) {
        // several parameters, several learnable moves, one shard: one launch per move whose own tail folds, accumulates and (opt) takes
        // the move's learning step -- what np_single_launch does for one move, move after move.  A move's estimator reads its own
        // parameters and the positions only (estimator.jl:111-129), so a learning step taken before the next move's samples are drawn
        // changes nothing those samples see; the draws are named by the move's index in the call (l_base), as ever.
        uint64_t ids_mask = 0;
        for (int l = 0; l < n_learn; ++l) {
            // (with_sweep: the time step's make_step!(::Metropolis) rides in the first move's launch, as in the one-move fused step)
            const int rc = pg_launch(h, "amc_pg_accumulate", 1, learn_ids + l, q_batch, &nl, opt ? 3 : 2, opt, with_sweep && l == 0, false, nullptr, l,
                                     l + 1 == n_learn, learn_ids, n_learn);
            if (rc != AMC_OK) return rc;
            ids_mask |= 1ull << (learn_ids[l] & 63);
        }
        if (opt) h->gd_nonzero &= ~ids_mask;
        else h->gd_nonzero |= ids_mask;
        return AMC_OK;
    }
    if ((per_move_launches(h) && !(np_single_launch(h, n_learn) && !h->comm)) || !general) {
        // per learnable move: estimator launch (records in d_out), the gather across shards, gradients_data[k] += gd
        if (with_sweep) return fail(AMC_ERR_STATE, "amc_pg_accumulate: no fused time step for a policy with several parameters (a pool of several classes whose fused form does not build)");
        if (n_learn < 0 || n_learn > AMC_MAX_LEARN) return fail(AMC_ERR_BAD_ARG, "amc_pg_accumulate: n_learn must be in [0, %d]", AMC_MAX_LEARN);
        if (n_learn == 0) return pg_launch(h, "amc_pg_accumulate", 0, learn_ids, q_batch, &nl);
        const int nc = amc::pg_n_columns(h->n_params);
        const int ranks = h->comm ? h->comm_ranks : 1;
        for (int l = 0; l < n_learn; ++l) {
            int rc = pg_launch(h, "amc_pg_accumulate", 1, learn_ids + l, q_batch, &nl, 1, nullptr, false, false, nullptr, l, l + 1 == n_learn);
            if (rc == AMC_OK && h->comm) rc = pg_allreduce_records(h, nc);
            if (rc != AMC_OK) return rc;
            if (h->n_params > 1)
                hipLaunchKernelGGL(amc::pg_accumulate_np_kernel, dim3(1), dim3(64), 0, h->stream, h->d_out, ranks, h->n_params, learn_ids[l],
                                   (double)h->M_global * (double)q_batch, h->d_gd_acc);
            else          // one parameter (a pool of several classes): the one-parameter accumulators, one move at a time
                hipLaunchKernelGGL(amc::pg_accumulate_kernel, dim3(1), dim3(64), 0, h->stream, h->d_out, ranks, 1, make_ids(1, learn_ids + l),
                                   (double)h->M_global * (double)q_batch, h->d_gd_acc);
            AMC_HIP(hipGetLastError());
        }
        if (opt && h->n_params > 1) return pg_update_np(h, n_learn, learn_ids, *opt);
        if (opt) {
            hipLaunchKernelGGL(amc::pg_update_kernel, dim3(1), dim3(64), 0, h->stream, h->d_ptab, h->d_gd_acc, n_learn,
                               make_ids(n_learn, learn_ids), *opt, h->K, h->d_status);
            AMC_HIP(hipGetLastError());
        }
        return AMC_OK;
    }
    // (a communicator of ONE rank: its all-reduce is the identity -- the single-shard forms, no collective; AMC_SHARD_ROUTE_ON_ONE_RANK=1
    // sends it down the shards' route all the same: the only way to time that route's launches and collective on a one-GPU box)
    const bool shards = h->comm && (h->comm_ranks > 1 || h->shard_route_one_rank);
    // A fused time step that also updates may leave the learning step to the next launch's prologue (amc::pg_apply_pending,
    // round 5): the tail then ends at the group sums -- between shards: at this shard's records and the all-reduce behind them --
    // instead of going on through the second level of sums, a ticket and the update (single shard: 63.3 -> 62.5 us per
    // time step at 1e7 chains; shards: fused launch -> all-reduce -> next fused launch, no small launch in between).  Only behind
    // an update: gradients_data is then zero and stays untouched.  And only where the successor is known to be a plain fused step
    // (may_defer): the launch that also forms the callback sums holds four blocks per CU, and the prologue's extra microsecond
    // shows there (+2 us measured) where it pays on the plain steps (-0.75 us each) -- so a pending step never outlives the
    // amc_pgmc_steps call that left it.
    uint64_t ids_mask = 0;
    for (int l = 0; l < n_learn; ++l) ids_mask |= 1ull << (learn_ids[l] & 63);
    // More than four learnable moves of the built-in policy whose samples need the flushing form (q_batch x trips beyond what a lane's
    // accumulators take): the form of capacity 8 keeps 64 KiB of lane integers in LDS and a CU holds two of its blocks; two launches
    // of capacity <= 4 hold four each -- the pool of the reference's test/pgmc_test.jl (six learnable moves, q_batch_size = 10) at 1e7
    // chains: 29.6 -> 25.x us per sample.  The launches take the call's moves in order, so every chain sees its samples in the
    // reference's order (move-major), the draws are named by the move's index in the call (l_base): the same bits.
    if (!shards && !with_sweep && n_learn > 4 && !h->use_rtc && !h->np_small_launches) {
        PgPlan whole;
        { const int rcp = pg_plan(h, nl_capacity(n_learn), 0, false, q_batch, &whole); if (rcp != AMC_OK) return rcp; }
        if (whole.mid) {
            for (int base = 0; base < n_learn; base += 4) {
                const int n = n_learn - base < 4 ? n_learn - base : 4;
                amc::PgOpts sub;
                if (opt) {
                    sub = *opt;
                    for (int l = 0; l < n; ++l) { sub.kind[l] = opt->kind[base + l]; sub.h0[l] = opt->h0[base + l]; sub.h1[l] = opt->h1[base + l]; }
                }
                const int rcl = pg_launch(h, "amc_pg_accumulate", n, learn_ids + base, q_batch, &nl, opt ? 3 : 2, opt ? &sub : nullptr, false, false, nullptr,
                                          base, base + n >= n_learn);
                if (rcl != AMC_OK) return rcl;
            }
            if (opt) h->gd_nonzero &= ~ids_mask;
            else h->gd_nonzero |= ids_mask;
            return AMC_OK;
        }
    }
    const bool defer = may_defer && opt && with_sweep && n_learn > 0 && pg_form_defers(h, n_learn) && (h->gd_nonzero & ids_mask) == 0;
    const int tail = shards ? 1 : (defer ? (int)amc::PG_TAIL_GROUPS : (opt ? 3 : 2));
    const uint64_t t_est = h->t_est;
    int grid = 0;
    const int rc = pg_launch(h, "amc_pg_accumulate", n_learn, learn_ids, q_batch, &nl, tail, opt, with_sweep, reduce, &grid);
    if (grid_out) *grid_out = grid;
    if (rc != AMC_OK || n_learn == 0) return rc;
    if (defer) {
        if (!h->pend_consumed) {
            // the first step of a stretch (or one whose predecessor's step was taken by pg_resolve): the ring slot of this step's
            // parity is the sigma the launch has just used -- the table's.  (Otherwise block 0 of the launch wrote sigma' there.)
            // Queued behind the launch: it reads the table, this copies from it, nothing writes it in between.
            for (int l = 0; l < n_learn; ++l)
                AMC_HIP(hipMemcpyAsync(h->d_theta_ring + (size_t)(t_est & 1ull) * AMC_MAX_LEARN + l, h->d_ptab + amc::PT_SIGMA * AMC_MAX_MOVES + learn_ids[l],
                                       sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
        if (shards) { const int rca = pg_allreduce_records(h, n_learn * 4); if (rca != AMC_OK) return rca; }
        h->pend.active = true;
        h->pend.source = shards ? (int)amc::PG_PENDING_RECORDS : (int)amc::PG_PENDING_GROUPS;
        h->pend.groups = (grid + amc::PG_GROUP - 1) / amc::PG_GROUP;
        h->pend.n_learn = n_learn;
        h->pend.t_est = t_est;
        return AMC_OK;
    }
    if (opt) h->gd_nonzero &= ~ids_mask;      // an update leaves gradients_data of its moves zero, an estimator step alone does not
    else h->gd_nonzero |= ids_mask;
    if (!shards) return AMC_OK;
    // shards: the launch wrote this shard's records into its slot of d_out[ranks][n_learn * 4][XS_WORDS] and zeroed the other
    // slots, so ONE in-place all-reduce(sum) on the engine's stream is a gather -- exact whatever order RCCL adds in; the
    // kernel behind it merges the shards' integer totals and rounds once (pg_merge_slots): every shard, and a single shard
    // holding all the chains, arrive at the same bits
    { const int rca = pg_allreduce_records(h, n_learn * 4); if (rca != AMC_OK) return rca; }
    const double n_samples = (double)h->M_global * (double)q_batch;
    if (opt) {      // gradients_data += gd and the learning step in ONE launch: both sit on the critical path of the next sweep
        // (a launch that took a pending step in its prologue proposed with sigma', which block 0 left in this step's ring slot)
        const double* theta_used = h->pend_consumed ? h->d_theta_ring + (size_t)(t_est & 1ull) * AMC_MAX_LEARN : nullptr;
        hipLaunchKernelGGL(amc::pg_accumulate_update_kernel, dim3(1), dim3(64), 0, h->stream, h->d_out, h->comm_ranks, h->d_ptab,
                           h->d_gd_acc, n_learn, make_ids(n_learn, learn_ids), n_samples, *opt, h->K, h->d_status, theta_used);
    } else {
        hipLaunchKernelGGL(amc::pg_accumulate_kernel, dim3(1), dim3(64), 0, h->stream, h->d_out, h->comm_ranks, n_learn,
                           make_ids(n_learn, learn_ids), n_samples, h->d_gd_acc);
    }
    AMC_HIP(hipGetLastError());
    return AMC_OK;
}

int amc_pg_route(amc_handle* h, int n_learn, int q_batch, int fused, char* why, int why_capacity)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_pg_route: NULL handle");
    if (n_learn < 1 || n_learn > AMC_MAX_LEARN) return fail(AMC_ERR_BAD_ARG, "amc_pg_route: n_learn must be in [1, %d]", AMC_MAX_LEARN);
    if (q_batch < 1 || q_batch > AMC_MAX_QBATCH) return fail(AMC_ERR_BAD_ARG, "amc_pg_route: q_batch must be in [1, %d]", AMC_MAX_QBATCH);
    if (why && why_capacity > 0) why[0] = 0;
    AMC_HIP(hipSetDevice(h->device));
    // the conditions of pgmc_steps_impl's `fused`: the sweep rides in the estimator launch
    const bool can_fuse = (!per_move_launches(h) || (np_single_launch(h, n_learn) && !h->comm)) && h->sweepstep == 1 && (h->d_log != nullptr || h->K == 1) &&
                          n_learn <= 2 && log_form(h) != AMC_LOG_BYTES && std::getenv("AMC_NO_SWEEP_ESTIMATOR_FUSION") == nullptr;
    auto say_why = [&]() {
        if (why && why_capacity > 0 && !h->class_form_error.empty()) {
            std::strncpy(why, h->class_form_error.c_str(), (size_t)why_capacity - 1);
            why[why_capacity - 1] = 0;
        }
    };
    if (fused != 0 && can_fuse) {
        bool general = true;
        { const int rc = class_general_route(h, n_learn, true, false, q_batch, &general); if (rc != AMC_OK) return rc; }
        if (general) return 2;
        say_why();
    }
    if (per_move_launches(h)) return np_single_launch(h, n_learn) ? 1 : 0;
    bool general = true;
    { const int rc = class_general_route(h, n_learn, false, false, q_batch, &general); if (rc != AMC_OK) return rc; }
    if (!general) say_why();
    return general ? 1 : 0;
}

int amc_pg_accumulate(amc_handle* h, int n_learn, const int* learn_ids, int q_batch)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_pg_accumulate: NULL handle");
    return pg_accumulate_impl(h, n_learn, learn_ids, q_batch, nullptr);
}

static int make_opts(amc_handle* h, int n_learn, const int* learn_ids, const int* optimiser, const double* hyper0,
                     const double* hyper1, amc::PgOpts* opt)
{
    for (int l = 0; l < AMC_MAX_LEARN; ++l) {
        opt->kind[l] = 0; opt->h0[l] = 0.0; opt->h1[l] = 0.0;
    }
    for (int l = 0; l < n_learn; ++l) {
        if (learn_ids[l] < 0 || learn_ids[l] >= h->K) return fail(AMC_ERR_BAD_ARG, "amc_pg_update: learn_ids[%d] out of range", l);
        if (optimiser[l] < AMC_OPT_STATIC || optimiser[l] > AMC_OPT_BLANPG)
            return fail(AMC_ERR_BAD_ARG, "amc_pg_update: No learning_step! is defined for optimiser id %d", optimiser[l]);
        opt->kind[l] = optimiser[l]; opt->h0[l] = hyper0[l]; opt->h1[l] = hyper1[l];
    }
    return AMC_OK;
}

int amc_pg_update(amc_handle* h, int n_learn, const int* learn_ids, const int* optimiser, const double* hyper0,
                  const double* hyper1)
{
    if (!h || (n_learn > 0 && (!learn_ids || !optimiser || !hyper0 || !hyper1)))
        return fail(AMC_ERR_BAD_ARG, "amc_pg_update: NULL argument");
    if (n_learn < 0 || n_learn > AMC_MAX_LEARN) return fail(AMC_ERR_BAD_ARG, "amc_pg_update: n_learn must be in [0, %d]", AMC_MAX_LEARN);
    if (n_learn == 0) return AMC_OK;
    amc::PgOpts opt;
    { const int rc = make_opts(h, n_learn, learn_ids, optimiser, hyper0, hyper1, &opt); if (rc != AMC_OK) return rc; }
    AMC_HIP(hipSetDevice(h->device));
    { const int rc = pg_resolve(h); if (rc != AMC_OK) return rc; }
    for (int l = 0; l < n_learn; ++l) h->gd_nonzero &= ~(1ull << (learn_ids[l] & 63));
    if (h->n_params > 1) return pg_update_np(h, n_learn, learn_ids, opt);
    hipLaunchKernelGGL(amc::pg_update_kernel, dim3(1), dim3(64), 0, h->stream, h->d_ptab, h->d_gd_acc, n_learn,
                       make_ids(n_learn, learn_ids), opt, h->K, h->d_status);
    AMC_HIP(hipGetLastError());
    return AMC_OK;
}

// reduce: after the last time step, begin a reduction of the state it leaves (see amc_pgmc_steps_reduce_begin)
static int pgmc_steps_impl(amc_handle* h, const char* who, int64_t n_steps, int n_learn, const int* learn_ids, int q_batch,
                           int do_update, const int* optimiser, const double* hyper0, const double* hyper1, bool reduce)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "%s: NULL handle", who);
    if (n_steps < 0) return fail(AMC_ERR_BAD_ARG, "%s: n_steps < 0", who);
    if (n_learn < 0 || n_learn > AMC_MAX_LEARN) return fail(AMC_ERR_BAD_ARG, "%s: n_learn must be in [0, %d]", who, AMC_MAX_LEARN);
    if (n_learn > 0 && (!learn_ids || (do_update && (!optimiser || !hyper0 || !hyper1))))
        return fail(AMC_ERR_BAD_ARG, "%s: NULL argument", who);
    if (reduce && n_steps < 1) return fail(AMC_ERR_BAD_ARG, "%s: n_steps must be >= 1", who);
    if (reduce && !red_next(h)) return fail(AMC_ERR_STATE, "%s: %d reductions are already in flight (call amc_reduce_end)", who, RED_TICKETS);
    amc::PgOpts opt;
    if (do_update && n_learn > 0) {
        const int rc = make_opts(h, n_learn, learn_ids, optimiser, hyper0, hyper1, &opt);
        if (rc != AMC_OK) return rc;
    }
    // the three make_step!s of one time step (src/simulation.jl:185-190), n_steps times, from one host call: two
    // launches per step on a single shard (sweep; estimator whose last block accumulates and takes the learning step)
    // ... and ONE launch per step when the sweep can ride in the estimator launch: sweepstep = 1, at most two learnable
    // moves (the kernel forms offered with a leading sweep: K = 1 with either counter form, K > 1 with its step log)
    // (pools of more than AMC_PACKED_LOG_MOVES moves take the two launches: the fused forms write the packed step log)
    bool fused = (!per_move_launches(h) || (np_single_launch(h, n_learn) && !h->comm)) && h->sweepstep == 1 && (h->d_log != nullptr || h->K == 1) && n_learn >= 1 && n_learn <= 2 &&
                 log_form(h) != AMC_LOG_BYTES && std::getenv("AMC_NO_SWEEP_ESTIMATOR_FUSION") == nullptr;
    // several parameters, several learnable moves, one shard: the sweep rides in the first move's launch of the chain
    const bool chain = np_move_chain(h, n_learn);
    if (chain)
        fused = h->sweepstep == 1 && (h->d_log != nullptr || h->K == 1) && log_form(h) != AMC_LOG_BYTES && std::getenv("AMC_NO_SWEEP_ESTIMATOR_FUSION") == nullptr;
    if (fused && h->n_classes > 1) {      // a pool of several classes: the fused form where it builds (class_general_route)
        const int rcg = class_general_route(h, n_learn, true, false, q_batch, &fused);
        if (rcg != AMC_OK) return rcg;
    }
    // the callback sums ride in the last fused launch (rows the host sums: K <= 4; launches that need no flush on the way)
    bool fused_reduce = reduce && fused && !chain && h->K <= 4 && !h->d_acc_base;
    if (fused_reduce && h->n_classes > 1) {
        const int rcg = class_general_route(h, n_learn, true, true, q_batch, &fused_reduce);
        if (rcg != AMC_OK) return rcg;
    }
    if (fused_reduce) {            // ... and launches that need no flush on the way
        PgPlan plan;
        const int rcp = pg_plan(h, nl_capacity(n_learn), h->K > 1 ? 2 : (h->d_log ? 1 : 3), true, q_batch, &plan);
        if (rcp != AMC_OK) return rcp;
        fused_reduce = !plan.mid && reduce_fits_in_grid(h, plan.grid);
    }
    int grid = 0;
    for (int64_t i = 0; i < n_steps; ++i) {
        int rc = fused ? AMC_OK : sweep_impl(h, 1, false, nullptr);
        if (rc == AMC_OK)
            rc = pg_accumulate_impl(h, n_learn, learn_ids, q_batch, (do_update && n_learn > 0) ? &opt : nullptr, fused,
                                    fused_reduce && i + 1 == n_steps, &grid,
                                    /* may_defer: */ fused && i + 1 < n_steps && !(fused_reduce && i + 2 == n_steps));
        if (rc != AMC_OK) return rc;
    }
    if (!reduce) return AMC_OK;
    return fused_reduce ? finish_fused_reduce(h, grid) : amc_reduce_begin(h);
}

int amc_pgmc_steps(amc_handle* h, int64_t n_steps, int n_learn, const int* learn_ids, int q_batch, int do_update,
                   const int* optimiser, const double* hyper0, const double* hyper1)
{
    return pgmc_steps_impl(h, "amc_pgmc_steps", n_steps, n_learn, learn_ids, q_batch, do_update, optimiser, hyper0, hyper1, false);
}

int amc_pgmc_steps_reduce_begin(amc_handle* h, int64_t n_steps, int n_learn, const int* learn_ids, int q_batch, int do_update,
                                const int* optimiser, const double* hyper0, const double* hyper1)
{
    return pgmc_steps_impl(h, "amc_pgmc_steps_reduce_begin", n_steps, n_learn, learn_ids, q_batch, do_update, optimiser, hyper0,
                           hyper1, true);
}

int amc_pg_get_accumulated(amc_handle* h, int n_learn, const int* learn_ids, double* out)
{
    if (!h || !out || (n_learn > 0 && !learn_ids)) return fail(AMC_ERR_BAD_ARG, "amc_pg_get_accumulated: NULL argument");
    AMC_HIP(hipSetDevice(h->device));
    { const int rc = pg_resolve(h); if (rc != AMC_OK) return rc; }      // (a pending step may set the status flag)
    const int np = h->n_params;
    const size_t dev_stride = np > 1 ? (size_t)AMC_GD_STRIDE_MAX : 5, out_stride = (size_t)amc::pg_gd_stride(np);
    std::vector<double> acc((size_t)AMC_MAX_MOVES * dev_stride);
    int status = 0;
    AMC_HIP(hipMemcpyAsync(acc.data(), h->d_gd_acc, acc.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    AMC_HIP(hipMemcpyAsync(&status, h->d_status, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    AMC_HIP(hipStreamSynchronize(h->stream));
    for (int l = 0; l < n_learn; ++l) {
        if (learn_ids[l] < 0 || learn_ids[l] >= h->K) return fail(AMC_ERR_BAD_ARG, "amc_pg_get_accumulated: learn_ids[%d] out of range", l);
        for (size_t i = 0; i < out_stride; ++i) out[(size_t)l * out_stride + i] = acc[(size_t)learn_ids[l] * dev_stride + i];
    }
    if (status != 0)
        return fail(AMC_ERR_STATE, np > 1 ? "a learning step produced a parameter that is not finite (or met a singular metric) and was not applied"
                                          : "a learning step produced a sigma outside [1e-100, 1e100] (or NaN) and was not applied");
    return AMC_OK;
}

int amc_pg_set_accumulated(amc_handle* h, int n_learn, const int* learn_ids, const double* in)
{
    if (!h || (n_learn > 0 && (!learn_ids || !in))) return fail(AMC_ERR_BAD_ARG, "amc_pg_set_accumulated: NULL argument");
    if (n_learn < 0 || n_learn > AMC_MAX_LEARN) return fail(AMC_ERR_BAD_ARG, "amc_pg_set_accumulated: n_learn must be in [0, %d]", AMC_MAX_LEARN);
    for (int l = 0; l < n_learn; ++l) {
        if (learn_ids[l] < 0 || learn_ids[l] >= h->K) return fail(AMC_ERR_BAD_ARG, "amc_pg_set_accumulated: learn_ids[%d] out of range", l);
        const double n = in[(size_t)l * amc::pg_gd_stride(h->n_params) + amc::pg_gd_stride(h->n_params) - 1];
        if (!(n >= 0.0) || n != std::floor(n)) return fail(AMC_ERR_BAD_ARG, "amc_pg_set_accumulated: n of move %d is not a sample count", learn_ids[l]);
    }
    AMC_HIP(hipSetDevice(h->device));
    { const int rc = pg_resolve(h); if (rc != AMC_OK) return rc; }
    for (int l = 0; l < n_learn; ++l) h->gd_nonzero |= 1ull << (learn_ids[l] & 63);
    {
        const size_t dev_stride = h->n_params > 1 ? (size_t)AMC_GD_STRIDE_MAX : 5, in_stride = (size_t)amc::pg_gd_stride(h->n_params);
        for (int l = 0; l < n_learn; ++l)
            AMC_HIP(hipMemcpyAsync(h->d_gd_acc + (size_t)learn_ids[l] * dev_stride, in + (size_t)l * in_stride, in_stride * sizeof(double),
                                   hipMemcpyHostToDevice, h->stream));
    }
    AMC_HIP(hipStreamSynchronize(h->stream));      // the caller's buffer is only valid during the call
    return AMC_OK;
}

}  // extern "C"
