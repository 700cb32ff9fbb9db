// The fused "one make_step!(::Metropolis) + one estimator call" kernels of pools with K > 1 (pg_estimate_kernel<.., SWEEP = 2>),
// compiled on their own.  Reason: with LLVM's Machine LICM hoisting every loop-invariant it can find out of their sampling loop
// these kernels need 103 VGPRs (4 waves per SIMD) and park 25 scalars in VGPR lanes; with that pass off (the Makefile gives this
// file `-mllvm -disable-machine-licm`, the only way: it is no per-function option) they need 90 (5 waves) and spill nothing --
// 64.5-65.4 instead of 66.5-66.8 us per launch at 1e7 chains on one box.  The other kernels lose 0-2 % with the pass off, so
// they stay in amc_api.hip.  Same sources, same arithmetic: the parity tests run over both objects.
#include <hip/hip_runtime.h>

#define AMC_KERNEL_LINKAGE static      // only the instantiations below: none of the header's plain kernels in this object
#include "amc_kernels.h"

namespace amc {
// (.., RED_FORM_NONE): the time step alone; (.., RED_FORM_COLS / RED_FORM_E): with the callback sums of the state it leaves (REDUCE: those
// SweepArgs.red_cols names / sum e alone); (.., RED_FORM_NONE, true): the time step of launches whose lanes empty their accumulators on
// the way (MIDFLUSH: large q_batch)
#define AMC_PG_FUSED(POT, NL, BETA)                                                                              \
    template __global__ void pg_estimate_kernel<POT, NL, BETA, 2, RED_FORM_NONE>(const PgArgs, const SweepArgs);   \
    template __global__ void pg_estimate_kernel<POT, NL, BETA, 2, RED_FORM_COLS>(const PgArgs, const SweepArgs);   \
    template __global__ void pg_estimate_kernel<POT, NL, BETA, 2, RED_FORM_E>(const PgArgs, const SweepArgs);      \
    template __global__ void pg_estimate_kernel<POT, NL, BETA, 2, RED_FORM_NONE, true>(const PgArgs, const SweepArgs)
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
