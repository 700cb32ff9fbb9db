// amc_kernels.h -- HIP kernels of the many-chain Metropolis engine (gfx950 / CDNA4).
//
// Data layout in HBM: SoA, one f64 per chain
//   x[M_pad]                      chain positions (Particle.x); e is NOT stored: it is
//                                 potential(x) by construction (particle_1d.jl:13-15,33)
//   beta[M_pad]      (optional)   per-chain Particle.beta
//   acc[K][M_pad], tot[K-1][M_pad]  u32 Move.accepted_calls / total_calls (when kept; the last move's total_calls is the
//                                 step count minus the other moves': every chain takes the same number of steps)
//   ptab[PT_ROWS][AMC_MAX_MOVES]  per-move derived parameters (device-computed)
// One lane owns TWO adjacent chains (one global "pair"): 16-byte loads/stores, one
// Box-Muller and one accept-uniform Philox call serve both chains.
// All kernels are HBM-streaming in shape but f64-VALU-bound in practice; no MFMA.
//
// The sources, in include order (each includes its predecessor; hiprtc gets all of them by name, embed_sources.py):
//   amc_model.h        configuration, script-defined hooks, potential / proposal / acceptance, one mc_step! of a chain pair
//   amc_wave_sums.h    device side of the reproducible cross-chain sums (wave totals, lane accumulators, block rows)
//   amc_sweep.h        K1: sweep_kernel, the step log and its fold into the per-chain counters
//   amc_params.h       initial ensemble, parameter / pick tables, the device-resident learning step
//   amc_reduce_pass.h  K2a: the callback reductions as a pass of their own; counter passes
//   amc_pg_tail.h      estimator launch records, GradientData samples, accumulate / update kernels, pending learning steps
//   amc_estimator.h    K3: pg_estimate_kernel (optionally fused with the sweep and the callback sums)
//   amc_aux_kernels.h  histogram / energy / conversion kernels, parity-test hooks
#pragma once

#include "amc_model.h"
#include "amc_wave_sums.h"
#include "amc_sweep.h"
#include "amc_params.h"
#include "amc_reduce_pass.h"
#include "amc_pg_tail.h"
#include "amc_estimator.h"
#include "amc_aux_kernels.h"
