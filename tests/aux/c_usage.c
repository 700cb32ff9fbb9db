/* Plain-C client of libamc.so: no Python, no C++ -- what a `ccall` / cgo / JNI binding sees. */
#include <math.h>
#include <stdio.h>
#include "amc.h"

int main(void)
{
    double sigma[2] = {0.1, 1.0}, weight[2] = {0.5, 0.5}, red[AMC_RED_HEADER + 2];
    amc_config cfg = {sizeof(amc_config), 0, 200000, 0, 200000, AMC_POTENTIAL_DOUBLE_WELL, 2, 2.0,
                      sigma, weight, 1, 1, 1, NULL};
    amc_handle *h = NULL;
    if (amc_create(&cfg, &h)) { fprintf(stderr, "amc_create: %s\n", amc_last_error()); return 1; }
    if (amc_init_uniform(h, -2.0, 2.0) || amc_sweep(h, 500) || amc_reduce(h, red)) {
        fprintf(stderr, "%s\n", amc_last_error());
        return 1;
    }
    printf("double_well mean_U %.6f mean_x2 %.6f acc0 %.5f acc1 %.5f\n", red[AMC_RED_SUM_E] / red[AMC_RED_COUNT],
           red[AMC_RED_SUM_XX] / red[AMC_RED_COUNT], red[4] / red[AMC_RED_COUNT], red[5] / red[AMC_RED_COUNT]);
    amc_destroy(h);

    /* the script-defined potential of the reference, as a C expression compiled for the GPU at run time */
    cfg.potential = AMC_POTENTIAL_CUSTOM;
    cfg.n_moves = 1;
    weight[0] = 1.0;
    sigma[0] = 0.5;
    cfg.per_chain_counters = 0;
    if (amc_create_custom(&cfg, "0.5*x*x + 0.25*x*x*x*x", &h)) { fprintf(stderr, "custom: %s\n", amc_last_error()); return 1; }
    if (amc_init_uniform(h, -2.0, 2.0) || amc_sweep(h, 500) || amc_reduce(h, red)) {
        fprintf(stderr, "%s\n", amc_last_error());
        return 1;
    }
    printf("custom mean_U %.6f mean_x %.6f\n", red[AMC_RED_SUM_E] / red[AMC_RED_COUNT], red[AMC_RED_SUM_X] / red[AMC_RED_COUNT]);
    amc_destroy(h);

    /* Particle{Float32}: the same calls, Float32 state on the device, double host buffers */
    cfg.potential = AMC_POTENTIAL_HARMONIC;
    cfg.state_dtype = AMC_DTYPE_F32;
    sigma[0] = 0.1;
    if (amc_create(&cfg, &h)) { fprintf(stderr, "f32: %s\n", amc_last_error()); return 1; }
    if (amc_init_uniform(h, -2.0, 2.0) || amc_sweep(h, 2000) || amc_reduce(h, red)) {
        fprintf(stderr, "%s\n", amc_last_error());
        return 1;
    }
    printf("f32 harmonic mean_U %.6f acc %.5f\n", red[AMC_RED_SUM_E] / red[AMC_RED_COUNT], red[4] / red[AMC_RED_COUNT]);
    amc_destroy(h);
    cfg.state_dtype = AMC_DTYPE_F64;

    /* errors come back as codes + message, never as exceptions */
    cfg.n_moves = 0;
    if (amc_create(&cfg, &h) != AMC_ERR_BAD_ARG) return 2;
    printf("error path: %s\n", amc_last_error());

    /* PolicyGuided MC, config 5's pool: ten time steps of [Metropolis, estimator, update] from ONE call, the tenth launch also
     * forming the callback sums (amc_pgmc_steps_reduce_begin); more time steps queued behind before the sums are read */
    {
        double sg[2] = {0.2, 0.1}, w[2] = {0.6, 0.4}, eta[1] = {0.5}, zero[1] = {0.0}, s2 = 0.0, r2[AMC_RED_HEADER + 2];
        int ids[1] = {1}, opt[1] = {AMC_OPT_VPG}, n_ranks = -1, rank = -1, version = -1, hip_version = 0, i;
        char path[256], hip_path[256];
        amc_config c5 = {sizeof(amc_config), 0, 200000, 0, 200000, AMC_POTENTIAL_HARMONIC, 2, 2.0, sg, w, 42, 1, 1, NULL, AMC_DTYPE_F64, 0};
        if (amc_create(&c5, &h) || amc_init_uniform(h, -2.0, 2.0)) { fprintf(stderr, "pgmc: %s\n", amc_last_error()); return 1; }
        for (i = 0; i < 30; ++i) {
            if (amc_pgmc_steps_reduce_begin(h, 10, 1, ids, 1, 1, opt, eta, zero) || amc_pgmc_steps(h, 3, 1, ids, 1, 1, opt, eta, zero) ||
                amc_reduce_end(h, r2)) { fprintf(stderr, "pgmc: %s\n", amc_last_error()); return 1; }
        }
        if (amc_get_parameters(h, 1, &s2, 1)) return 1;
        if (amc_pgmc_steps_reduce_begin(h, 1, 1, ids, 1, 1, opt, eta, zero)) return 1;
        if (amc_pgmc_steps_reduce_begin(h, 1, 1, ids, 1, 1, opt, eta, zero)) return 1;                         /* two may be in flight */
        if (amc_pgmc_steps_reduce_begin(h, 1, 1, ids, 1, 1, opt, eta, zero) != AMC_ERR_STATE) return 3;      /* ... a third may not */
        if (amc_reduce_end(h, r2)) return 1;                                                                   /* the oldest first */
        if (amc_reduce_end(h, r2)) return 1;
        /* no communicator: one rank, no library; and which HIP runtime this process is bound to */
        if (amc_comm_info(h, &n_ranks, &rank, &version, path, (int)sizeof(path)) || amc_runtime_info(&hip_version, hip_path, (int)sizeof(hip_path)))
            return 1;
        if (n_ranks != 1 || rank != 0 || version != 0 || path[0] != 0 || hip_version <= 0) return 4;
        printf("pgmc sigma2 %.6f mean_e %.6f acc0 %.5f acc1 %.5f hip %d\n", s2, r2[AMC_RED_SUM_E] / r2[AMC_RED_COUNT],
               r2[4] / r2[AMC_RED_COUNT], r2[5] / r2[AMC_RED_COUNT], hip_version);
        amc_destroy(h);
    }
    return 0;
}
